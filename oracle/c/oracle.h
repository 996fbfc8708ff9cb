/* ORACLE (test infrastructure; never linked into or called by the product path). */
#ifndef ORACLE_H
#define ORACLE_H

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

void orc_fft(double* re, double* im, int n, int sign);
void orc_rfft(const double* x, int n, double* Xr, double* Xi, double* work);
void orc_irfft(const double* Xr, const double* Xi, int n, double* x, double* work);

/* shared WORLD helpers (world.c) */
int orc_mround(double x);
void orc_interp1(const double* x, const double* y, int n, const double* xi, int m, double* yi);
void orc_nuttall(int n, double* w);

/* Harvest (harvest.c) */
void orc_harvest_set_mirror_write(int on);
int orc_harvest_num_frames(int xl, int fs, double frame_period);
void orc_decimate_coefficients(int r, double* a3, double* b2);
int orc_harvest_waveform(const double* x, int xl, int fs, double* y);
int orc_harvest(const double* x, int xl, int fs, double frame_period, double f0_floor,
                double f0_ceil, double* f0_out, double* tp_out);
int orc_harvest_debug(const double* x, int xl, int fs, double frame_period, double f0_floor,
                      double f0_ceil, double* f0_out, double* tp_out, double* raw_out,
                      double* cand_out, double* score_out, double* best_out, int* dims);

/* WORLD's randn() (common.cpp / matlabfunctions.cpp): xorshift128 with the fixed seed that
 * randn_reseed() restores at the start of CheapTrick / D4C / Synthesis; one normal deviate is
 * the sum of 12 draws of 28 bits. */
#include <stdint.h>
typedef struct { uint32_t x, y, z, w; } rng_t;
static inline void rng_seed(rng_t* s) { s->x = 123456789u; s->y = 362436069u; s->z = 521288629u; s->w = 88675123u; }
static inline double rng_randn(rng_t* s) {
  uint32_t t = s->x ^ (s->x << 11);
  s->x = s->y; s->y = s->z; s->z = s->w;
  s->w = (s->w ^ (s->w >> 19)) ^ (t ^ (t >> 8));
  uint32_t tmp = s->w >> 4;
  for (int i = 0; i < 11; ++i) {
    t = s->x ^ (s->x << 11);
    s->x = s->y; s->y = s->z; s->z = s->w;
    s->w = (s->w ^ (s->w >> 19)) ^ (t ^ (t >> 8));
    tmp += s->w >> 4;
  }
  return tmp / 268435456.0 - 6.0;
}

#endif
