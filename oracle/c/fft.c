/* ORACLE (test infrastructure; never linked into or called by the product path).
 * Plain radix-2 double-precision FFTs used by the CPU restatements in world.c / sptk.c /
 * synth.c.  (WORLD uses Ooura's fft, SPTK its own; both are mathematically the DFT.) */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAX_LOG2 24
typedef struct {
  double* wr; /* cos(2 pi k / n), k < n/2 */
  double* wi; /* sin(2 pi k / n) */
} twiddle_t;
static twiddle_t g_tw[MAX_LOG2 + 1];

static int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}

static const twiddle_t* get_tw(int n) {
  const int l = ilog2(n);
  if (!g_tw[l].wr) {
    const int h = n / 2 > 0 ? n / 2 : 1;
    g_tw[l].wr = (double*)malloc(sizeof(double) * h);
    g_tw[l].wi = (double*)malloc(sizeof(double) * h);
    for (int k = 0; k < h; ++k) {
      const double a = 2.0 * M_PI * (double)k / (double)n;
      g_tw[l].wr[k] = cos(a);
      g_tw[l].wi[k] = sin(a);
    }
  }
  return &g_tw[l];
}

/* in-place complex DFT, n a power of two. sign = -1: X[k] = sum x[j] e^{-2 pi i jk/n};
 * sign = +1: unnormalised inverse. */
void orc_fft(double* re, double* im, int n, int sign) {
  if (n <= 1) return;
  for (int i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      double t = re[i]; re[i] = re[j]; re[j] = t;
      t = im[i]; im[i] = im[j]; im[j] = t;
    }
  }
  const twiddle_t* tw = get_tw(n);
  for (int len = 2; len <= n; len <<= 1) {
    const int half = len >> 1, step = n / len;
    for (int i = 0; i < n; i += len) {
      for (int k = 0; k < half; ++k) {
        const double wr = tw->wr[k * step];
        const double wi = sign < 0 ? -tw->wi[k * step] : tw->wi[k * step];
        const int a = i + k, b = a + half;
        const double xr = re[b] * wr - im[b] * wi;
        const double xi = re[b] * wi + im[b] * wr;
        re[b] = re[a] - xr; im[b] = im[a] - xi;
        re[a] += xr; im[a] += xi;
      }
    }
  }
}

/* real input x[n] (n power of two >= 2) -> X[0..n/2] (numpy.fft.rfft). work: 2*(n/2) doubles */
void orc_rfft(const double* x, int n, double* Xr, double* Xi, double* work) {
  const int h = n / 2;
  double* zr = work;
  double* zi = work + h;
  for (int k = 0; k < h; ++k) { zr[k] = x[2 * k]; zi[k] = x[2 * k + 1]; }
  orc_fft(zr, zi, h, -1);
  const twiddle_t* tw = get_tw(n);
  for (int k = 0; k <= h; ++k) {
    const int a = k % h, b = (h - k) % h;
    const double ar = zr[a], ai = zi[a], br = zr[b], bi = -zi[b]; /* conj(Z[h-k]) */
    const double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
    /* O = (Z[k] - conj(Z[h-k])) / (2i) */
    const double dr = 0.5 * (ar - br), di = 0.5 * (ai - bi);
    const double or_ = di, oi = -dr;
    double wr, wi; /* w^k = e^{-2 pi i k / n} */
    if (k == h) { wr = -1.0; wi = 0.0; } else { wr = tw->wr[k]; wi = -tw->wi[k]; }
    Xr[k] = er + (or_ * wr - oi * wi);
    Xi[k] = ei + (or_ * wi + oi * wr);
  }
}

/* X[0..n/2] -> real x[n], normalised (numpy.fft.irfft). Imaginary parts of X[0], X[n/2] are
 * ignored like numpy does. work: 2*(n/2) doubles */
void orc_irfft(const double* Xr, const double* Xi, int n, double* x, double* work) {
  const int h = n / 2;
  double* zr = work;
  double* zi = work + h;
  const twiddle_t* tw = get_tw(n);
  for (int k = 0; k < h; ++k) {
    const double ar = Xr[k], ai = (k == 0) ? 0.0 : Xi[k];
    const double br = Xr[h - k], bi = (k == 0) ? 0.0 : -Xi[h - k]; /* conj(X[h-k]) */
    const double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
    const double dr = 0.5 * (ar - br), di = 0.5 * (ai - bi);
    /* O = D * conj(w^k), conj(w^k) = e^{+2 pi i k / n} */
    const double cr = tw->wr[k], ci = tw->wi[k];
    const double or_ = dr * cr - di * ci, oi = dr * ci + di * cr;
    /* Z = E + i O */
    zr[k] = er - oi;
    zi[k] = ei + or_;
  }
  orc_fft(zr, zi, h, +1);
  const double s = 1.0 / (double)h;
  for (int k = 0; k < h; ++k) { x[2 * k] = zr[k] * s; x[2 * k + 1] = zi[k] * s; }
}
