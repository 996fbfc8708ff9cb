/* ORACLE (test infrastructure; never linked into or called by the product path).
 *
 * CPU (fp64) restatement of WORLD DecodeAperiodicity (codec.cpp) and Synthesis (synthesis.cpp),
 * which the reference reaches through pyworld.decode_aperiodicity / pyworld.synthesize
 * (idiaptts/src/data_preparation/world/WorldFeatLabelGen.py:940-943).  pyworld is unpinned
 * (requirements.txt:6) and absent from /root/reference.
 *
 * PARITY UNPINNED: the reference holds no golden waveform (only a length check,
 * test_AcousticModelTrainer.py:161-168, and sum err^2 < 10000, test_WorldFeatLabelGen.py:761-763).
 * Variant restated: the current mmorise/World release -- fractional pulse time shift, integer
 * `fs / fft_size + 1` lowest f0, CheckVUV (mean band aperiodicity > -0.5 dB => unvoiced) in
 * DecodeAperiodicity, xorshift128 randn() reseeded per Synthesis call.
 */
#include "oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EPS 1e-12
#define DEFAULT_F0 500.0

static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

int orc_decode_aperiodicity(const double* bap, int T, int fs, int fft_size, double* ap) {
  const int K = fft_size / 2 + 1;
  const int nap = (int)(dmin(15000.0, fs / 2.0 - 3000.0) / 3000.0);
  double* fa = (double*)malloc(sizeof(double) * (K + 2 * (nap + 2)));
  if (!fa) return -1;
  double* cfa = fa + K;
  double* cap = cfa + nap + 2;
  for (int k = 0; k < K; ++k) fa[k] = (double)fs / fft_size * k;
  for (int i = 0; i <= nap; ++i) cfa[i] = i * 3000.0;
  cfa[nap + 1] = fs / 2.0;
  cap[0] = -60.0;
  cap[nap + 1] = -EPS;
  for (int t = 0; t < T; ++t) {
    double* row = ap + (size_t)t * K;
    double mean = 0.0;
    for (int i = 0; i < nap; ++i) {
      mean += bap[(size_t)t * nap + i];
      cap[i + 1] = bap[(size_t)t * nap + i];
    }
    mean /= nap;
    if (mean > -0.5) { /* CheckVUV: unvoiced */
      for (int k = 0; k < K; ++k) row[k] = 1.0 - EPS;
      continue;
    }
    orc_interp1(cfa, cap, nap + 2, fa, K, row);
    for (int k = 0; k < K; ++k) row[k] = pow(10.0, row[k] / 20.0);
  }
  free(fa);
  return 0;
}

/* log_half[0..h] (log amplitude) -> minimum phase spectrum (re, im)[0..h] */
static void min_phase(const double* log_half, int fft, double* re, double* im, double* cr, double* ci) {
  const int h = fft / 2;
  for (int i = 0; i <= h; ++i) { cr[i] = log_half[i]; ci[i] = 0.0; }
  for (int i = h + 1; i < fft; ++i) { cr[i] = log_half[fft - i]; ci[i] = 0.0; }
  orc_fft(cr, ci, fft, -1); /* real even input: result real = fft * cepstrum */
  /* cepstrum folding: c[0], 2 c[1..h-1], c[h], 0 ...; imaginary parts are (numerically) zero */
  ci[0] = -ci[0];
  for (int i = 1; i < h; ++i) { cr[i] *= 2.0; ci[i] *= -2.0; }
  ci[h] = -ci[h];
  for (int i = h + 1; i < fft; ++i) { cr[i] = 0.0; ci[i] = 0.0; }
  orc_fft(cr, ci, fft, -1);
  for (int i = 0; i <= h; ++i) {
    const double t = exp(cr[i] / fft);
    re[i] = t * cos(ci[i] / fft);
    im[i] = t * sin(ci[i] / fft);
  }
}

static void fftshift(const double* x, int n, double* y) {
  for (int i = 0; i < n / 2; ++i) { y[i] = x[i + n / 2]; y[i + n / 2] = x[i]; }
}

/* f0 [T], sp [T,K] power spectrum, ap [T,K]; y [y_length] */
int orc_synthesize(const double* f0, int T, const double* sp, const double* ap, int fft, double frame_period_ms,
                   int fs, int y_length, double* y) {
  const int h = fft / 2, K = h + 1;
  const double fp = frame_period_ms / 1000.0;
  const double lowest_f0 = (double)(fs / fft) + 1.0; /* integer division as in WORLD */
  memset(y, 0, sizeof(double) * (y_length > 0 ? y_length : 0));
  if (y_length <= 1 || T < 2) return 0;
  double* pool = (double*)malloc(sizeof(double) * ((size_t)y_length * 5 + (size_t)(T + 1) * 3 + (size_t)fft * 12 + (size_t)K * 8));
  int* pidx = (int*)malloc(sizeof(int) * y_length);
  if (!pool || !pidx) return -1;
  double* p = pool;
  double* ta = p; p += y_length;
  double* if0 = p; p += y_length;
  double* ivuv = p; p += y_length;
  double* ploc = p; p += y_length;
  double* pshift = p; p += y_length;
  double* cta = p; p += T + 1;
  double* cf0 = p; p += T + 1;
  double* cvuv = p; p += T + 1;
  double* dcr = p; p += fft;
  double* cr = p; p += fft;
  double* ci = p; p += fft;
  double* per = p; p += fft;
  double* ape = p; p += fft;
  double* tmpw = p; p += fft;
  double* nz = p; p += fft;
  double* work = p; p += fft;
  double* resp = p; p += fft;
  p += 3 * fft; /* spare */
  double* se = p; p += K;
  double* ar = p; p += K;
  double* lg = p; p += K;
  double* mr = p; p += K;
  double* mi = p; p += K;
  double* Nr = p; p += K;
  double* Ni = p; p += K;

  for (int i = 0; i < y_length; ++i) ta[i] = i / (double)fs;
  for (int i = 0; i < T; ++i) {
    cta[i] = i * fp;
    cf0[i] = f0[i] < lowest_f0 ? 0.0 : f0[i];
    cvuv[i] = cf0[i] == 0.0 ? 0.0 : 1.0;
  }
  cta[T] = T * fp;
  cf0[T] = cf0[T - 1] * 2 - cf0[T - 2];
  cvuv[T] = cvuv[T - 1] * 2 - cvuv[T - 2];
  orc_interp1(cta, cf0, T + 1, ta, y_length, if0);
  orc_interp1(cta, cvuv, T + 1, ta, y_length, ivuv);
  for (int i = 0; i < y_length; ++i) {
    ivuv[i] = ivuv[i] > 0.5 ? 1.0 : 0.0;
    if0[i] = ivuv[i] == 0.0 ? DEFAULT_F0 : if0[i];
  }
  /* pulse locations */
  int P = 0;
  {
    double total = 2.0 * M_PI * if0[0] / fs;
    double wrap_prev = fmod(total, 2.0 * M_PI);
    for (int i = 1; i < y_length; ++i) {
      total += 2.0 * M_PI * if0[i] / fs;
      const double wrap = fmod(total, 2.0 * M_PI);
      if (fabs(wrap - wrap_prev) > M_PI) {
        ploc[P] = ta[i - 1];
        pidx[P] = i - 1;
        const double y1 = wrap_prev - 2.0 * M_PI, y2 = wrap;
        pshift[P] = (-y1 / (y2 - y1)) / fs;
        ++P;
      }
      wrap_prev = wrap;
    }
  }
  /* DC remover */
  {
    double dc = 0.0;
    for (int i = 0; i < h; ++i) {
      dcr[i] = 0.5 - 0.5 * cos(2.0 * M_PI * (i + 1.0) / (1.0 + fft));
      dcr[fft - i - 1] = dcr[i];
      dc += dcr[i] * 2.0;
    }
    for (int i = 0; i < h; ++i) {
      dcr[i] /= dc;
      dcr[fft - i - 1] = dcr[i];
    }
  }
  rng_t rng;
  rng_seed(&rng);
  for (int q = 0; q < P; ++q) {
    const int noise_size = pidx[imin(P - 1, q + 1)] - pidx[q];
    const double t = ploc[q];
    const double vuv = ivuv[pidx[q]];
    const int fl = imin(T - 1, (int)floor(t / fp));
    const int ce = imin(T - 1, (int)ceil(t / fp));
    const double a = t / fp - fl;
    for (int k = 0; k < K; ++k) {
      const double s0 = fabs(sp[(size_t)fl * K + k]);
      const double a0 = dmax(0.001, dmin(0.999999999999, ap[(size_t)fl * K + k]));
      if (fl == ce) {
        se[k] = s0;
        ar[k] = pow(a0, 2.0);
      } else {
        const double s1 = fabs(sp[(size_t)ce * K + k]);
        const double a1 = dmax(0.001, dmin(0.999999999999, ap[(size_t)ce * K + k]));
        se[k] = (1.0 - a) * s0 + a * s1;
        ar[k] = (1.0 - a) * pow(a0, 2.0) + a * pow(a1, 2.0);
      }
    }
    /* periodic response */
    if (vuv <= 0.5 || ar[0] > 0.999) {
      memset(per, 0, sizeof(double) * fft);
    } else {
      for (int k = 0; k < K; ++k) lg[k] = log(se[k] * (1.0 - ar[k]) + EPS) / 2.0;
      min_phase(lg, fft, mr, mi, cr, ci);
      const double coef = 2.0 * M_PI * pshift[q] * fs / fft;
      for (int k = 0; k < K; ++k) {
        const double re2 = cos(coef * k);
        const double im2 = sqrt(1.0 - re2 * re2);
        const double re = mr[k], im = mi[k];
        mr[k] = re * re2 + im * im2;
        mi[k] = im * re2 - re * im2;
      }
      orc_irfft(mr, mi, fft, tmpw, work);
      fftshift(tmpw, fft, per);
      double dc = 0.0;
      for (int i = h; i < fft; ++i) dc += per[i];
      for (int i = 0; i < h; ++i) per[i] = -dc * dcr[i];
      for (int i = h; i < fft; ++i) per[i] -= dc * dcr[i];
    }
    /* aperiodic response */
    memset(nz, 0, sizeof(double) * fft);
    if (noise_size > 0) {
      double avg = 0.0;
      for (int i = 0; i < noise_size && i < fft; ++i) {
        nz[i] = rng_randn(&rng);
        avg += nz[i];
      }
      /* WORLD draws noise_size samples; noise_size never exceeds fft_size for f0 >= lowest */
      avg /= noise_size;
      for (int i = 0; i < noise_size && i < fft; ++i) nz[i] -= avg;
    }
    orc_rfft(nz, fft, Nr, Ni, work);
    if (vuv != 0.0)
      for (int k = 0; k < K; ++k) lg[k] = log(se[k] * ar[k]) / 2.0;
    else
      for (int k = 0; k < K; ++k) lg[k] = log(se[k]) / 2.0;
    min_phase(lg, fft, mr, mi, cr, ci);
    for (int k = 0; k < K; ++k) {
      const double re = mr[k] * Nr[k] - mi[k] * Ni[k];
      const double im = mr[k] * Ni[k] + mi[k] * Nr[k];
      mr[k] = re;
      mi[k] = im;
    }
    orc_irfft(mr, mi, fft, tmpw, work);
    fftshift(tmpw, fft, ape);
    const double sq = sqrt((double)noise_size);
    for (int i = 0; i < fft; ++i) resp[i] = per[i] * sq + ape[i];
    const int off = pidx[q] - h + 1;
    const int lo = imax(0, -off), hi = imin(fft, y_length - off);
    for (int j = lo; j < hi; ++j) y[j + off] += resp[j];
  }
  free(pidx);
  free(pool);
  return 0;
}
