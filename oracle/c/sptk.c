/* ORACLE (test infrastructure; never linked into or called by the product path).
 *
 * CPU (fp64) restatement of the SPTK routines the reference reaches through pysptk
 * (unpinned, requirements.txt:11; SPTK 3.x C sources are NOT in /root/reference):
 *   pysptk.mcep   -- idiaptts/src/data_preparation/audio/AudioProcessing.py:146-152
 *                    (itype=3 amplitude-spectrum input, etype=1, eps=1e-8, min_det=0;
 *                     pysptk defaults miniter=2, maxiter=30, threshold=1e-3)
 *   pysptk.mgc2sp -- AudioProcessing.py:252-255 (gamma=0): freqt(-alpha) + FFT, real part
 * restated from the published SPTK algorithms (mcep.c: Tokuda et al. UELS Newton iteration
 * with freqt/frqtr and a Toeplitz-plus-Hankel solve `theq`).  The linear system is solved by
 * Gaussian elimination with partial pivoting here (theq is a fast solver of the same system).
 * Pinned by the reference's golden fixtures (cmp_mcep20/.cmp columns 0..19 == mcep(order 19,
 * alpha 0.58) of the CheapTrick envelope) through tests/test_oracle_golden.py.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* c1[0..m1] -> c2[0..m2], all-pass constant a (SPTK freqt) */
static void freqt(const double* c1, int m1, double* c2, int m2, double a, double* d) {
  const double b = 1.0 - a * a;
  double* g = c2;
  memset(g, 0, sizeof(double) * (m2 + 1));
  memset(d, 0, sizeof(double) * (m2 + 1));
  for (int i = -m1; i <= 0; ++i) {
    d[0] = g[0];
    g[0] = c1[-i] + a * d[0];
    if (m2 >= 1) {
      d[1] = g[1];
      g[1] = b * d[0] + a * d[1];
    }
    for (int j = 2; j <= m2; ++j) {
      d[j] = g[j];
      g[j] = d[j - 1] + a * (d[j] - g[j - 1]);
    }
  }
}

/* SPTK frqtr (mcep.c) */
static void frqtr(const double* c1, int m1, double* c2, int m2, double a, double* d) {
  double* g = c2;
  memset(g, 0, sizeof(double) * (m2 + 1));
  memset(d, 0, sizeof(double) * (m2 + 1));
  for (int i = -m1; i <= 0; ++i) {
    d[0] = g[0];
    g[0] = c1[-i];
    for (int j = 1; j <= m2; ++j) {
      d[j] = g[j];
      g[j] = d[j - 1] + a * (d[j] - g[j - 1]);
    }
  }
}

/* solve A x = b (n x n, row-major, destroyed) with partial pivoting */
static int solve_dense(double* A, double* b, int n) {
  for (int c = 0; c < n; ++c) {
    int p = c;
    double best = fabs(A[c * n + c]);
    for (int r = c + 1; r < n; ++r)
      if (fabs(A[r * n + c]) > best) {
        best = fabs(A[r * n + c]);
        p = r;
      }
    if (best == 0.0) return -1;
    if (p != c) {
      for (int k = 0; k < n; ++k) {
        const double t = A[c * n + k];
        A[c * n + k] = A[p * n + k];
        A[p * n + k] = t;
      }
      const double t = b[c];
      b[c] = b[p];
      b[p] = t;
    }
    for (int r = c + 1; r < n; ++r) {
      const double f = A[r * n + c] / A[c * n + c];
      if (f != 0.0) {
        for (int k = c; k < n; ++k) A[r * n + k] -= f * A[c * n + k];
        b[r] -= f * b[c];
      }
    }
  }
  for (int r = n - 1; r >= 0; --r) {
    double s = b[r];
    for (int k = r + 1; k < n; ++k) s -= A[r * n + k] * b[k];
    b[r] = s / A[r * n + r];
  }
  return 0;
}

/* amp [T, K=flng/2+1] amplitude spectra -> mc [T, m+1]; iters[T] (optional) Newton steps used */
int orc_mcep(const double* amp, int T, int K, int m, double a, double eps, int itr1, int itr2,
             double dd, double* mc_out, int* iters) {
  const int flng = (K - 1) * 2, f2 = flng / 2, m2 = 2 * m;
  double* x = (double*)malloc(sizeof(double) * (flng * 6 + (m2 + 1) * 4 + (m + 1) * (m + 1) + 64));
  if (!x) return -1;
  double* cr_ = x + flng;      /* real work */
  double* ci_ = cr_ + flng;    /* imag work */
  double* c = ci_ + flng;      /* cepstrum [flng] */
  double* dwork = c + flng;    /* freqt d buffer [f2+1] <= flng */
  double* rr = dwork + flng;   /* r [flng] */
  double* cr = rr + flng;      /* [m2+1] */
  double* al = cr + m2 + 1;    /* [m+1] */
  double* bvec = al + m2 + 1;  /* [m+1] */
  double* tmpv = bvec + m2 + 1;
  double* A = tmpv + m2 + 1;
  for (int t = 0; t < T; ++t) {
    const double* sp = amp + (size_t)t * K;
    double* mc = mc_out + (size_t)t * (m + 1);
    for (int k = 0; k <= f2; ++k) x[k] = sp[k] * sp[k] + eps;
    for (int k = 1; k < f2; ++k) x[flng - k] = x[k];
    /* c = IFFT(log x).real */
    for (int k = 0; k < flng; ++k) {
      cr_[k] = log(x[k]);
      ci_[k] = 0.0;
    }
    orc_fft(cr_, ci_, flng, +1);
    for (int k = 0; k < flng; ++k) c[k] = cr_[k] / flng;
    c[0] /= 2;
    c[f2] /= 2;
    freqt(c, f2, mc, m, a, dwork);
    double s = c[0];
    al[0] = 1.0;
    for (int i = 1; i <= m; ++i) al[i] = -a * al[i - 1];
    int j;
    for (j = 1; j <= itr2; ++j) {
      /* c' = freqt(mc, -a) to order f2, zero padded to flng */
      freqt(mc, m, cr_, f2, -a, dwork);
      for (int k = f2 + 1; k < flng; ++k) cr_[k] = 0.0;
      memset(ci_, 0, sizeof(double) * flng);
      orc_fft(cr_, ci_, flng, -1);
      for (int k = 0; k < flng; ++k) {
        cr_[k] = x[k] / exp(2.0 * cr_[k]);
        ci_[k] = 0.0;
      }
      orc_fft(cr_, ci_, flng, +1);
      for (int k = 0; k < flng; ++k) rr[k] = cr_[k] / flng;
      frqtr(rr, f2, cr, m2, a, dwork);
      const double tt = cr[0];
      if (j >= itr1) {
        if (fabs((tt - s) / tt) < dd) break;
        s = tt;
      }
      for (int i = 0; i <= m; ++i) bvec[i] = cr[i] - al[i];
      /* Toeplitz + Hankel system (mcep.c): */
      for (int i = 0; i <= m2; ++i) tmpv[i] = cr[i]; /* Hankel part h */
      for (int i = 0; i <= m2; i += 2) tmpv[i] -= cr[0];
      /* Toeplitz part t: t[i] = cr[i] (+ cr[0] for even i >= 2), t[0] = 2 cr[0] */
      for (int i = 0; i <= m; ++i) {
        for (int k = 0; k <= m; ++k) {
          const int dfk = i > k ? i - k : k - i;
          double tv = cr[dfk];
          if (dfk == 0)
            tv += cr[0];
          else if (dfk >= 2 && dfk % 2 == 0 && dfk <= m)
            tv += cr[0];
          A[i * (m + 1) + k] = tv + tmpv[i + k];
        }
      }
      if (solve_dense(A, bvec, m + 1) != 0) {
        free(x);
        return -2;
      }
      for (int i = 0; i <= m; ++i) mc[i] += bvec[i];
    }
    if (iters) iters[t] = j > itr2 ? itr2 : j;
  }
  free(x);
  return 0;
}

/* pysptk.mgc2sp(mc, alpha, gamma=0, fftlen).real: log-amplitude spectrum [T, fftlen/2+1] */
int orc_mgc2sp_logamp(const double* mc, int T, int m, double alpha, int fftlen, double* out) {
  const int f2 = fftlen / 2;
  double* cr_ = (double*)malloc(sizeof(double) * (3 * fftlen + 8));
  if (!cr_) return -1;
  double* ci_ = cr_ + fftlen;
  double* d = ci_ + fftlen;
  for (int t = 0; t < T; ++t) {
    freqt(mc + (size_t)t * (m + 1), m, cr_, f2, -alpha, d);
    for (int k = f2 + 1; k < fftlen; ++k) cr_[k] = 0.0;
    memset(ci_, 0, sizeof(double) * fftlen);
    orc_fft(cr_, ci_, fftlen, -1);
    for (int k = 0; k <= f2; ++k) out[(size_t)t * (f2 + 1) + k] = cr_[k];
  }
  free(cr_);
  return 0;
}
