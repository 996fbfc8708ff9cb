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

/* ------------------------------------------------------------------------------------------------
 * pysptk.mgcep(amp_sp, order, alpha, gamma, eps=1e-8, min_det=0, etype=1, itype=3) with pysptk's
 * defaults num_recursions = len(amp_sp) - 1, miniter 2, maxiter 30, threshold 1e-3, otype 0 --
 * AudioProcessing.extract_mgc (idiaptts/src/data_preparation/audio/AudioProcessing.py:123-140,
 * gamma = -1/3) -- and pysptk.mgc2sp for any gamma (AudioProcessing.mgc_to_amp_sp :259-275).
 * Restated from the published algorithm (Tokuda, Kobayashi, Masuko, Imai: "Mel-generalized
 * cepstral analysis", ICSLP 1994; SPTK mgcep.c follows it): the spectral model is
 *     D(w) = (1 + g C(w~))^(1/g),  C = sum_{m>=1} c(m) e^{-j w~ m},  w~ the all-pass warped frequency,
 * the criterion eps(c) = 1/2pi int x(w) / |D(w)|^2 dw is minimised by Newton steps
 *     [Toeplitz(p~) + (1+g) Hankel(q~)] dc = r~      with, per frequency bin and s = |1 + g C|^2,
 *     p = x s^(-1/g) / s,   r = p (1 + g C),   q = p (1 + g C)^2 / s
 * (p~, q~, r~: inverse transforms taken to the warped axis and to the b'-parametrisation by the
 * ptrans / qtrans recursions).  The first step is the g = -1 step from zero (an LPC solution),
 * converted to the target gamma by gc2gc; convergence is tested on log(eps).
 * PARITY UNPINNED: pysptk is not installable here and the reference holds no MGC golden vector
 * (only the loose reconstruction bound of test_WorldFeatLabelGen.py:827-836); the g -> 0 limit is
 * checked against the pinned mcep above (tests/test_oracle_golden.py).
 */

/* SPTK's static b2c() of mgcep.c: like freqt, without the `+ a d[0]` in the zeroth term */
static void mg_b2c(const double* b, int m1, double* c, int m2, double a, double* d, double* g) {
  const double k = 1.0 - a * a;
  memset(g, 0, sizeof(double) * (m2 + 1));
  memset(d, 0, sizeof(double) * (m2 + 1));
  for (int i = -m1; i <= 0; ++i) {
    d[0] = g[0];
    g[0] = b[-i];
    if (m2 >= 1) {
      d[1] = g[1];
      g[1] = k * d[0] + a * d[1];
    }
    for (int j = 2; j <= m2; ++j) {
      d[j] = g[j];
      g[j] = d[j - 1] + a * (d[j] - g[j - 1]);
    }
  }
  memcpy(c, g, sizeof(double) * (m2 + 1));
}

static void mg_ptrans(double* p, int m, double a) {
  double d = p[m], o;
  for (m--; m > 0; m--) {
    o = p[m] + a * d;
    d = p[m];
    p[m] = o;
  }
  o = a * d;
  p[m] = (1.0 - a * a) * p[m] + o + o;
}

static void mg_qtrans(double* q, int m, double a) {
  m += m;
  int i = 1;
  double d = q[i], o;
  for (i++; i <= m; i++) {
    o = q[i] + a * d;
    d = q[i];
    q[i] = o;
  }
}

static void mg_gnorm(const double* c1, double* c2, int m, double g) {
  if (g != 0.0) {
    const double k = 1.0 + g * c1[0];
    for (int i = m; i >= 1; --i) c2[i] = c1[i] / k;
    c2[0] = pow(k, 1.0 / g);
  } else {
    for (int i = m; i >= 1; --i) c2[i] = c1[i];
    c2[0] = exp(c1[0]);
  }
}

static void mg_ignorm(const double* c1, double* c2, int m, double g) {
  if (g != 0.0) {
    const double k = pow(c1[0], g);
    for (int i = m; i >= 1; --i) c2[i] = k * c1[i];
    c2[0] = (k - 1.0) / g;
  } else {
    for (int i = m; i >= 1; --i) c2[i] = c1[i];
    c2[0] = log(c1[0]);
  }
}

static void mg_b2mc(const double* b, double* mc, int m, double a) {
  double d = b[m], o;
  mc[m] = d;
  for (m--; m >= 0; m--) {
    o = b[m] + a * d;
    d = b[m];
    mc[m] = o;
  }
}

static void mg_mc2b(const double* mc, double* b, int m, double a) {
  b[m] = mc[m];
  for (m--; m >= 0; m--) b[m] = mc[m] - a * b[m + 1];
}

static void mg_gc2gc(const double* c1, int m1, double g1, double* c2, int m2, double g2, double* cin) {
  memcpy(cin, c1, sizeof(double) * (m1 + 1));
  c2[0] = cin[0];
  for (int i = 1; i <= m2; ++i) {
    double ss1 = 0.0, ss2 = 0.0;
    const int mn = (m1 < i) ? m1 : i - 1;
    for (int k = 1; k <= mn; ++k) {
      const int mk = i - k;
      const double cc = cin[k] * c2[mk];
      ss2 += k * cc;
      ss1 += mk * cc;
    }
    if (i <= m1)
      c2[i] = cin[i] + (g2 * ss2 - g1 * ss1) / i;
    else
      c2[i] = (g2 * ss2 - g1 * ss1) / i;
  }
}

typedef struct {
  int flng, m, n;
  double *cr, *ci, *pr, *pi, *qr, *qi, *rr, *ri, *d, *gw, *A, *b;
} mg_work;

/* one Newton step on c[0..m] (c[1..m]: b' coefficients, c[0]: gain); returns log(eps) */
static double mg_newton(const double* x, mg_work* w, double* c, double a, double g, int* fail) {
  const int flng = w->flng, m = w->m, n = w->n, m2 = 2 * m;
  double *cr = w->cr, *ci = w->ci, *pr = w->pr, *pi = w->pi, *qr = w->qr, *qi = w->qi, *rr = w->rr,
         *ri = w->ri;
  memset(cr, 0, sizeof(double) * flng);
  memcpy(cr + 1, c + 1, sizeof(double) * m);
  if (a != 0.0) mg_b2c(cr, m, cr, n, -a, w->d, w->gw);
  memset(ci, 0, sizeof(double) * flng);
  orc_fft(cr, ci, flng, -1);            /* cr + j ci = FFT[c] */
  for (int i = 0; i < flng; ++i) {
    if (g == -1.0) {
      pr[i] = x[i];
    } else if (g == 0.0) {
      pr[i] = x[i] / exp(cr[i] + cr[i]);
    } else {
      const double tr = 1.0 + g * cr[i], ti = g * ci[i];
      const double s = tr * tr + ti * ti;
      double t = x[i] * pow(s, -1.0 / g);
      t /= s;
      pr[i] = t;
      rr[i] = tr * t;
      ri[i] = ti * t;
      t /= s;
      qr[i] = (tr * tr - ti * ti) * t;
      qi[i] = 2.0 * tr * ti * t;
    }
  }
  memset(pi, 0, sizeof(double) * flng);
  orc_fft(pr, pi, flng, +1);
  for (int i = 0; i < flng; ++i) pr[i] /= flng;
  if (a != 0.0) mg_b2c(pr, n, pr, m2, a, w->d, w->gw);
  if (g == 0.0 || g == -1.0) {
    memcpy(qr, pr, sizeof(double) * (m2 + 1));
    memcpy(rr, pr, sizeof(double) * (m + 1));
  } else {
    orc_fft(qr, qi, flng, +1);
    orc_fft(rr, ri, flng, +1);
    for (int i = 0; i < flng; ++i) {
      qr[i] /= flng;
      rr[i] /= flng;
    }
    if (a != 0.0) {
      mg_b2c(qr, n, qr, n, a, w->d, w->gw);
      mg_b2c(rr, n, rr, m, a, w->d, w->gw);
    }
  }
  if (a != 0.0) {
    mg_ptrans(pr, m, a);
    mg_qtrans(qr, m, a);
  }
  /* gain: eps = r(0) + g sum r(i) c(i) */
  double t = rr[0];
  if (g != 0.0)
    for (int i = 1; i <= m; ++i) t += g * rr[i] * c[i];
  c[0] = sqrt(t);
  if (g == -1.0)
    memset(qr, 0, sizeof(double) * (m2 + 1));
  else if (g != 0.0)
    for (int i = 2; i <= m2; ++i) qr[i] *= 1.0 + g;
  /* theq(pr, &qr[2], &b[1], &rr[1], m): (T + H) b = r with T[i][j] = pr[|i-j|], H[i][j] = qr[2+i+j] */
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < m; ++j) w->A[i * m + j] = pr[i > j ? i - j : j - i] + qr[2 + i + j];
    w->b[i] = rr[1 + i];
  }
  if (solve_dense(w->A, w->b, m) != 0) {
    *fail = 1;
    return 0.0;
  }
  for (int i = 1; i <= m; ++i) c[i] += w->b[i - 1];
  return log(t);
}

int orc_mgcep(const double* amp, int T, int K, int m, double a, double g, double eps, int itr1,
              int itr2, double dd, double* mgc_out, int* iters) {
  const int flng = (K - 1) * 2, f2 = flng / 2, n = K - 1;
  mg_work w;
  w.flng = flng; w.m = m; w.n = n;
  double* buf = (double*)malloc(sizeof(double) * ((size_t)flng * 11 + (size_t)m * m + 4 * (m + 2) + 64));
  if (!buf) return -1;
  double* x = buf;
  w.cr = x + flng; w.ci = w.cr + flng; w.pr = w.ci + flng; w.pi = w.pr + flng; w.qr = w.pi + flng;
  w.qi = w.qr + flng; w.rr = w.qi + flng; w.ri = w.rr + flng; w.d = w.ri + flng; w.gw = w.d + flng;
  w.A = w.gw + flng; w.b = w.A + (size_t)m * m;
  double* dv = w.b + m + 2;
  double* cin = dv + m + 2;
  int rc = 0;
  for (int t = 0; t < T && rc == 0; ++t) {
    const double* sp = amp + (size_t)t * K;
    double* b = mgc_out + (size_t)t * (m + 1);
    for (int k = 0; k <= f2; ++k) x[k] = sp[k] * sp[k] + eps;
    for (int k = 1; k < f2; ++k) x[flng - k] = x[k];
    memset(b, 0, sizeof(double) * (m + 1));
    int fail = 0;
    double ep = mg_newton(x, &w, b, a, -1.0, &fail);
    int j = 0, conv = 0;
    if (g != -1.0 && !fail) {
      if (a != 0.0) {
        mg_ignorm(b, b, m, -1.0);
        mg_b2mc(b, b, m, a);
        mg_gnorm(b, dv, m, -1.0);
      } else {
        memcpy(dv, b, sizeof(double) * (m + 1));
      }
      mg_gc2gc(dv, m, -1.0, b, m, g, cin);
      if (a != 0.0) {
        mg_ignorm(b, b, m, g);
        mg_mc2b(b, b, m, a);
        mg_gnorm(b, b, m, g);
      }
      for (j = 1; j <= itr2 && !fail; ++j) {
        const double epo = ep;
        ep = mg_newton(x, &w, b, a, g, &fail);
        if (j >= itr1 && fabs((epo - ep) / ep) < dd) {
          conv = 1;
          break;
        }
      }
    }
    if (fail) rc = -2;
    (void)conv;
    /* pysptk otype 0: ignorm, then b2mc */
    mg_ignorm(b, b, m, g);
    if (a != 0.0) mg_b2mc(b, b, m, a);
    if (iters) iters[t] = j > itr2 ? itr2 : j;
  }
  free(buf);
  return rc;
}

/* pysptk.mgc2sp(mgc, alpha, gamma, fftlen).real: mgc2mgc to (order fftlen/2, alpha 0, gamma 0),
 * then the real part of the FFT of the cepstrum (log amplitude) */
int orc_mgc2sp_gamma(const double* mgc, int T, int m, double alpha, double gamma, int fftlen,
                     double* out) {
  const int f2 = fftlen / 2;
  double* c = (double*)malloc(sizeof(double) * (5 * fftlen + 16));
  if (!c) return -1;
  double* ci_ = c + fftlen;
  double* d = ci_ + fftlen;
  double* cin = d + fftlen;
  double* c2 = cin + fftlen;
  for (int t = 0; t < T; ++t) {
    const double* in = mgc + (size_t)t * (m + 1);
    if (alpha == 0.0) {
      mg_gnorm(in, c2, m, gamma);
      mg_gc2gc(c2, m, gamma, c, f2, 0.0, cin);
      mg_ignorm(c, c, f2, 0.0);
    } else {
      freqt(in, m, c, f2, -alpha, d);          /* a = (0 - alpha) / (1 - 0) */
      mg_gnorm(c, c, f2, gamma);
      memcpy(c2, c, sizeof(double) * (f2 + 1));
      mg_gc2gc(c2, f2, gamma, c, f2, 0.0, cin);
      mg_ignorm(c, c, f2, 0.0);
    }
    for (int k = f2 + 1; k < fftlen; ++k) c[k] = 0.0;
    memset(ci_, 0, sizeof(double) * fftlen);
    orc_fft(c, ci_, fftlen, -1);
    for (int k = 0; k <= f2; ++k) out[(size_t)t * (f2 + 1) + k] = c[k];
  }
  free(c);
  return 0;
}
