/* ORACLE (test infrastructure; never linked into or called by the product path).
 *
 * CPU (fp64) restatement of the WORLD analysis chain the reference reaches through
 * pyworld.wav2world / pyworld.code_aperiodicity
 * (idiaptts/src/data_preparation/world/WorldFeatLabelGen.py:792-793, :805).
 * pyworld (unpinned, requirements.txt:6; a Cython wrapper of mmorise/World) is NOT in
 * /root/reference and cannot be installed, so this restates the published algorithms:
 *   Dio (dio.cpp), StoneMask (stonemask.cpp, two-stage variant), CheapTrick (cheaptrick.cpp),
 *   D4C + LoveTrain (d4c.cpp), CodeAperiodicity (codec.cpp), common helpers (common.cpp,
 *   matlabfunctions.cpp).  CheapTrick carries WORLD's two safeguard noise terms (1e-12 * randn on
 *   the windowed waveform, eps * |randn| on the smoothed spectrum, one stream per call) -- they
 *   decide the envelope of digitally silent frames; D4C's 1e-12 window noise is omitted (it only
 *   acts on voiced frames, where it is below f32 resolution of every output).
 * Pinned against the reference's golden fixtures test/integration/fixtures/WORLD/cmp_mcep20/
 * (tests/test_oracle_golden.py; settings: pre-emphasis 0.97, frame period 5 ms).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS 1e-12      /* WORLD kMySafeGuardMinimum */
#define KMAX 100000.0  /* WORLD kMaximumValue */

int orc_mround(double x) { return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5); }
static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }
static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }

/* ---- matlabfunctions.cpp ----------------------------------------------------------------- */
/* interp1Q: equally spaced abscissa starting at x0 with step `shift` */
static void interp1Q(double x0, double shift, const double* y, int ylen, const double* xi, int n,
                     double* out) {
  for (int i = 0; i < n; ++i) {
    const double pos = (xi[i] - x0) / shift;
    const int base = (int)pos;
    const double frac = pos - base;
    const double dy = (base + 1 < ylen) ? y[base + 1] - y[base] : 0.0;
    out[i] = y[base] + dy * frac;
  }
}

static void histc(const double* x, int n, const double* edges, int m, int* index) {
  int count = 1, i = 0;
  for (; i < m; ++i) {
    index[i] = 1;
    if (edges[i] >= x[0]) break;
  }
  for (; i < m; ++i) {
    if (edges[i] < x[count]) {
      index[i] = count;
    } else {
      index[i--] = count++;
    }
    if (count == n) break;
  }
  count--;
  for (i++; i < m; ++i) index[i] = count;
}

/* linear interpolation that EXTRAPOLATES with the first / last segment */
void orc_interp1(const double* x, const double* y, int n, const double* xi, int m, double* yi) {
  int* k = (int*)malloc(sizeof(int) * (m > 0 ? m : 1));
  histc(x, n, xi, m, k);
  for (int i = 0; i < m; ++i) {
    const int j = k[i] - 1;
    const double h = x[j + 1] - x[j];
    const double s = (xi[i] - x[j]) / h;
    yi[i] = y[j] + s * (y[j + 1] - y[j]);
  }
  free(k);
}

void orc_nuttall(int n, double* w) {
  for (int i = 0; i < n; ++i) {
    const double t = (double)i / (n - 1.0);
    w[i] = 0.355768 - 0.487396 * cos(2.0 * M_PI * t) + 0.144232 * cos(4.0 * M_PI * t) -
           0.012604 * cos(6.0 * M_PI * t);
  }
}

/* ---- common.cpp ----------------------------------------------------------------------------- */
static void dc_correction(double* P, double f0, int fs, int fft) {
  const int upper = 2 + (int)(f0 * fft / fs);
  double* lfa = (double*)malloc(sizeof(double) * upper * 2);
  double* rep = lfa + upper;
  for (int i = 0; i < upper; ++i) lfa[i] = (double)i * fs / fft;
  interp1Q(f0 - lfa[0], -(double)fs / fft, P, upper + 1, lfa, upper - 1, rep);
  for (int i = 0; i < upper - 1; ++i) P[i] += rep[i];
  free(lfa);
}

/* out may alias P */
static void linear_smoothing(const double* P, double width, int fs, int fft, double* out) {
  const int boundary = (int)(width * fft / fs) + 1;
  const int h = fft / 2;
  const int ml = h + boundary * 2 + 1;
  double* mir = (double*)malloc(sizeof(double) * (ml + 2 * (h + 1)));
  double* fa = mir + ml;
  double* lo = fa + h + 1;
  for (int i = 0; i < boundary; ++i) mir[i] = P[boundary - i];
  for (int i = boundary; i < h + boundary; ++i) mir[i] = P[i - boundary];
  for (int i = h + boundary; i <= h + boundary * 2; ++i) mir[i] = P[h - (i - (h + boundary))];
  mir[0] = mir[0] * fs / fft;
  for (int i = 1; i < ml; ++i) mir[i] = mir[i] * fs / fft + mir[i - 1];
  for (int i = 0; i <= h; ++i) fa[i] = (double)i / fft * fs - width / 2.0;
  const double org = -((double)boundary - 0.5) * fs / fft;
  const double dfi = (double)fs / fft;
  interp1Q(org, dfi, mir, ml, fa, h + 1, lo);
  for (int i = 0; i <= h; ++i) fa[i] += width;
  double* hi = (double*)malloc(sizeof(double) * (h + 1));
  interp1Q(org, dfi, mir, ml, fa, h + 1, hi);
  for (int i = 0; i <= h; ++i) out[i] = (hi[i] - lo[i]) / width;
  free(hi);
  free(mir);
}

/* ---- DIO (dio.cpp) ------------------------------------------------------------------------- */
typedef struct {
  int n;          /* number of intervals */
  double* loc;    /* interval locations [s] */
  double* f0;     /* interval based f0 */
} events_t;

static void zero_crossing_engine(const double* s, int n, double fs, events_t* ev, double* fine) {
  int count = 0;
  for (int i = 0; i < n - 1; ++i) {
    if (s[i] > 0.0 && s[i + 1] <= 0.0) {
      const int e = i + 1; /* numpy spec: ng = i + 1 (0-based), fine = ng - s[ng-1]/(s[ng]-s[ng-1]) */
      fine[count++] = (double)e - s[e - 1] / (s[e] - s[e - 1]);
    }
  }
  if (count < 2) {
    ev->n = -1;
    return;
  }
  ev->n = count - 1;
  for (int i = 0; i < count - 1; ++i) {
    ev->loc[i] = (fine[i] + fine[i + 1]) / 2.0 / fs;
    ev->f0[i] = fs / (fine[i + 1] - fine[i]);
  }
}

static double select_best(double cur, double past, const double* cands, int nb, int T, int ti,
                          double ar) {
  const double ref = (cur * 3.0 - past) / 2.0;
  double me = fabs(ref - cands[0 * T + ti]);
  double bf = cands[0 * T + ti];
  for (int i = 1; i < nb; ++i) {
    const double ce = fabs(ref - cands[i * T + ti]);
    if (ce < me) {
      me = ce;
      bf = cands[i * T + ti];
    }
  }
  if (fabs(1.0 - bf / ref) > ar) return 0.0;
  return bf;
}

static void fix_f0_contour(double fp, int nb, const double* cands, const double* best, int T,
                           double f0_floor, double ar, double* out) {
  const int vrm = (int)(0.5 + 1000.0 / fp / f0_floor) * 2 + 1;
  memset(out, 0, sizeof(double) * T);
  if (T <= vrm) return;
  double* base = (double*)calloc((size_t)T * 4, sizeof(double));
  double *s1 = base + T, *s2 = s1 + T, *s3 = s2 + T;
  int* pos = (int*)malloc(sizeof(int) * 2 * T);
  int* neg = pos + T;
  for (int i = vrm; i < T - vrm; ++i) base[i] = best[i];
  for (int i = vrm; i < T; ++i)
    s1[i] = fabs((base[i] - base[i - 1]) / (EPS + base[i])) < ar ? base[i] : 0.0;
  memcpy(s2, s1, sizeof(double) * T);
  const int c = (vrm - 1) / 2;
  for (int i = c; i < T - c; ++i) {
    for (int j = -c; j <= c; ++j)
      if (s1[i + j] == 0.0) {
        s2[i] = 0.0;
        break;
      }
  }
  int np = 0, nn = 0;
  for (int i = 1; i < T; ++i) {
    if (s2[i] == 0.0 && s2[i - 1] != 0.0)
      neg[nn++] = i - 1;
    else if (s2[i - 1] == 0.0 && s2[i] != 0.0)
      pos[np++] = i;
  }
  memcpy(s3, s2, sizeof(double) * T);
  for (int i = 0; i < nn; ++i) {
    const int limit = (i == nn - 1) ? T - 1 : neg[i + 1];
    for (int j = neg[i]; j < limit; ++j) {
      s3[j + 1] = select_best(s3[j], s3[j - 1], cands, nb, T, j + 1, ar);
      if (s3[j + 1] == 0.0) break;
    }
  }
  memcpy(out, s3, sizeof(double) * T);
  for (int i = np - 1; i >= 0; --i) {
    const int limit = (i == 0) ? 1 : pos[i - 1];
    for (int j = pos[i]; j > limit; --j) {
      out[j - 1] = select_best(out[j], out[j + 1], cands, nb, T, j - 1, ar);
      if (out[j - 1] == 0.0) break;
    }
  }
  free(pos);
  free(base);
}

/* pyworld.dio defaults: f0_floor 71, f0_ceil 800, channels_in_octave 2, speed 1,
 * allowed_range 0.1 */
int orc_dio(const double* x, int xl, int fs, double frame_period, double f0_floor, double f0_ceil,
            double channels_in_octave, double allowed_range, double* f0_out, double* tp_out) {
  const int nb = 1 + (int)(log(f0_ceil / f0_floor) / log(2.0) * channels_in_octave);
  double* bnd = (double*)malloc(sizeof(double) * nb);
  for (int i = 0; i < nb; ++i) bnd[i] = f0_floor * pow(2.0, (i + 1) / channels_in_octave);
  const int yl = 1 + xl;
  const double afs = (double)fs;
  const int fft =
      1 << ((int)(log2(yl + orc_mround(afs / 50.0) * 2 + 1 + 4 * (int)(1.0 + afs / bnd[0] / 2.0))) + 1);
  const int T = (int)(1000.0 * xl / fs / frame_period) + 1;
  const int hb = fft / 2 + 1;
  double* y = (double*)calloc(fft, sizeof(double));
  double* f = (double*)calloc(fft, sizeof(double));
  double* work = (double*)malloc(sizeof(double) * fft);
  double* Yr = (double*)malloc(sizeof(double) * hb * 6);
  double *Yi = Yr + hb, *Fr = Yi + hb, *Fi = Fr + hb, *Zr = Fi + hb, *Zi = Zr + hb;
  double* sig = (double*)malloc(sizeof(double) * fft * 2);
  double* d = sig + fft;
  double* cands = (double*)calloc((size_t)nb * T * 2, sizeof(double));
  double* scores = cands + (size_t)nb * T;
  double* sets = (double*)malloc(sizeof(double) * 4 * T);
  double* fine = (double*)malloc(sizeof(double) * (yl + 1));
  events_t ev[4];
  for (int k = 0; k < 4; ++k) {
    ev[k].loc = (double*)malloc(sizeof(double) * (yl + 1));
    ev[k].f0 = (double*)malloc(sizeof(double) * (yl + 1));
  }
  if (!y || !f || !work || !Yr || !sig || !cands || !sets || !fine) return -1;

  /* DC removal over yl = xl + 1 samples */
  memcpy(y, x, sizeof(double) * xl);
  double mean = 0.0;
  for (int i = 0; i < yl; ++i) mean += y[i];
  mean /= yl;
  for (int i = 0; i < yl; ++i) y[i] -= mean;
  /* 50 Hz low-cut filter (unit impulse minus normalised Hann), centred at 0 */
  const int N = orc_mround(afs / 50.0) * 2 + 1;
  double wsum = 0.0;
  for (int i = 1; i <= N; ++i) {
    f[i - 1] = 0.5 - 0.5 * cos(i * 2.0 * M_PI / (N + 1));
    wsum += f[i - 1];
  }
  for (int i = 0; i < N; ++i) f[i] = -f[i] / wsum;
  const int half = (N - 1) / 2;
  for (int k = 0; k < half; ++k) f[fft - half + k] = f[k];
  for (int k = 0; k < N; ++k) f[k] = f[k + half];
  f[0] += 1.0;
  orc_rfft(y, fft, Yr, Yi, work);
  orc_rfft(f, fft, Fr, Fi, work);
  for (int k = 0; k < hb; ++k) {
    const double r = Yr[k] * Fr[k] - Yi[k] * Fi[k];
    const double im = Yr[k] * Fi[k] + Yi[k] * Fr[k];
    Yr[k] = r;
    Yi[k] = im;
  }
  for (int i = 0; i < T; ++i) tp_out[i] = i * frame_period / 1000.0;

  for (int b = 0; b < nb; ++b) {
    const int hal = orc_mround(afs / bnd[b] / 2.0);
    memset(f, 0, sizeof(double) * fft);
    orc_nuttall(hal * 4, f);
    orc_rfft(f, fft, Fr, Fi, work);
    for (int k = 0; k < hb; ++k) {
      Zr[k] = Yr[k] * Fr[k] - Yi[k] * Fi[k];
      Zi[k] = Yr[k] * Fi[k] + Yi[k] * Fr[k];
    }
    orc_irfft(Zr, Zi, fft, y, work); /* y reused as the filtered signal */
    for (int i = 0; i < yl; ++i) sig[i] = y[hal * 2 + i];
    zero_crossing_engine(sig, yl, afs, &ev[0], fine);
    for (int i = 0; i < yl; ++i) sig[i] = -sig[i];
    zero_crossing_engine(sig, yl, afs, &ev[1], fine);
    for (int i = 0; i < yl - 1; ++i) d[i] = sig[i] - sig[i + 1];
    zero_crossing_engine(d, yl - 1, afs, &ev[2], fine);
    for (int i = 0; i < yl - 1; ++i) d[i] = -d[i];
    zero_crossing_engine(d, yl - 1, afs, &ev[3], fine);
    double* cb = cands + (size_t)b * T;
    double* sb = scores + (size_t)b * T;
    int ok = 1;
    for (int k = 0; k < 4; ++k)
      if (ev[k].n < 0 || ev[k].n - 2 <= 0) ok = 0;
    if (!ok) {
      for (int i = 0; i < T; ++i) {
        cb[i] = 0.0;
        sb[i] = KMAX;
      }
    } else {
      for (int k = 0; k < 4; ++k) orc_interp1(ev[k].loc, ev[k].f0, ev[k].n, tp_out, T, sets + k * T);
      for (int i = 0; i < T; ++i) {
        const double c = (sets[i] + sets[T + i] + sets[2 * T + i] + sets[3 * T + i]) / 4.0;
        double sc = 0.0;
        for (int k = 0; k < 4; ++k) sc += (sets[k * T + i] - c) * (sets[k * T + i] - c);
        sc = sqrt(sc / 3.0);
        if (c > bnd[b] || c < bnd[b] / 2.0 || c > f0_ceil || c < f0_floor) {
          cb[i] = 0.0;
          sb[i] = KMAX;
        } else {
          cb[i] = c;
          sb[i] = sc;
        }
      }
    }
    for (int i = 0; i < T; ++i) sb[i] = sb[i] / (cb[i] + EPS);
  }
  double* best = (double*)malloc(sizeof(double) * T);
  for (int i = 0; i < T; ++i) {
    double t = scores[i];
    best[i] = cands[i];
    for (int j = 1; j < nb; ++j)
      if (t > scores[(size_t)j * T + i]) {
        t = scores[(size_t)j * T + i];
        best[i] = cands[(size_t)j * T + i];
      }
  }
  fix_f0_contour(frame_period, nb, cands, best, T, f0_floor, allowed_range, f0_out);
  free(best);
  for (int k = 0; k < 4; ++k) {
    free(ev[k].loc);
    free(ev[k].f0);
  }
  free(fine); free(sets); free(cands); free(sig); free(Yr); free(work); free(f); free(y); free(bnd);
  return 0;
}

/* ---- StoneMask (two stages: 2 harmonics, then min(6, fs/2/f0)) -------------------------------- */
static double fixf0(const double* ps, const double* num, int fft, int fs, double f0, int nh) {
  double a = 0.0, b = 0.0;
  for (int i = 0; i < nh; ++i) {
    const int idx = orc_mround(f0 * fft / fs * (i + 1));
    const double inst =
        ps[idx] == 0.0 ? 0.0 : (double)idx * fs / fft + num[idx] / ps[idx] * fs / 2.0 / M_PI;
    const double amp = sqrt(ps[idx]);
    a += amp * inst;
    b += amp * (i + 1);
  }
  return a / (b + EPS);
}

static double stonemask_frame(const double* x, int xl, int fs, double pos, double f0, double* buf) {
  if (f0 <= 40.0 || f0 > fs / 12.0) return 0.0;
  const int half = (int)(1.5 * fs / f0 + 1.0);
  const double wlt = (2.0 * half + 1.0) / fs;
  const int n = 2 * half + 1;
  const int fft = 1 << (2 + (int)(log(half * 2.0 + 1.0) / log(2.0)));
  const int hb = fft / 2 + 1;
  /* buf layout: mw[n] dw[n] seg[fft] Mr Mi Dr Di [hb each] work[fft] */
  double* mw = buf;
  double* dw = mw + n;
  double* seg = dw + n;
  double* Mr = seg + fft;
  double* Mi = Mr + hb;
  double* Dr = Mi + hb;
  double* Di = Dr + hb;
  double* work = Di + hb;
  const int i0 = orc_mround((pos - (double)half / fs) * fs + 0.001);
  for (int i = 0; i < n; ++i) {
    const double tmp = ((double)(i0 + i) - 1.0) / fs - pos;
    mw[i] = 0.42 + 0.5 * cos(2.0 * M_PI * tmp / wlt) + 0.08 * cos(4.0 * M_PI * tmp / wlt);
  }
  dw[0] = -mw[1] / 2.0;
  for (int i = 1; i < n - 1; ++i) dw[i] = -(mw[i + 1] - mw[i - 1]) / 2.0;
  dw[n - 1] = mw[n - 2] / 2.0;
  memset(seg, 0, sizeof(double) * fft);
  for (int i = 0; i < n; ++i) seg[i] = x[imax(0, imin(xl - 1, i0 + i - 1))] * mw[i];
  orc_rfft(seg, fft, Mr, Mi, work);
  memset(seg, 0, sizeof(double) * fft);
  for (int i = 0; i < n; ++i) seg[i] = x[imax(0, imin(xl - 1, i0 + i - 1))] * dw[i];
  orc_rfft(seg, fft, Dr, Di, work);
  /* reuse Dr as numerator, Mr as power spectrum */
  for (int k = 0; k < hb; ++k) {
    const double num = Mr[k] * Di[k] - Mi[k] * Dr[k];
    const double ps = Mr[k] * Mr[k] + Mi[k] * Mi[k];
    Dr[k] = num;
    Mr[k] = ps;
  }
  const double t = fixf0(Mr, Dr, fft, fs, f0, 2);
  double mean;
  if (t <= 0.0 || t > f0 * 2)
    mean = 0.0;
  else
    mean = fixf0(Mr, Dr, fft, fs, t, imin((int)(fs / 2.0 / f0), 6));
  return fabs(mean - f0) > f0 * 0.2 ? f0 : mean;
}

int orc_stonemask(const double* x, int xl, int fs, const double* tp, const double* f0, int T,
                  double* out) {
  /* worst case f0 just above 40 Hz */
  const int half = (int)(1.5 * fs / 40.0 + 1.0);
  const int fft = 1 << (2 + (int)(log(half * 2.0 + 1.0) / log(2.0)));
  double* buf = (double*)malloc(sizeof(double) * (2 * (2 * half + 1) + 2 * fft + 4 * (fft / 2 + 1)));
  if (!buf) return -1;
  for (int i = 0; i < T; ++i) out[i] = stonemask_frame(x, xl, fs, tp[i], f0[i], buf);
  free(buf);
  return 0;
}

/* ---- CheapTrick -------------------------------------------------------------------------------- */
static void cheaptrick_frame(const double* x, int xl, int fs, double f0, double pos, int fft,
                             double q1, double* out, double* buf, rng_t* rng) {
  const int half = orc_mround(1.5 * fs / f0);
  const int n = 2 * half + 1;
  const int h = fft / 2;
  /* buf: win[n] wf[fft] Sr Si [h+1] P[h+1] work[fft] lp[fft] */
  double* win = buf;
  double* wf = win + n;
  double* Sr = wf + fft;
  double* Si = Sr + h + 1;
  double* P = Si + h + 1;
  double* work = P + h + 1;
  double* lp = work + fft;
  const int c = orc_mround(pos * fs + 0.001);
  double e = 0.0;
  for (int i = 0; i < n; ++i) {
    const int b = i - half;
    win[i] = 0.5 * cos(M_PI * ((double)b / 1.5 / fs) * f0) + 0.5;
    e += win[i] * win[i];
  }
  e = sqrt(e);
  for (int i = 0; i < n; ++i) win[i] /= e;
  memset(wf, 0, sizeof(double) * fft);
  double swf = 0.0, sw = 0.0;
  for (int i = 0; i < n; ++i) {
    const int idx = imin(xl - 1, imax(0, c + i - half));
    wf[i] = x[idx] * win[i] + rng_randn(rng) * 1e-12;   /* cheaptrick.cpp GetWindowedWaveform */
    swf += wf[i];
    sw += win[i];
  }
  const double m = swf / sw;
  for (int i = 0; i < n; ++i) wf[i] -= win[i] * m;
  orc_rfft(wf, fft, Sr, Si, work);
  for (int k = 0; k <= h; ++k) P[k] = Sr[k] * Sr[k] + Si[k] * Si[k];
  dc_correction(P, f0, fs, fft);
  linear_smoothing(P, f0 * 2.0 / 3.0, fs, fft, P);
  /* cheaptrick.cpp AddInfinitesimalNoise (kEps = 2^-52) */
  for (int k = 0; k <= h; ++k) P[k] = P[k] + fabs(rng_randn(rng)) * 2.2204460492503131e-16;
  /* smoothing with recovery: cepstral liftering */
  for (int k = 0; k <= h; ++k) lp[k] = log(P[k]);
  for (int k = 1; k < h; ++k) lp[fft - k] = lp[k];
  orc_rfft(lp, fft, Sr, Si, work);
  for (int k = 0; k <= h; ++k) {
    double sl = 1.0, cl = 1.0;
    if (k > 0) {
      const double q = (double)k / fs;
      sl = sin(M_PI * f0 * q) / (M_PI * f0 * q);
      cl = (1.0 - 2.0 * q1) + 2.0 * q1 * cos(2.0 * M_PI * q * f0);
    }
    Sr[k] = Sr[k] * sl * cl;
    Si[k] = 0.0;
  }
  orc_irfft(Sr, Si, fft, lp, work);
  for (int k = 0; k <= h; ++k) out[k] = exp(lp[k]);
}

int orc_cheaptrick(const double* x, int xl, int fs, const double* tp, const double* f0, int T,
                   int fft, double q1, double* sp) {
  const double floor_f0 = 3.0 * fs / (fft - 3.0);
  const int maxhalf = orc_mround(1.5 * fs / floor_f0) + 2;
  double* buf = (double*)malloc(sizeof(double) * ((2 * maxhalf + 1) + 4 * fft + 3 * (fft / 2 + 1)));
  if (!buf) return -1;
  rng_t rng;
  rng_seed(&rng);                                       /* randn_reseed() at the top of CheapTrick() */
  for (int i = 0; i < T; ++i) {
    const double cf0 = f0[i] > floor_f0 ? f0[i] : 500.0;
    cheaptrick_frame(x, xl, fs, cf0, tp[i], fft, q1, sp + (size_t)i * (fft / 2 + 1), buf, &rng);
  }
  free(buf);
  return 0;
}

/* ---- D4C + LoveTrain --------------------------------------------------------------------------- */
/* windowed waveform of half-length mround(ratio*fs/f0/2); returns its length (2*half+1) */
static int windowed(const double* x, int xl, int fs, double f0, double pos, int blackman,
                    double ratio, double* wf, double* win) {
  const int half = orc_mround(ratio * fs / f0 / 2.0);
  const int n = 2 * half + 1;
  const int c = orc_mround(pos * fs + 0.001);
  double swf = 0.0, sw = 0.0;
  for (int i = 0; i < n; ++i) {
    const int b = i - half;
    const double p = (2.0 * b / ratio) / fs;
    if (blackman)
      win[i] = 0.42 + 0.5 * cos(M_PI * p * f0) + 0.08 * cos(M_PI * p * f0 * 2);
    else
      win[i] = 0.5 * cos(M_PI * p * f0) + 0.5;
    const int idx = imin(xl - 1, imax(0, c + b));
    wf[i] = x[idx] * win[i];
    swf += wf[i];
    sw += win[i];
  }
  const double m = swf / sw;
  for (int i = 0; i < n; ++i) wf[i] -= win[i] * m;
  return n;
}

typedef struct {
  double *wf, *win, *buf, *S1r, *S1i, *S2r, *S2i, *work, *sc, *sps, *sgd, *tmp;
} d4c_ws;

static void centroid(const double* x, int xl, int fs, double f0, int fft, double pos, d4c_ws* w,
                     double* out) {
  const int h = fft / 2;
  windowed(x, xl, fs, f0, pos, 1, 4.0, w->wf, w->win);
  const int n = orc_mround(2.0 * fs / f0) * 2 + 1;
  memset(w->buf, 0, sizeof(double) * fft);
  memcpy(w->buf, w->wf, sizeof(double) * n);
  double p = 0.0;
  for (int i = 0; i < n; ++i) p += w->buf[i] * w->buf[i];
  p = sqrt(p);
  for (int i = 0; i < n; ++i) w->buf[i] /= p;
  orc_rfft(w->buf, fft, w->S1r, w->S1i, w->work);
  for (int i = 0; i < fft; ++i) w->buf[i] *= (i + 1.0);
  orc_rfft(w->buf, fft, w->S2r, w->S2i, w->work);
  for (int k = 0; k <= h; ++k) out[k] = w->S2r[k] * w->S1r[k] + w->S1i[k] * w->S2i[k];
}

static int cmp_double(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

static void d4c_frame(const double* x, int xl, int fs, double f0, double pos, int fftd, int nap,
                      const double* window, int wl, d4c_ws* w, double* coarse) {
  const int h = fftd / 2;
  centroid(x, xl, fs, f0, fftd, pos - 0.25 / f0, w, w->sc);
  centroid(x, xl, fs, f0, fftd, pos + 0.25 / f0, w, w->tmp);
  for (int k = 0; k <= h; ++k) w->sc[k] += w->tmp[k];
  dc_correction(w->sc, f0, fs, fftd);
  const int n = windowed(x, xl, fs, f0, pos, 0, 4.0, w->wf, w->win);
  memset(w->buf, 0, sizeof(double) * fftd);
  memcpy(w->buf, w->wf, sizeof(double) * n);
  orc_rfft(w->buf, fftd, w->S1r, w->S1i, w->work);
  for (int k = 0; k <= h; ++k) w->sps[k] = w->S1r[k] * w->S1r[k] + w->S1i[k] * w->S1i[k];
  dc_correction(w->sps, f0, fs, fftd);
  linear_smoothing(w->sps, f0, fs, fftd, w->sps);
  for (int k = 0; k <= h; ++k) w->sgd[k] = w->sc[k] / w->sps[k];
  linear_smoothing(w->sgd, f0 / 2.0, fs, fftd, w->sgd);
  linear_smoothing(w->sgd, f0, fs, fftd, w->tmp);
  for (int k = 0; k <= h; ++k) w->sgd[k] -= w->tmp[k];
  const int boundary = orc_mround(fftd * 8.0 / wl);
  const int half = wl / 2;
  for (int i = 0; i < nap; ++i) {
    const int center = (int)(3000.0 * (i + 1) * fftd / fs);
    memset(w->buf, 0, sizeof(double) * fftd);
    for (int j = 0; j <= half * 2; ++j) w->buf[j] = w->sgd[center - half + j] * window[j];
    orc_rfft(w->buf, fftd, w->S1r, w->S1i, w->work);
    for (int k = 0; k <= h; ++k) w->tmp[k] = w->S1r[k] * w->S1r[k] + w->S1i[k] * w->S1i[k];
    qsort(w->tmp, h + 1, sizeof(double), cmp_double);
    for (int k = 1; k <= h; ++k) w->tmp[k] += w->tmp[k - 1];
    const double ca = 10.0 * log10(w->tmp[h - boundary - 1] / w->tmp[h]);
    coarse[i] = dmin(0.0, ca + (f0 - 100.0) / 50.0);
  }
}

static double lovetrain(const double* x, int xl, int fs, double f0, double pos, int fft, d4c_ws* w) {
  const int b0 = (int)ceil(100.0 * fft / fs);
  const int b1 = (int)ceil(4000.0 * fft / fs);
  const int b2 = (int)ceil(7900.0 * fft / fs);
  const int n = windowed(x, xl, fs, dmax(f0, 40.0), pos, 1, 3.0, w->wf, w->win);
  memset(w->buf, 0, sizeof(double) * fft);
  memcpy(w->buf, w->wf, sizeof(double) * n);
  orc_rfft(w->buf, fft, w->S1r, w->S1i, w->work);
  double* ps = w->tmp;
  for (int k = 0; k <= b2; ++k) ps[k] = w->S1r[k] * w->S1r[k] + w->S1i[k] * w->S1i[k];
  for (int k = 0; k <= b0; ++k) ps[k] = 0.0;
  for (int k = 1; k <= b2; ++k) ps[k] += ps[k - 1];
  return ps[b1] / ps[b2];
}

/* ap [T, fft_size/2+1] */
int orc_d4c(const double* x, int xl, int fs, const double* tp, const double* f0, int T,
            int fft_size, double threshold, double* ap) {
  const int K = fft_size / 2 + 1;
  const int fftd = 1 << (1 + (int)log2(4.0 * fs / 47.0 + 1));
  const int fftl = 1 << (1 + (int)log2(3.0 * fs / 40.0 + 1));
  const int fmax = imax(fftd, fftl);
  const int nap = (int)(dmin(15000.0, fs / 2.0 - 3000.0) / 3000.0);
  const int wl = (int)(3000.0 * fftd / fs) * 2 + 1;
  double* window = (double*)malloc(sizeof(double) * wl);
  orc_nuttall(wl, window);
  const int maxn = orc_mround(4.0 * fs / 40.0 / 2.0) * 2 + 8;
  d4c_ws w;
  double* pool = (double*)malloc(sizeof(double) * (2 * maxn + 2 * fmax + 8 * (fmax / 2 + 1)));
  if (!pool) return -1;
  double* p = pool;
  w.wf = p; p += maxn;
  w.win = p; p += maxn;
  w.buf = p; p += fmax;
  w.work = p; p += fmax;
  w.S1r = p; p += fmax / 2 + 1;
  w.S1i = p; p += fmax / 2 + 1;
  w.S2r = p; p += fmax / 2 + 1;
  w.S2i = p; p += fmax / 2 + 1;
  w.sc = p; p += fmax / 2 + 1;
  w.sps = p; p += fmax / 2 + 1;
  w.sgd = p; p += fmax / 2 + 1;
  w.tmp = p;
  double* cfa = (double*)malloc(sizeof(double) * (2 * (nap + 2) + K));
  double* cap = cfa + nap + 2;
  double* fa = cap + nap + 2;
  for (int i = 0; i <= nap; ++i) cfa[i] = i * 3000.0;
  cfa[nap + 1] = fs / 2.0;
  for (int k = 0; k < K; ++k) fa[k] = (double)k * fs / fft_size;
  for (int i = 0; i < T; ++i) {
    double* row = ap + (size_t)i * K;
    for (int k = 0; k < K; ++k) row[k] = 1.0 - EPS;
    if (f0[i] == 0.0) continue;
    if (lovetrain(x, xl, fs, f0[i], tp[i], fftl, &w) <= threshold) continue;
    cap[0] = -60.0;
    cap[nap + 1] = -EPS;
    d4c_frame(x, xl, fs, dmax(47.0, f0[i]), tp[i], fftd, nap, window, wl, &w, cap + 1);
    orc_interp1(cfa, cap, nap + 2, fa, K, row);
    for (int k = 0; k < K; ++k) row[k] = pow(10.0, row[k] / 20.0);
  }
  free(cfa);
  free(pool);
  free(window);
  return 0;
}

/* pyworld.code_aperiodicity(ap, fs): 20 log10(ap) interpolated at k*3000 Hz */
int orc_code_aperiodicity(const double* ap, int T, int fft_size, int fs, double* bap) {
  const int K = fft_size / 2 + 1;
  const int nap = (int)(dmin(15000.0, fs / 2.0 - 3000.0) / 3000.0);
  double* fa = (double*)malloc(sizeof(double) * (2 * K + nap));
  double* la = fa + K;
  double* cf = la + K;
  for (int k = 0; k < K; ++k) fa[k] = (double)k * fs / fft_size;
  for (int i = 0; i < nap; ++i) cf[i] = 3000.0 * (i + 1);
  for (int i = 0; i < T; ++i) {
    for (int k = 0; k < K; ++k) la[k] = 20.0 * log10(ap[(size_t)i * K + k]);
    orc_interp1(fa, la, K, cf, nap, bap + (size_t)i * nap);
  }
  free(fa);
  return 0;
}

/* pyworld.wav2world(x, fs, fft_size, frame_period): dio -> stonemask -> cheaptrick -> d4c.
 * f0 [T], sp [T,K], ap [T,K]; T = int(1000*xl/fs/frame_period)+1 */
int orc_wav2world(const double* x, int xl, int fs, double frame_period, int fft_size, double* f0,
                  double* sp, double* ap) {
  const int T = (int)(1000.0 * xl / fs / frame_period) + 1;
  double* tp = (double*)malloc(sizeof(double) * 2 * T);
  double* f0d = tp + T;
  int rc = orc_dio(x, xl, fs, frame_period, 71.0, 800.0, 2.0, 0.1, f0d, tp);
  if (!rc) rc = orc_stonemask(x, xl, fs, tp, f0d, T, f0);
  if (!rc && sp) rc = orc_cheaptrick(x, xl, fs, tp, f0, T, fft_size, -0.15, sp);
  if (!rc && ap) rc = orc_d4c(x, xl, fs, tp, f0, T, fft_size, 0.85, ap);
  free(tp);
  return rc;
}
