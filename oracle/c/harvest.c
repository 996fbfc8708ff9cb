/* TEST INFRASTRUCTURE ONLY -- CPU restatement of WORLD's Harvest F0 estimator (pyworld.harvest:
 * the alternative to DIO that BASELINE.json's north_star names).  The reference repository never
 * calls it (its extractor is dio + stonemask, WorldFeatLabelGen.py:432-440 via pyworld.wav2world)
 * and pyworld is not vendored, so no reference-held vector exists for it: PARITY UNPINNED.  The
 * restatement follows the published algorithm of WORLD's harvest.cpp (Morise, Interspeech 2017)
 * function by function, including its quirks (the mirrored spectrum write in the band-pass
 * convolution, the running mean in ExtendSub, the insertion sort in MakeSortedOrder).  The
 * anti-aliasing IIR of `decimate` is cheby1(3, 0.05 dB, 0.8 / r) (checked numerically against the
 * literals for r = 2 and r = 11; other ratios carry scipy's digits).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define SAFE_MIN 0.000000000001

static int hmin(int a, int b) { return a < b ? a : b; }
static int hmax(int a, int b) { return a > b ? a : b; }

/* 1 (default): reproduce the mirrored write of GetFilteredSignal; 0: plain linear convolution
 * (used by tests to separate that term from everything else) */
static int g_mirror_write = 1;
void orc_harvest_set_mirror_write(int on) { g_mirror_write = on; }

int orc_harvest_num_frames(int xl, int fs, double frame_period) {
  return (int)(1000.0 * xl / fs / frame_period) + 1;
}

/* ---- matlabfunctions.cpp: decimate ---------------------------------------------------------- */
static const double DEC_A[13][3] = {
    {0, 0, 0}, {0, 0, 0},
    {0.041156734567757189, -0.42599112459189636, 0.041037215479961225},
    {0.95039378983237421, -0.67429146741526802, 0.15412211621346472},
    {1.4499664446880223, -0.98943497080950538, 0.24578252340690199},
    {1.761093965428056, -1.255491484385977, 0.32371865077882145},
    {1.9715352749512141, -1.4686795689225343, 0.38939084349657005},
    {2.1225239019534698, -1.6395144861046296, 0.44469707800587344},
    {2.2357462340187593, -1.7780899984041356, 0.49152555365968698},
    {2.3236003491759578, -1.89215456174636, 0.53148928133729068},
    {2.3936475118069382, -1.9873904075111852, 0.56588799790270516},
    {2.450743295230728, -2.06794904601978, 0.59574774438332101},
    {2.4981398605924205, -2.1368928194784025, 0.62187513816221485}};
static const double DEC_B[13][2] = {
    {0, 0}, {0, 0},
    {0.16797464681802227, 0.50392394045406674},
    {0.071221945171178622, 0.21366583551353585},
    {0.03671075033932264, 0.11013225101796792},
    {0.021334858522387451, 0.064004575567162353},
    {0.013469181309343806, 0.04040754392803142},
    {0.0090366882681607811, 0.027110064804482345},
    {0.0063522763407111793, 0.019056829022133539},
    {0.0046331164041389242, 0.013899349212416773},
    {0.0034818622251927374, 0.010445586675578211},
    {0.0026822508007163792, 0.0080467524021491377},
    {0.0021097275904708771, 0.0063291827714126309}};

void orc_decimate_coefficients(int r, double* a3, double* b2) {
  memcpy(a3, DEC_A[r], sizeof(double) * 3);
  memcpy(b2, DEC_B[r], sizeof(double) * 2);
}

static void filter_for_decimate(const double* x, int n, int r, double* y) {
  const double* a = DEC_A[r];
  const double* b = DEC_B[r];
  double w0 = 0.0, w1 = 0.0, w2 = 0.0;
  for (int i = 0; i < n; ++i) {
    const double wt = x[i] + a[0] * w0 + a[1] * w1 + a[2] * w2;
    y[i] = b[0] * wt + b[1] * w0 + b[1] * w1 + b[0] * w2;
    w2 = w1;
    w1 = w0;
    w0 = wt;
  }
}

static void decimate(const double* x, int n, int r, double* y) {
  const int nf = 9;
  const int m = n + 2 * nf;
  double* t1 = (double*)malloc(sizeof(double) * m * 2);
  double* t2 = t1 + m;
  for (int i = 0; i < nf; ++i) t1[i] = 2 * x[0] - x[nf - i];
  for (int i = nf; i < nf + n; ++i) t1[i] = x[i - nf];
  for (int i = nf + n; i < m; ++i) t1[i] = 2 * x[n - 1] - x[n - 2 - (i - (nf + n))];
  filter_for_decimate(t1, m, r, t2);
  for (int i = 0; i < m; ++i) t1[i] = t2[m - i - 1];
  filter_for_decimate(t1, m, r, t2);
  for (int i = 0; i < m; ++i) t1[i] = t2[m - i - 1];
  const int nout = (n - 1) / r + 1;
  const int nbeg = r - r * nout + n;
  int count = 0;
  for (int i = nbeg; i < n + nf; i += r) y[count++] = t1[i + nf - 1];
  free(t1);
}

/* GetWaveformAndSpectrumSub + DC removal (harvest.cpp); y has fft entries, zero beyond yl */
static void get_waveform(const double* x, int xl, int yl, int r, int fft, double* y) {
  for (int i = 0; i < fft; ++i) y[i] = 0.0;
  if (r == 1) {
    for (int i = 0; i < xl; ++i) y[i] = x[i];
  } else {
    const int lag = (int)(ceil(140.0 / r) * r);
    const int nl = xl + lag * 2;
    double* nx = (double*)malloc(sizeof(double) * nl * 2);
    double* ny = nx + nl;
    for (int i = 0; i < nl; ++i) ny[i] = 0.0;
    for (int i = 0; i < lag; ++i) nx[i] = x[0];
    for (int i = lag; i < lag + xl; ++i) nx[i] = x[i - lag];
    for (int i = lag + xl; i < nl; ++i) nx[i] = x[xl - 1];
    decimate(nx, nl, r, ny);
    for (int i = 0; i < yl; ++i) y[i] = ny[lag / r + i];
    free(nx);
  }
  double mean = 0.0;
  for (int i = 0; i < yl; ++i) mean += y[i];
  mean /= yl;
  for (int i = 0; i < yl; ++i) y[i] -= mean;
  for (int i = yl; i < fft; ++i) y[i] = 0.0;
}

int orc_harvest_waveform(const double* x, int xl, int fs, double* y) {
  const int r = hmax(hmin(orc_mround(fs / 8000.0), 12), 1);
  const int yl = (int)ceil((double)xl / r);
  double* buf = (double*)malloc(sizeof(double) * (yl + 1));
  get_waveform(x, xl, yl, r, yl, buf);
  memcpy(y, buf, sizeof(double) * yl);
  free(buf);
  return yl;
}

/* ---- zero crossings (harvest.cpp ZeroCrossingEngine) ---------------------------------------- */
static int zc_engine(const double* s, int n, double fs, double* loc, double* itv, double* fine) {
  int count = 0;
  for (int i = 0; i < n - 1; ++i)
    if (0.0 < s[i] && s[i + 1] <= 0.0) {
      const int e = i + 1;
      fine[count++] = (double)e - s[e - 1] / (s[e] - s[e - 1]);
    }
  if (count < 2) return 0;
  for (int i = 0; i < count - 1; ++i) {
    itv[i] = fs / (fine[i + 1] - fine[i]);
    loc[i] = (fine[i] + fine[i + 1]) / 2.0 / fs;
  }
  return count - 1;
}

/* ---- candidate refinement (GetRefinedF0) ---------------------------------------------------- */
static void refined_f0(const double* x, int xl, double fs, double pos, double f0, double f0_floor,
                       double f0_ceil, double* out_f0, double* out_score, double* ws) {
  if (f0 <= 0.0) {
    *out_f0 = 0.0;
    *out_score = 0.0;
    return;
  }
  const int hw = (int)(1.5 * fs / f0 + 1.0);
  const int bl = hw * 2 + 1;
  const double wlt = (2.0 * hw + 1.0) / fs;
  const int fft = (int)pow(2.0, 2.0 + (int)(log(hw * 2.0 + 1.0) / log(2.0)));
  const int hb = fft / 2 + 1;
  double* mw = ws;
  double* dw = mw + fft;
  double* seg = dw + fft;
  double* Mr = seg + fft;
  double* Mi = Mr + hb;
  double* Dr = Mi + hb;
  double* Di = Dr + hb;
  double* work = Di + hb;
  const double bt0 = (double)(-hw) / fs;
  const int basic = orc_mround((pos + bt0) * fs + 0.001);
  for (int i = 0; i < bl; ++i) {
    const double t = ((basic + i) - 1.0) / fs - pos;
    mw[i] = 0.42 + 0.5 * cos(2.0 * M_PI * t / wlt) + 0.08 * cos(4.0 * M_PI * t / wlt);
  }
  dw[0] = -mw[1] / 2.0;
  for (int i = 1; i < bl - 1; ++i) dw[i] = -(mw[i + 1] - mw[i - 1]) / 2.0;
  dw[bl - 1] = mw[bl - 2] / 2.0;
  for (int i = 0; i < bl; ++i) seg[i] = x[hmax(0, hmin(xl - 1, basic + i - 1))] * mw[i];
  for (int i = bl; i < fft; ++i) seg[i] = 0.0;
  orc_rfft(seg, fft, Mr, Mi, work);
  for (int i = 0; i < bl; ++i) seg[i] = x[hmax(0, hmin(xl - 1, basic + i - 1))] * dw[i];
  for (int i = bl; i < fft; ++i) seg[i] = 0.0;
  orc_rfft(seg, fft, Dr, Di, work);
  const int nh = hmin((int)(fs / 2.0 / f0), 6);
  double num = 0.0, den = 0.0, score = 0.0;
  for (int i = 0; i < nh; ++i) {
    const int idx = hmin(orc_mround(f0 * fft / fs * (i + 1)), fft / 2);
    const double ps = Mr[idx] * Mr[idx] + Mi[idx] * Mi[idx];
    const double ni = Mr[idx] * Di[idx] - Mi[idx] * Dr[idx];
    const double inst = ps == 0.0 ? 0.0 : (double)idx * fs / fft + ni / ps * fs / 2.0 / M_PI;
    const double amp = sqrt(ps);
    num += amp * inst;
    den += amp * (i + 1.0);
    score += fabs((inst / (i + 1.0) - f0) / f0);
  }
  double rf = num / (den + SAFE_MIN);
  double rs = 1.0 / (score / nh + SAFE_MIN);
  if (rf < f0_floor || rf > f0_ceil || rs < 2.5) {
    rf = 0.0;
    rs = 0.0;
  }
  *out_f0 = rf;
  *out_score = rs;
}

/* ---- contour fixing ------------------------------------------------------------------------- */
static double select_best_f0(double ref, const double* cands, int nc, double allowed, double* err) {
  double best = 0.0;
  *err = allowed;
  for (int i = 0; i < nc; ++i) {
    const double t = fabs(ref - cands[i]) / ref;
    if (t > *err) continue;
    best = cands[i];
    *err = t;
  }
  return best;
}

static int boundary_list(const double* f0, int T, int* bl) {
  int nb = 0;
  int prev = 0;
  for (int i = 1; i < T; ++i) {
    const int v = (i == T - 1) ? 0 : (f0[i] > 0 ? 1 : 0);
    if (v - prev != 0) {
      bl[nb] = i - nb % 2;
      nb++;
    }
    prev = v;
  }
  return nb;
}

static void multi_channel(const double* f0, int T, const int* bl, int nb, double** mc) {
  for (int i = 0; i < nb / 2; ++i) {
    for (int j = 0; j < T; ++j) mc[i][j] = 0.0;
    for (int j = bl[i * 2]; j <= bl[i * 2 + 1]; ++j) mc[i][j] = f0[j];
  }
}

static int extend_f0(int origin, int last, int shift, const double* cand, int ldc, int nc,
                     double allowed, double* ext) {
  const int threshold = 4;
  double tmp = ext[origin];
  int shifted = origin;
  const int distance = abs(last - origin);
  int count = 0;
  double dummy;
  for (int i = 0; i <= distance; ++i) {
    const int at = origin + shift * i + shift;
    ext[at] = select_best_f0(tmp, cand + (size_t)at * ldc, nc, allowed, &dummy);
    if (ext[at] == 0.0) {
      count++;
    } else {
      tmp = ext[at];
      count = 0;
      shifted = at;
    }
    if (count == threshold) break;
  }
  return shifted;
}

static double search_score(double f0, const double* cands, const double* scores, int nc) {
  double score = 0.0;
  for (int i = 0; i < nc; ++i)
    if (f0 == cands[i] && score < scores[i]) score = scores[i];
  return score;
}

static int merge_sub(const double* f1, int st1, int ed1, const double* f2, int st2, int ed2,
                     const double* cand, const double* scr, int ldc, int nc, double* merged) {
  if (st1 <= st2 && ed1 >= ed2) return ed1;
  double s1 = 0.0, s2 = 0.0;
  for (int i = st2; i <= ed1; ++i) {
    s1 += search_score(f1[i], cand + (size_t)i * ldc, scr + (size_t)i * ldc, nc);
    s2 += search_score(f2[i], cand + (size_t)i * ldc, scr + (size_t)i * ldc, nc);
  }
  if (s1 > s2)
    for (int i = ed1; i <= ed2; ++i) merged[i] = f2[i];
  else
    for (int i = st2; i <= ed2; ++i) merged[i] = f2[i];
  return ed2;
}

static void fix_step3(const double* in, int T, int nc, const double* cand, const double* scr,
                      int ldc, double allowed, double* out) {
  for (int i = 0; i < T; ++i) out[i] = in[i];
  int* bl = (int*)malloc(sizeof(int) * (T + 2));
  const int nb = boundary_list(in, T, bl);
  const int ns = nb / 2;
  if (ns == 0) {
    free(bl);
    return;
  }
  double** mc = (double**)malloc(sizeof(double*) * ns);
  double* store = (double*)malloc(sizeof(double) * (size_t)ns * T);
  for (int i = 0; i < ns; ++i) mc[i] = store + (size_t)i * T;
  multi_channel(in, T, bl, nb, mc);
  /* Extend */
  for (int i = 0; i < ns; ++i) {
    const int e = extend_f0(bl[i * 2 + 1], hmin(T - 2, bl[i * 2 + 1] + 100), 1, cand, ldc, nc,
                            allowed, mc[i]);
    const int s = extend_f0(bl[i * 2], hmax(1, bl[i * 2] - 100), -1, cand, ldc, nc, allowed, mc[i]);
    bl[i * 2 + 1] = e;
    bl[i * 2] = s;
  }
  /* ExtendSub: the mean is carried from one section into the next, as published */
  int count = 0;
  double mean = 0.0;
  for (int i = 0; i < ns; ++i) {
    const int st = bl[i * 2], ed = bl[i * 2 + 1];
    for (int j = st; j < ed; ++j) mean += mc[i][j];
    mean /= ed - st;
    if (2200.0 / mean < ed - st) {
      double* tp = mc[count];
      mc[count] = mc[i];
      mc[i] = tp;
      int ti = bl[count * 2];
      bl[count * 2] = bl[i * 2];
      bl[i * 2] = ti;
      ti = bl[count * 2 + 1];
      bl[count * 2 + 1] = bl[i * 2 + 1];
      bl[i * 2 + 1] = ti;
      count++;
    }
  }
  if (count != 0) {
    /* MergeF0 */
    int* order = (int*)malloc(sizeof(int) * count);
    for (int i = 0; i < count; ++i) order[i] = i;
    for (int i = 1; i < count; ++i)
      for (int j = i - 1; j >= 0; --j)
        if (bl[order[j] * 2] > bl[order[i] * 2]) {
          const int t = order[i];
          order[i] = order[j];
          order[j] = t;
        } else {
          break;
        }
    for (int i = 0; i < T; ++i) out[i] = mc[0][i];
    for (int i = 1; i < count; ++i) {
      const int o = order[i];
      if (bl[o * 2] - bl[1] > 0) {
        for (int j = bl[o * 2]; j <= bl[o * 2 + 1]; ++j) out[j] = mc[o][j];
        bl[0] = bl[o * 2];
        bl[1] = bl[o * 2 + 1];
      } else {
        bl[1] = merge_sub(out, bl[0], bl[1], mc[o], bl[o * 2], bl[o * 2 + 1], cand, scr, ldc, nc, out);
      }
    }
    free(order);
  }
  free(store);
  free(mc);
  free(bl);
}

static void fix_contour(const double* cand, const double* scr, int T, int nc, int ldc, double* best) {
  double* tmp = (double*)malloc(sizeof(double) * T);
  int* bl = (int*)malloc(sizeof(int) * (T + 2));
  /* SearchF0Base */
  for (int i = 0; i < T; ++i) {
    double bs = 0.0;
    tmp[i] = 0.0;
    for (int j = 0; j < nc; ++j)
      if (scr[(size_t)i * ldc + j] > bs) {
        tmp[i] = cand[(size_t)i * ldc + j];
        bs = scr[(size_t)i * ldc + j];
      }
  }
  /* FixStep1 (0.008) */
  for (int i = 0; i < T; ++i) best[i] = 0.0;
  for (int i = 2; i < T; ++i) {
    if (tmp[i] == 0.0) continue;
    const double ref = tmp[i - 1] * 2 - tmp[i - 2];
    best[i] = (fabs((tmp[i] - ref) / ref) > 0.008 && fabs(tmp[i] - tmp[i - 1]) / tmp[i - 1] > 0.008)
                  ? 0.0
                  : tmp[i];
  }
  /* FixStep2 (6) */
  for (int i = 0; i < T; ++i) tmp[i] = best[i];
  int nb = boundary_list(best, T, bl);
  for (int i = 0; i < nb / 2; ++i) {
    if (bl[i * 2 + 1] - bl[i * 2] >= 6) continue;
    for (int j = bl[i * 2]; j <= bl[i * 2 + 1]; ++j) tmp[j] = 0.0;
  }
  /* FixStep3 (0.18) */
  fix_step3(tmp, T, nc, cand, scr, ldc, 0.18, best);
  /* FixStep4 (9) */
  for (int i = 0; i < T; ++i) tmp[i] = best[i];
  nb = boundary_list(best, T, bl);
  for (int i = 0; i < nb / 2 - 1; ++i) {
    const int distance = bl[(i + 1) * 2] - bl[i * 2 + 1] - 1;
    if (distance >= 9) continue;
    const double t0 = best[bl[i * 2 + 1]] + 1;
    const double t1 = best[bl[(i + 1) * 2]] - 1;
    const double c = (t1 - t0) / (distance + 1.0);
    int count = 1;
    for (int j = bl[i * 2 + 1] + 1; j <= bl[(i + 1) * 2] - 1; ++j) tmp[j] = t0 + c * count++;
  }
  for (int i = 0; i < T; ++i) best[i] = tmp[i];
  free(bl);
  free(tmp);
}

static void smooth_contour(const double* f0, int T, double* out) {
  const double b[2] = {0.0078202080334971724, 0.015640416066994345};
  const double a[2] = {1.7347257688092754, -0.76600660094326412};
  const int lag = 300;
  const int n = T + lag * 2;
  double* c = (double*)calloc((size_t)n * 3, sizeof(double));
  double* xs = c + n;
  double* tx = xs + n;
  int* bl = (int*)malloc(sizeof(int) * (n + 2));
  for (int i = 0; i < T; ++i) c[i + lag] = f0[i];
  const int nb = boundary_list(c, n, bl);
  for (int s = 0; s < nb / 2; ++s) {
    const int st = bl[s * 2], ed = bl[s * 2 + 1];
    for (int j = 0; j < st; ++j) xs[j] = c[st];
    for (int j = st; j <= ed; ++j) xs[j] = c[j];
    for (int j = ed + 1; j < n; ++j) xs[j] = c[ed];
    double w0 = 0.0, w1 = 0.0;
    for (int i = 0; i < n; ++i) {
      const double wt = xs[i] + a[0] * w0 + a[1] * w1;
      tx[n - i - 1] = b[0] * wt + b[1] * w0 + b[0] * w1;
      w1 = w0;
      w0 = wt;
    }
    w0 = w1 = 0.0;
    for (int i = 0; i < n; ++i) {
      const double wt = tx[i] + a[0] * w0 + a[1] * w1;
      xs[n - i - 1] = b[0] * wt + b[1] * w0 + b[0] * w1;
      w1 = w0;
      w0 = wt;
    }
    for (int i = st; i <= ed; ++i) c[i] = xs[i];
  }
  for (int i = 0; i < T; ++i) out[i] = c[i + lag];
  free(bl);
  free(c);
}

/* HarvestGeneralBody at a 1 ms frame period + the pick of every frame_period-th value (Harvest()).
 * Optional debug outputs (may be NULL): raw [channels][T1] band candidates, the refined
 * candidates/scores [T1][max_candidates] after RemoveUnreliableCandidates, and the 1 ms contour
 * before smoothing. */
int orc_harvest_debug(const double* x, int xl, int fs, double frame_period, double f0_floor,
                      double f0_ceil, double* f0_out, double* tp_out, double* raw_out,
                      double* cand_out, double* score_out, double* best_out, int* dims) {
  const double target_fs = 8000.0;
  const int r = hmax(hmin(orc_mround(fs / target_fs), 12), 1);
  const double cio = 40.0;
  const double af = f0_floor * 0.9, ac = f0_ceil * 1.1;
  const int nch = 1 + (int)(log(ac / af) / log(2.0) * cio);
  double* bnd = (double*)malloc(sizeof(double) * nch);
  for (int i = 0; i < nch; ++i) bnd[i] = af * pow(2.0, (i + 1) / cio);
  const int yl = (int)ceil((double)xl / r);
  const double afs = (double)fs / r;
  const int sample = yl + 5 + 2 * (int)(2.0 * afs / bnd[0]);
  const int fft = (int)pow(2.0, (int)(log((double)sample) / log(2.0)) + 1.0);
  const int hb = fft / 2 + 1;
  const int T1 = orc_harvest_num_frames(xl, fs, 1.0);
  const int maxc = orc_mround(nch / 10.0) * 7;
  if (dims) {
    dims[0] = nch;
    dims[1] = T1;
    dims[2] = maxc;
    dims[3] = yl;
  }

  double* y = (double*)malloc(sizeof(double) * fft);
  double* flt = (double*)malloc(sizeof(double) * fft);
  double* work = (double*)malloc(sizeof(double) * fft);
  double* Yr = (double*)malloc(sizeof(double) * hb * 4);
  double *Yi = Yr + hb, *Fr = Yi + hb, *Fi = Fr + hb;
  double* tp1 = (double*)malloc(sizeof(double) * T1);
  double* raw = (double*)calloc((size_t)nch * T1, sizeof(double));
  double* sets = (double*)malloc(sizeof(double) * 4 * T1);
  double* fine = (double*)malloc(sizeof(double) * (yl + 1));
  double* loc = (double*)malloc(sizeof(double) * (yl + 1) * 8);
  double* cand = (double*)calloc((size_t)T1 * maxc * 3, sizeof(double));
  double* scr = cand + (size_t)T1 * maxc;
  double* snap = scr + (size_t)T1 * maxc;
  if (!y || !flt || !work || !Yr || !tp1 || !raw || !sets || !fine || !loc || !cand) return -1;

  get_waveform(x, xl, yl, r, fft, y);
  orc_rfft(y, fft, Yr, Yi, work);
  for (int i = 0; i < T1; ++i) tp1[i] = i * 1.0 / 1000.0;

  /* GetRawF0Candidates */
  for (int c = 0; c < nch; ++c) {
    const int half = orc_mround(afs / bnd[c] * 2.0);
    orc_nuttall(half * 2 + 1, flt);
    for (int i = -half; i <= half; ++i) flt[i + half] *= cos(2 * M_PI * bnd[c] * i / afs);
    for (int i = half * 2 + 1; i < fft; ++i) flt[i] = 0.0;
    orc_rfft(flt, fft, Fr, Fi, work);
    /* convolution with the published mirror write: bin fft-i-1 receives bin i while the loop is
     * still running, so bins fft/2-1 and fft/2 of the product are overwritten (harvest.cpp
     * GetFilteredSignal) */
    {
      double t = Yr[0] * Fr[0] - Yi[0] * Fi[0];
      Fi[0] = Yr[0] * Fi[0] + Yi[0] * Fr[0];
      Fr[0] = t;
      for (int i = 1; i <= fft / 2; ++i) {
        t = Yr[i] * Fr[i] - Yi[i] * Fi[i];
        Fi[i] = Yr[i] * Fi[i] + Yi[i] * Fr[i];
        Fr[i] = t;
        const int m = fft - i - 1;
        if (g_mirror_write && m <= fft / 2) {
          Fr[m] = Fr[i];
          Fi[m] = Fi[i];
        }
      }
    }
    orc_irfft(Fr, Fi, fft, flt, work);
    const int bias = half + 1;
    for (int i = 0; i < yl; ++i) flt[i] = flt[i + bias];
    double* L[4];
    double* I[4];
    int cnt[4];
    for (int k = 0; k < 4; ++k) {
      L[k] = loc + (size_t)(2 * k) * (yl + 1);
      I[k] = loc + (size_t)(2 * k + 1) * (yl + 1);
    }
    cnt[0] = zc_engine(flt, yl, afs, L[0], I[0], fine);
    for (int i = 0; i < yl; ++i) flt[i] = -flt[i];
    cnt[1] = zc_engine(flt, yl, afs, L[1], I[1], fine);
    for (int i = 0; i < yl - 1; ++i) flt[i] = flt[i] - flt[i + 1];
    cnt[2] = zc_engine(flt, yl - 1, afs, L[2], I[2], fine);
    for (int i = 0; i < yl - 1; ++i) flt[i] = -flt[i];
    cnt[3] = zc_engine(flt, yl - 1, afs, L[3], I[3], fine);
    double* rc = raw + (size_t)c * T1;
    if (cnt[0] - 2 > 0 && cnt[1] - 2 > 0 && cnt[2] - 2 > 0 && cnt[3] - 2 > 0) {
      for (int k = 0; k < 4; ++k) orc_interp1(L[k], I[k], cnt[k], tp1, T1, sets + (size_t)k * T1);
      const double upper = bnd[c] * 1.1, lower = bnd[c] * 0.9;
      for (int i = 0; i < T1; ++i) {
        double v = (sets[i] + sets[T1 + i] + sets[2 * T1 + i] + sets[3 * T1 + i]) / 4.0;
        if (v > upper || v < lower || v > f0_ceil || v < f0_floor) v = 0.0;
        rc[i] = v;
      }
    }
  }
  if (raw_out) memcpy(raw_out, raw, sizeof(double) * (size_t)nch * T1);

  /* DetectOfficialF0Candidates */
  int ncand = 0;
  {
    int* vuv = (int*)malloc(sizeof(int) * nch * 3);
    int *st = vuv + nch, *ed = st + nch;
    for (int i = 0; i < T1; ++i) {
      for (int j = 0; j < nch; ++j) vuv[j] = raw[(size_t)j * T1 + i] > 0 ? 1 : 0;
      vuv[0] = vuv[nch - 1] = 0;
      int ns = 0;
      for (int j = 1; j < nch; ++j) {
        const int d = vuv[j] - vuv[j - 1];
        if (d == 1) st[ns] = j;
        if (d == -1) ed[ns++] = j;
      }
      int n = 0;
      double* row = cand + (size_t)i * maxc;
      for (int s = 0; s < ns; ++s) {
        if (ed[s] - st[s] < 10) continue;
        double t = 0.0;
        for (int j = st[s]; j < ed[s]; ++j) t += raw[(size_t)j * T1 + i];
        t /= (ed[s] - st[s]);
        row[n++] = t;
      }
      ncand = hmax(ncand, n);
    }
    free(vuv);
  }
  /* OverlapF0Candidates */
  for (int i = 1; i <= 3; ++i)
    for (int j = 0; j < ncand; ++j) {
      for (int k = i; k < T1; ++k) cand[(size_t)k * maxc + j + ncand * i] = cand[(size_t)(k - i) * maxc + j];
      for (int k = 0; k < T1 - i; ++k)
        cand[(size_t)k * maxc + j + ncand * (i + 3)] = cand[(size_t)(k + i) * maxc + j];
    }
  const int nc = ncand * 7;
  if (dims) dims[4] = nc;

  /* RefineF0Candidates */
  {
    const int hwmax = (int)(1.5 * afs / f0_floor + 1.0) + 2;
    const int fmax = (int)pow(2.0, 2.0 + (int)(log(hwmax * 2.0 + 1.0) / log(2.0)));
    double* ws = (double*)malloc(sizeof(double) * (size_t)fmax * 8);
    for (int i = 0; i < T1; ++i)
      for (int j = 0; j < nc; ++j) {
        const size_t at = (size_t)i * maxc + j;
        double f = cand[at];
        /* candidates above what the window supports cannot occur: the band limit is f0_ceil */
        refined_f0(y, yl, afs, tp1[i], f, f0_floor, f0_ceil, &cand[at], &scr[at], ws);
      }
    free(ws);
  }
  /* RemoveUnreliableCandidates */
  memcpy(snap, cand, sizeof(double) * (size_t)T1 * maxc);
  for (int i = 1; i < T1 - 1; ++i)
    for (int j = 0; j < nc; ++j) {
      const size_t at = (size_t)i * maxc + j;
      const double ref = cand[at];
      if (ref == 0) continue;
      double e1, e2;
      select_best_f0(ref, snap + (size_t)(i + 1) * maxc, nc, 1.0, &e1);
      select_best_f0(ref, snap + (size_t)(i - 1) * maxc, nc, 1.0, &e2);
      if ((e1 < e2 ? e1 : e2) <= 0.05) continue;
      cand[at] = 0;
      scr[at] = 0;
    }
  if (cand_out) memcpy(cand_out, cand, sizeof(double) * (size_t)T1 * maxc);
  if (score_out) memcpy(score_out, scr, sizeof(double) * (size_t)T1 * maxc);

  double* best = (double*)malloc(sizeof(double) * T1 * 2);
  double* smooth = best + T1;
  fix_contour(cand, scr, T1, nc, maxc, best);
  if (best_out) memcpy(best_out, best, sizeof(double) * T1);
  smooth_contour(best, T1, smooth);

  const int T = orc_harvest_num_frames(xl, fs, frame_period);
  for (int i = 0; i < T; ++i) {
    tp_out[i] = i * frame_period / 1000.0;
    f0_out[i] = frame_period == 1.0 ? smooth[i] : smooth[hmin(T1 - 1, orc_mround(tp_out[i] * 1000.0))];
  }
  free(best);
  free(cand);
  free(loc);
  free(fine);
  free(sets);
  free(raw);
  free(tp1);
  free(Yr);
  free(work);
  free(flt);
  free(y);
  free(bnd);
  return 0;
}

int orc_harvest(const double* x, int xl, int fs, double frame_period, double f0_floor,
                double f0_ceil, double* f0_out, double* tp_out) {
  return orc_harvest_debug(x, xl, fs, frame_period, f0_floor, f0_ceil, f0_out, tp_out, NULL, NULL,
                           NULL, NULL, NULL);
}
