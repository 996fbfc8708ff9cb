// Harvest F0 estimation (WORLD harvest.cpp; pyworld.harvest) for batches of utterances: the
// alternative to DIO + StoneMask that BASELINE.json's north_star names for the WORLD feature path
// (the reference itself extracts with pyworld.wav2world = dio + stonemask,
// src/data_preparation/world/WorldFeatLabelGen.py:792-793, so nothing in it calls this).
//
// WORLD runs the estimator on a 1 ms grid of the signal decimated to about 8 kHz:
//   hv_decimate_kernel    zero-phase 3rd-order Chebyshev IIR + pick every r-th sample, DC removal
//   hv_bandpass_kernel    152 Nuttall x cosine band-pass filters (40 per octave).  WORLD multiplies
//                         spectra of a 2^16-point FFT; here the same linear convolutions are a
//                         Toeplitz product on the fp64 matrix cores: A = taps of 16 channels
//                         [16 x K], B = sliding windows of the signal [K x 16 samples]
//   hv_events_kernel      negative-going zero crossings of {s, -s, ds, -ds} per channel, compacted
//   hv_raw_kernel         per (channel, frame): mean of the four interpolated interval-f0
//   hv_detect_kernel      per frame: runs of >= 10 voiced channels -> candidates
//   hv_refine_kernel      per (frame, candidate incl. the +-3 frame overlap): instantaneous-
//                         frequency refinement.  Only <= 6 harmonic bins of each windowed segment
//                         are ever read, so the two FFTs per candidate are replaced by a direct
//                         evaluation of those bins (one wave per candidate)
//   hv_remove_kernel      candidates without a neighbour within 5 % in the adjacent frames
//   hv_contour*_kernel    FixF0Contour steps 1-4 (sections extend in parallel, the merge walks
//                         them in order) and the zero-phase 2nd-order smoothing per section
#include <algorithm>
#include <cmath>
#include <vector>

#include "context.h"
#include "world_dev.h"

namespace itts {
using namespace wd;

typedef double hv_double4 __attribute__((ext_vector_type(4)));

constexpr int HV_MAXCH = 512;
constexpr int HV_EXT = 102;             // frames a section may grow on either side
constexpr int HV_SECW = 2 * HV_EXT + 2;
constexpr int HV_LAG = 300;             // SmoothF0Contour padding
constexpr int HV_REFINE_FRAMES = 4;     // one wave per frame

struct HvTables {
  double bnd[HV_MAXCH];
  int half[HV_MAXCH];
  int evoff[HV_MAXCH];   // offset of the channel's four event lists inside an utterance's block
  int evcap[HV_MAXCH];
  int hmax[HV_MAXCH / 16];
  int kt[HV_MAXCH / 16];
  int woff[HV_MAXCH / 16];
};

struct HvParams {
  int fs, r, lag;
  double afs, frame_period, f0_floor, f0_ceil;
  int nch, ntiles, nbase, maxc;
  int pad;        // zeros kept either side of the decimated signal
  int64_t evtot;  // doubles of event storage per utterance
  int fft_max, log_fft_max, bl_max;
  double da[3], db[2];
};

struct HvUtt {
  int64_t x_off, f_off;
  int xl, yl, T1, T;
  int nfft;  // WORLD's FFT length for this utterance (enters through the mirrored spectrum write)
  int64_t y_off, dec_off, sig_off, ev_off, cnt_off, raw_off, base_off, cand_off, ctr_off, mc_off,
      sm_off;
};

// ---- decimation --------------------------------------------------------------------------------
// y = b(z)/a(z) x with the state recurrence of WORLD's FilterForDecimate (state s = (w0, w1, w2),
// s' = C s + x e0); the output is written reversed so that the second call runs the time-reversed
// pass.  The recurrence is linear, so the workgroup splits the signal into one chunk per thread:
// every thread runs its chunk from a zero state, one lane chains the chunk states through C^L,
// and every thread runs its chunk again from the true state (same values as the sequential loop up
// to rounding).
__device__ inline void hv_iir3_reversed(const double* in, double* out, int m, const HvParams& p,
                                        double* sh /* 9 + 3 * NT doubles */) {
  const double a0 = p.da[0], a1 = p.da[1], a2 = p.da[2], b0 = p.db[0], b1 = p.db[1];
  const int L = (m + NT - 1) / NT;
  const int lo = min(m, (int)threadIdx.x * L), hi = min(m, lo + L);
  double* M = sh;        // C^L, column-major: M[3 * q + r] = (C^L)[r][q]
  double* init = sh + 9; // [NT][3]
  __syncthreads();
  if (threadIdx.x < 3) {
    double w0 = threadIdx.x == 0, w1 = threadIdx.x == 1, w2 = threadIdx.x == 2;
    for (int i = 0; i < L; ++i) {
      const double wt = a0 * w0 + a1 * w1 + a2 * w2;
      w2 = w1;
      w1 = w0;
      w0 = wt;
    }
    M[3 * threadIdx.x] = w0;
    M[3 * threadIdx.x + 1] = w1;
    M[3 * threadIdx.x + 2] = w2;
  }
  {
    double w0 = 0.0, w1 = 0.0, w2 = 0.0;
    for (int i = lo; i < hi; ++i) {
      const double wt = in[i] + a0 * w0 + a1 * w1 + a2 * w2;
      w2 = w1;
      w1 = w0;
      w0 = wt;
    }
    init[3 * threadIdx.x] = w0;
    init[3 * threadIdx.x + 1] = w1;
    init[3 * threadIdx.x + 2] = w2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int t = 0; t < NT; ++t) {
      const double l0 = init[3 * t], l1 = init[3 * t + 1], l2 = init[3 * t + 2];
      init[3 * t] = s0;
      init[3 * t + 1] = s1;
      init[3 * t + 2] = s2;
      const double n0 = M[0] * s0 + M[3] * s1 + M[6] * s2 + l0;
      const double n1 = M[1] * s0 + M[4] * s1 + M[7] * s2 + l1;
      const double n2 = M[2] * s0 + M[5] * s1 + M[8] * s2 + l2;
      s0 = n0;
      s1 = n1;
      s2 = n2;
    }
  }
  __syncthreads();
  {
    double w0 = init[3 * threadIdx.x], w1 = init[3 * threadIdx.x + 1], w2 = init[3 * threadIdx.x + 2];
    for (int i = lo; i < hi; ++i) {
      const double wt = in[i] + a0 * w0 + a1 * w1 + a2 * w2;
      out[m - 1 - i] = b0 * wt + b1 * w0 + b1 * w1 + b0 * w2;
      w2 = w1;
      w1 = w0;
      w0 = wt;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(NT) void hv_decimate_kernel(const double* __restrict__ x,
                                                         const HvUtt* __restrict__ utts, HvParams p,
                                                         double* __restrict__ dec,
                                                         double* __restrict__ ypad) {
  __shared__ double buf[9 + 3 * NT];
  __shared__ double red[8];
  const HvUtt u = utts[blockIdx.x];
  const double* xs = x + u.x_off;
  double* yp = ypad + u.y_off;
  for (int i = threadIdx.x; i < p.pad; i += NT) {
    yp[i] = 0.0;
    yp[p.pad + u.yl + i] = 0.0;
  }
  double* y = yp + p.pad;
  double s = 0.0;
  if (p.r == 1) {
    for (int i = threadIdx.x; i < u.yl; i += NT) {
      const double v = xs[i];
      y[i] = v;
      s += v;
    }
  } else {
    const int lag = p.lag, nl = u.xl + 2 * lag, nf = 9, m = nl + 2 * nf, xl = u.xl;
    double* t1 = dec + u.dec_off;
    double* t2 = t1 + m;
    auto nx = [&](int i) {
      int j = i - lag;
      j = j < 0 ? 0 : (j > xl - 1 ? xl - 1 : j);
      return xs[j];
    };
    for (int i = threadIdx.x; i < m; i += NT) {
      double v;
      if (i < nf) v = 2 * nx(0) - nx(nf - i);
      else if (i < nf + nl) v = nx(i - nf);
      else v = 2 * nx(nl - 1) - nx(nl - 2 - (i - (nf + nl)));
      t1[i] = v;
    }
    __syncthreads();
    hv_iir3_reversed(t1, t2, m, p, buf);
    hv_iir3_reversed(t2, t1, m, p, buf);
    const int nout = (nl - 1) / p.r + 1;
    const int nbeg = p.r - p.r * nout + nl;
    for (int i = threadIdx.x; i < u.yl; i += NT) {
      const double v = t1[nbeg + (lag / p.r + i) * p.r + nf - 1];
      y[i] = v;
      s += v;
    }
  }
  s = bsum(s, red);
  const double mean = s / u.yl;
  __syncthreads();
  for (int i = threadIdx.x; i < u.yl; i += NT) y[i] -= mean;
}

// ---- the mirrored spectrum write of GetFilteredSignal ------------------------------------------------
// WORLD multiplies the spectra in place and copies every product P[i] to bin N - i - 1 while the
// loop is still running: P[N/2 - 1] lands in bin N/2 before that bin is multiplied, so the inverse
// transform sees P'[N/2] = Y[N/2] * P[N/2 - 1] and P'[N/2 - 1] = P'[N/2].  In the time domain that
// adds (-1)^m / N * (dN + 2 Re(d1 exp(-2 pi i m / N))) to every filtered sample, with
// dN = Re P'[N/2] - Y[N/2] F[N/2] and d1 = P'[N/2] - P[N/2 - 1]: about 1e-5 of the in-band level for
// the shortest filters.  This kernel evaluates the two bins of Y and F directly and stores, per
// (utterance, channel), G and H such that sample n (m = n + half + 1) receives
// (-1)^n (G + Hr cos(2 pi n / N) + Hi sin(2 pi n / N)).  grid (utterances).
__global__ __launch_bounds__(NT) void hv_mirror_kernel(const HvUtt* __restrict__ utts, HvParams p,
                                                       const HvTables* __restrict__ tab,
                                                       const double* __restrict__ wt,
                                                       const double* __restrict__ ypad,
                                                       double* __restrict__ coef) {
  __shared__ double red[8];
  const HvUtt u = utts[blockIdx.x];
  const double* y = ypad + u.y_off + p.pad;
  const double invn = 1.0 / (double)u.nfft;
  double a = 0.0, br = 0.0, bi = 0.0;  // Y[N/2], Y[N/2 - 1] = sum y[n] (-1)^n exp(+2 pi i n / N)
  for (int n = threadIdx.x; n < u.yl; n += NT) {
    double sn, cs;
    sincospi(2.0 * n * invn, &sn, &cs);
    const double v = (n & 1) ? -y[n] : y[n];
    a += v;
    br += v * cs;
    bi += v * sn;
  }
  a = bsum(a, red);
  br = bsum(br, red);
  bi = bsum(bi, red);
  double* out = coef + (int64_t)blockIdx.x * p.nch * 3;
  for (int c = threadIdx.x; c < p.nch; c += NT) {
    const int t = c >> 4, cc = c & 15, hmax = tab->hmax[t], h = tab->half[c];
    const double* W = wt + tab->woff[t];
    double fn = 0.0, fr = 0.0, fi = 0.0;  // F[N/2], F[N/2 - 1] over taps f[j], j = d + h
    for (int d = -h; d <= h; ++d) {
      const int j = d + h;
      double sn, cs;
      sincospi(2.0 * j * invn, &sn, &cs);
      const double v = (j & 1) ? -W[(d + hmax) * 16 + cc] : W[(d + hmax) * 16 + cc];
      fn += v;
      fr += v * cs;
      fi += v * sn;
    }
    const double p1r = br * fr - bi * fi, p1i = br * fi + bi * fr;  // P[N/2 - 1]
    const double dN = a * p1r - a * fn;
    const double d1r = (a - 1.0) * p1r, d1i = (a - 1.0) * p1i;
    double sn, cs;  // exp(-2 pi i (h + 1) / N)
    sincospi(2.0 * (h + 1) * invn, &sn, &cs);
    const double sg = ((h + 1) & 1) ? -invn : invn;
    out[c * 3 + 0] = sg * dN;
    out[c * 3 + 1] = 2.0 * sg * (d1r * cs + d1i * sn);
    out[c * 3 + 2] = 2.0 * sg * (d1i * cs - d1r * sn);
  }
}

// ---- band-pass filters on the fp64 matrix cores ---------------------------------------------------
// sig_c[n] = sum_{d=-half_c}^{half_c} g_c(d) y[n + 1 + d]   (the +1 is WORLD's delay compensation
// of half + 1 for a filter centred at half).  v_mfma_f64_16x16x4_f64: A lane l = A[l % 16][l / 16],
// B lane l = B[l / 16][l % 16], D lane l = D[(l / 16) + 4 i][l % 16].  A = taps (rows = channels of
// the tile, zero outside each channel's own length), B = signal windows (columns = samples).
// grid (sample tiles of 256, channel tiles, utterances); one wave = 64 samples x 16 channels.
__global__ __launch_bounds__(NT) void hv_bandpass_kernel(const HvUtt* __restrict__ utts, HvParams p,
                                                         const HvTables* __restrict__ tab,
                                                         const double* __restrict__ wt,
                                                         const double* __restrict__ ypad,
                                                         const double* __restrict__ coef,
                                                         double* __restrict__ sig) {
  extern __shared__ __attribute__((aligned(16))) char smem_bp[];
  double* sy = reinterpret_cast<double*>(smem_bp);
  const HvUtt u = utts[blockIdx.z];
  const int n0 = blockIdx.x * NT;
  if (n0 >= u.yl) return;
  const int t = blockIdx.y;
  const int hmax = tab->hmax[t], kt = tab->kt[t];
  const double* W = wt + tab->woff[t];
  const double* yp = ypad + u.y_off;
  const int total = u.yl + 2 * p.pad;
  const int src0 = p.pad + n0 + 1 - hmax;  // >= 0 because pad > hmax
  for (int j = threadIdx.x; j < NT + kt; j += NT) {
    const int idx = src0 + j;
    sy[j] = idx < total ? yp[idx] : 0.0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  hv_double4 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = hv_double4{0.0, 0.0, 0.0, 0.0};
  const double* syw = sy + 64 * wv + l15 + l4;
  double a = W[l4 * 16 + l15];
  for (int kb = 0; kb < kt; kb += 4) {
    const double a_next = (kb + 4 < kt) ? W[(kb + 4 + l4) * 16 + l15] : 0.0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const double b = syw[16 * m + kb];
      acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
    }
    a = a_next;
  }
  double* out = sig + u.sig_off;
  // the mirrored-write term (hv_mirror_kernel)
  const double* cf = coef + (int64_t)blockIdx.z * p.nch * 3;
  double G[4], Hr[4], Hi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = min(16 * t + l4 + 4 * i, p.nch - 1);
    G[i] = cf[c * 3];
    Hr[i] = cf[c * 3 + 1];
    Hi[i] = cf[c * 3 + 2];
  }
  const double invn = 1.0 / (double)u.nfft;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int n = n0 + 64 * wv + 16 * m + l15;
    double sn, cs;
    sincospi(2.0 * n * invn, &sn, &cs);
    const double sg = (n & 1) ? -1.0 : 1.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = 16 * t + l4 + 4 * i;
      if (c < p.nch && n < u.yl)
        out[(int64_t)c * u.yl + n] = acc[m][i] + sg * (G[i] + Hr[i] * cs + Hi[i] * sn);
    }
  }
}

// ---- zero-crossing events -----------------------------------------------------------------------------
// grid (channels, utterances).  Event type 0: s, 1: -s, 2: differences of -s, 3: their negation
// (GetFourZeroCrossingIntervals).  fine[slot] = e - v[e-1] / (v[e] - v[e-1]), e = i + 1.
__global__ __launch_bounds__(NT) void hv_events_kernel(const HvUtt* __restrict__ utts, HvParams p,
                                                       const HvTables* __restrict__ tab,
                                                       const double* __restrict__ sig,
                                                       double* __restrict__ ev, int* __restrict__ counts,
                                                       int* __restrict__ err) {
  __shared__ int wcnt[4][4];
  __shared__ int base_s[4];
  const int c = blockIdx.x;
  const HvUtt u = utts[blockIdx.y];
  const int yl = u.yl;
  const double* s = sig + u.sig_off + (int64_t)c * yl;
  const int cap = tab->evcap[c];
  double* out = ev + u.ev_off + tab->evoff[c];
  if (threadIdx.x < 4) base_s[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i0 = 0; i0 < yl - 1; i0 += NT) {
    const int i = i0 + threadIdx.x;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    bool edge[4] = {false, false, false, false};
    if (i < yl - 1) {
      const double s0 = s[i], s1 = s[i + 1];
      a[0] = s0;
      b[0] = s1;
      a[1] = -s0;
      b[1] = -s1;
      if (i < yl - 2) {
        const double s2 = s[i + 2];
        a[2] = (-s0) - (-s1);
        b[2] = (-s1) - (-s2);
        a[3] = -a[2];
        b[3] = -b[2];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) edge[k] = (0.0 < a[k]) && (b[k] <= 0.0);
    }
    int before[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long bal = __ballot(edge[k]);
      before[k] = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wcnt[k][wv] = __popcll(bal);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (edge[k]) {
        int off = base_s[k];
        for (int q = 0; q < wv; ++q) off += wcnt[k][q];
        const int slot = off + before[k];
        if (slot < cap) out[(int64_t)k * cap + slot] = (double)(i + 1) - a[k] / (b[k] - a[k]);
        else *err = 1;
      }
    }
    __syncthreads();
    if (threadIdx.x < 4)
      base_s[threadIdx.x] += wcnt[threadIdx.x][0] + wcnt[threadIdx.x][1] + wcnt[threadIdx.x][2] +
                             wcnt[threadIdx.x][3];
    __syncthreads();
  }
  if (threadIdx.x < 4) counts[u.cnt_off + c * 4 + threadIdx.x] = min(base_s[threadIdx.x], cap);
}

// ---- raw candidates per (channel, 1 ms frame) -------------------------------------------------------
__global__ __launch_bounds__(NT) void hv_raw_kernel(const HvUtt* __restrict__ utts, HvParams p,
                                                    const HvTables* __restrict__ tab,
                                                    const double* __restrict__ ev,
                                                    const int* __restrict__ counts,
                                                    double* __restrict__ raw) {
  const int c = blockIdx.y;
  const HvUtt u = utts[blockIdx.z];
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= u.T1) return;
  const double fs = p.afs;
  const double t = (double)i * 1.0 / 1000.0;
  const int* cnt = counts + u.cnt_off + c * 4;
  const int cap = tab->evcap[c];
  const double* lists = ev + u.ev_off + tab->evoff[c];
  bool ok = true;
  for (int k = 0; k < 4; ++k) ok = ok && (cnt[k] >= 2) && (cnt[k] - 1 - 2 > 0);
  double cand = 0.0;
  if (ok) {
    double v[4];
    for (int k = 0; k < 4; ++k) {
      const double* fe = lists + (int64_t)k * cap;
      const int n = cnt[k] - 1;  // intervals: loc[j] = (fe[j]+fe[j+1])/2/fs, f0[j] = fs/(fe[j+1]-fe[j])
      // histc: number of loc[j] <= t, clamped to [1, n-1].  The events of a band are close to
      // evenly spaced, so the search starts from the proportional guess and gallops outwards
      // (same result as a bisection of [0, n), a third of the dependent loads)
      auto loc_le = [&](int j) { return (fe[j] + fe[j + 1]) / 2.0 / fs <= t; };
      int lo, hi;
      {
        int g = (int)((double)i / (double)u.T1 * n);
        g = g < 0 ? 0 : (g > n - 1 ? n - 1 : g);
        if (loc_le(g)) {
          lo = g + 1;
          int pb = g + 1, step = 1;
          while (pb < n && loc_le(pb)) {
            lo = pb + 1;
            pb += step;
            step <<= 1;
          }
          hi = pb < n ? pb : n;
        } else {
          hi = g;
          int pb = g - 1, step = 1;
          while (pb >= 0 && !loc_le(pb)) {
            hi = pb;
            pb -= step;
            step <<= 1;
          }
          lo = pb < 0 ? 0 : pb + 1;
        }
      }
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (loc_le(mid)) lo = mid + 1; else hi = mid;
      }
      int kk = lo;
      if (kk < 1) kk = 1;
      if (kk > n - 1) kk = n - 1;
      const double x0 = (fe[kk - 1] + fe[kk]) / 2.0 / fs, x1 = (fe[kk] + fe[kk + 1]) / 2.0 / fs;
      const double y0 = fs / (fe[kk] - fe[kk - 1]), y1 = fs / (fe[kk + 1] - fe[kk]);
      const double sfrac = (t - x0) / (x1 - x0);
      v[k] = y0 + sfrac * (y1 - y0);
    }
    const double m = (v[0] + v[1] + v[2] + v[3]) / 4.0;
    const double bf = tab->bnd[c];
    if (!(m > bf * 1.1 || m < bf * 0.9 || m > p.f0_ceil || m < p.f0_floor)) cand = m;
  }
  raw[u.raw_off + (int64_t)c * u.T1 + i] = cand;
}

// ---- DetectOfficialF0Candidates: runs of >= 10 voiced channels -----------------------------------------
__global__ __launch_bounds__(NT) void hv_detect_kernel(const HvUtt* __restrict__ utts, HvParams p,
                                                       const double* __restrict__ raw,
                                                       double* __restrict__ base, int* __restrict__ ncand) {
  const HvUtt u = utts[blockIdx.y];
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= u.T1) return;
  const double* r = raw + u.raw_off + i;
  double* row = base + u.base_off + (int64_t)i * p.nbase;
  int n = 0, prev = 0, st = 0;
  double sum = 0.0;
  for (int j = 1; j < p.nch; ++j) {
    const double v = r[(int64_t)j * u.T1];
    const int cur = (j == p.nch - 1) ? 0 : (v > 0 ? 1 : 0);
    if (cur - prev == 1) {
      st = j;
      sum = 0.0;
    }
    if (cur - prev == -1) {
      if (j - st >= 10 && n < p.nbase) row[n++] = sum / (j - st);
    }
    if (cur) sum += v;
    prev = cur;
  }
  for (int k = n; k < p.nbase; ++k) row[k] = 0.0;
  if (n > 0) atomicMax(ncand + blockIdx.y, n);
}

// ---- GetRefinedF0 for every (frame, candidate slot) --------------------------------------------------------
// Slot j of frame i: column j % nc0 of frame i + shift(j / nc0), shift = 0,-1,-2,-3,+1,+2,+3
// (OverlapF0Candidates).  One wave per frame walks the slots.  Per slot: (1) the Blackman-type
// window on the wave's lanes (one sincos per lane, then rotations), (2) windowed / derivative-
// windowed samples to LDS, (3) lanes regroup as 8 harmonics x 8 sample phases and accumulate the
// harmonic bins of both spectra, (4) two short butterfly reductions.
__device__ __forceinline__ double hv_xor_sum(double v, int mask) { return v + __shfl_xor(v, mask, 64); }

// The window's two rotations (by 64 samples and by one) depend on the half width hw only: tabulated per
// call with the expressions the refinement used per slot (same values bit for bit), so that a slot
// costs one sincospi instead of three.  rot[hw] = {sin, cos of 2 pi 64 / bl; sin, cos of 2 pi / bl}.
__global__ __launch_bounds__(64) void hv_rot_table_kernel(double fs, int hw_max, double4* __restrict__ rot) {
  const int hw = blockIdx.x * 64 + threadIdx.x;
  if (hw > hw_max) return;
  const double wlt = (2.0 * hw + 1.0) / fs;
  double rs, rc, ds1, dc1;
  sincospi(2.0 * 64.0 / (fs * wlt), &rs, &rc);
  sincospi(2.0 / (fs * wlt), &ds1, &dc1);
  rot[hw] = make_double4(rs, rc, ds1, dc1);
}

__global__ __launch_bounds__(NT) void hv_refine_kernel(const HvUtt* __restrict__ utts, HvParams p,
                                                       const double2* __restrict__ g_tw,
                                                       const double4* __restrict__ rot,
                                                       const double* __restrict__ ypad,
                                                       const double* __restrict__ base,
                                                       const int* __restrict__ ncand,
                                                       double* __restrict__ cand,
                                                       double* __restrict__ score) {
  extern __shared__ __attribute__((aligned(16))) char smem_rf[];
  double2* tw = reinterpret_cast<double2*>(smem_rf);                   // [fft_max]: exp(-2 pi i j / fft_max)
  double2* ws_all = tw + p.fft_max;                                     // [4][bl_max + 2]: (windowed, derivative-windowed)
  const HvUtt u = utts[blockIdx.y];
  const int f0i = blockIdx.x * HV_REFINE_FRAMES;
  if (f0i >= u.T1) return;
  // the whole circle from the half table exp(+2 pi i k / fft_max), k < fft_max / 2 (sign flips only: the
  // values the half table gave), so that the harmonic loop fetches a factor without a case split
  for (int k = threadIdx.x; k < p.fft_max / 2; k += NT) {
    const double2 w = g_tw[k];
    tw[k] = make_double2(w.x, -w.y);
    tw[k + p.fft_max / 2] = make_double2(-w.x, w.y);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = f0i + wv;
  if (i >= u.T1) return;
  double2* wsamp = ws_all + (size_t)wv * (p.bl_max + 2);
  const int nc0 = ncand[blockIdx.y];
  const int nc = nc0 * 7;
  const double* y = ypad + u.y_off + p.pad;
  const double* bs = base + u.base_off;
  double* crow = cand + u.cand_off + (int64_t)i * p.maxc;
  double* srow = score + u.cand_off + (int64_t)i * p.maxc;
  const double fs = p.afs;
  const double pos = (double)i * 1.0 / 1000.0;
  const int hh = lane >> 3, g = lane & 7;
  // 64 slots at a time: every lane fetches one slot's source candidate, the wave then walks the
  // non-empty ones
  for (int j0 = 0; j0 < nc; j0 += 64) {
  double fl = 0.0;
  {
    const int j = j0 + lane;
    if (j < nc) {
      const int q = j / nc0, col = j - q * nc0;
      const int src = q == 0 ? i : (q <= 3 ? i - q : i + (q - 3));
      if (src >= 0 && src < u.T1) fl = bs[(int64_t)src * p.nbase + col];
      if (fl <= 0.0) {
        crow[j] = 0.0;
        srow[j] = 0.0;
      }
    }
  }
  for (unsigned long long todo = __ballot(fl > 0.0); todo; todo &= todo - 1) {
    const int jl = __builtin_ctzll(todo);
    const int j = j0 + jl;
    const double f0 = __shfl(fl, jl, 64);
    const int hw = __builtin_amdgcn_readfirstlane((int)(1.5 * fs / f0 + 1.0));   // f0 is the same in every lane
    const int bl = hw * 2 + 1;
    const double wlt = (2.0 * hw + 1.0) / fs;
    const int lg = 2 + ilog2(bl);  // bl is odd: floor(log2) is exact
    const int fft = 1 << lg;
    const int tstride = p.fft_max >> lg;
    const double bt0 = (double)(-hw) / fs;
    const int basic = mround((pos + bt0) * fs + 0.001);
    {  // window theta_k = 2 pi t_k / wlt, t_k = (basic + k - 1) / fs - pos: one sincos per lane, then
       // rotations by 64 samples; the neighbours the derivative window needs are rotations by one
      const double t = ((basic + lane) - 1.0) / fs - pos;
      double sn, cs;
      sincospi(2.0 * t / wlt, &sn, &cs);
      const double4 rt = rot[hw];
      const double rs = rt.x, rc = rt.y, ds1 = rt.z, dc1 = rt.w;
      auto win = [](double c) { return 0.42 + 0.5 * c + 0.08 * (2.0 * c * c - 1.0); };
      // four rows of 64 samples at a time, their loads in flight together (one trip to the cache per
      // slot for windows of up to 256 samples instead of one per row)
      for (int k0 = lane; k0 < bl; k0 += 256) {
        double xq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int si = basic + (k0 + 64 * q) - 1;
          si = si < 0 ? 0 : (si > u.yl - 1 ? u.yl - 1 : si);
          xq[q] = y[si];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = k0 + 64 * q;
          if (k < bl) {
            const double xv = xq[q];
            const double mw = win(cs);
            const double up = win(cs * dc1 - sn * ds1), dn = win(cs * dc1 + sn * ds1);  // mw[k + 1], mw[k - 1]
            double dw;
            if (k == 0) dw = -up / 2.0;
            else if (k == bl - 1) dw = dn / 2.0;
            else dw = -(up - dn) / 2.0;
            wsamp[k] = make_double2(xv * mw, xv * dw);
            const double c2 = cs * rc - sn * rs;
            sn = sn * rc + cs * rs;
            cs = c2;
          }
        }
      }
    }
    const int nh = min((int)(fs / 2.0 / f0), 6);
    const int idx = min(mround(f0 * fft / fs * (hh + 1)), fft / 2);
    double mr = 0.0, mi = 0.0, dr = 0.0, di = 0.0;
    if (hh < nh) {
      // factor of sample k and bin idx: entry (idx k mod fft) * tstride of the circle, walked as a byte
      // offset; four samples per trip with their LDS reads in flight together, the sums keep their
      // order in k
      const uint32_t maskb = (uint32_t)p.fft_max * 16u - 1u;
      const uint32_t stepb = (uint32_t)(((idx * 8) & (fft - 1)) * tstride) * 16u;
      uint32_t mb = (uint32_t)(((idx * g) & (fft - 1)) * tstride) * 16u;
      const char* twb = reinterpret_cast<const char*>(tw);
      auto factor = [&](uint32_t ofs) { return *reinterpret_cast<const double2*>(twb + ofs); };
      int k = g;
      for (; k + 24 < bl; k += 32) {
        const uint32_t m1 = (mb + stepb) & maskb, m2 = (m1 + stepb) & maskb, m3 = (m2 + stepb) & maskb;
        const double2 w0 = factor(mb), w1 = factor(m1), w2 = factor(m2), w3 = factor(m3);
        const double2 s0 = wsamp[k], s1 = wsamp[k + 8], s2 = wsamp[k + 16], s3 = wsamp[k + 24];
        mb = (m3 + stepb) & maskb;
        mr += s0.x * w0.x; mi += s0.x * w0.y; dr += s0.y * w0.x; di += s0.y * w0.y;
        mr += s1.x * w1.x; mi += s1.x * w1.y; dr += s1.y * w1.x; di += s1.y * w1.y;
        mr += s2.x * w2.x; mi += s2.x * w2.y; dr += s2.y * w2.x; di += s2.y * w2.y;
        mr += s3.x * w3.x; mi += s3.x * w3.y; dr += s3.y * w3.x; di += s3.y * w3.y;
      }
      for (; k < bl; k += 8) {
        const double2 w0 = factor(mb);
        const double2 s0 = wsamp[k];
        mb = (mb + stepb) & maskb;
        mr += s0.x * w0.x; mi += s0.x * w0.y; dr += s0.y * w0.x; di += s0.y * w0.y;
      }
    }
#pragma unroll
    for (int mask = 1; mask < 8; mask <<= 1) {
      mr = hv_xor_sum(mr, mask);
      mi = hv_xor_sum(mi, mask);
      dr = hv_xor_sum(dr, mask);
      di = hv_xor_sum(di, mask);
    }
    double num = 0.0, den = 0.0, sc = 0.0;
    if (hh < nh) {
      const double ps = mr * mr + mi * mi;
      const double ni = mr * di - mi * dr;
      const double inst = ps == 0.0 ? 0.0 : (double)idx * fs / fft + ni / ps * fs / 2.0 / kPi;
      const double amp = sqrt(ps);
      num = amp * inst;
      den = amp * (hh + 1.0);
      sc = fabs((inst / (hh + 1.0) - f0) / f0);
    }
#pragma unroll
    for (int mask = 8; mask < 64; mask <<= 1) {
      num = hv_xor_sum(num, mask);
      den = hv_xor_sum(den, mask);
      sc = hv_xor_sum(sc, mask);
    }
    double rf = num / (den + kEps);
    double rs = 1.0 / (sc / nh + kEps);
    if (rf < p.f0_floor || rf > p.f0_ceil || rs < 2.5) {
      rf = 0.0;
      rs = 0.0;
    }
    if (lane == 0) {
      crow[j] = rf;
      srow[j] = rs;
    }
  }
  }
}

// ---- RemoveUnreliableCandidates ---------------------------------------------------------------------
constexpr int HV_RM_FRAMES = 32;
__global__ __launch_bounds__(NT) void hv_remove_kernel(const HvUtt* __restrict__ utts, HvParams p,
                                                       const int* __restrict__ ncand,
                                                       const double* __restrict__ cand,
                                                       const double* __restrict__ score,
                                                       double* __restrict__ cand2,
                                                       double* __restrict__ score2) {
  extern __shared__ __attribute__((aligned(16))) char smem_rm[];
  double* rows = reinterpret_cast<double*>(smem_rm);  // [HV_RM_FRAMES + 2][maxc]
  const HvUtt u = utts[blockIdx.y];
  const int i0 = blockIdx.x * HV_RM_FRAMES;
  if (i0 >= u.T1) return;
  const int nc = ncand[blockIdx.y] * 7;
  const double* cin = cand + u.cand_off;
  const double* sin_ = score + u.cand_off;
  for (int k = threadIdx.x; k < (HV_RM_FRAMES + 2) * nc; k += NT) {
    const int rr = k / nc, j = k - rr * nc;
    const int fr = i0 - 1 + rr;
    rows[rr * p.maxc + j] = (fr >= 0 && fr < u.T1) ? cin[(int64_t)fr * p.maxc + j] : 0.0;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < HV_RM_FRAMES * nc; k += NT) {
    const int rr = k / nc, j = k - rr * nc;
    const int fr = i0 + rr;
    if (fr >= u.T1) break;
    const double ref = rows[(rr + 1) * p.maxc + j];
    double keep_c = ref, keep_s = sin_[(int64_t)fr * p.maxc + j];
    if (fr >= 1 && fr < u.T1 - 1 && ref != 0.0) {
      double e1 = 1.0, e2 = 1.0;
      const double* nxt = rows + (rr + 2) * p.maxc;
      const double* prv = rows + rr * p.maxc;
      for (int q = 0; q < nc; ++q) {
        const double t1 = fabs(ref - nxt[q]) / ref;
        if (!(t1 > e1)) e1 = t1;
        const double t2 = fabs(ref - prv[q]) / ref;
        if (!(t2 > e2)) e2 = t2;
      }
      if (!(fmin(e1, e2) <= 0.05)) {
        keep_c = 0.0;
        keep_s = 0.0;
      }
    }
    cand2[u.cand_off + (int64_t)fr * p.maxc + j] = keep_c;
    score2[u.cand_off + (int64_t)fr * p.maxc + j] = keep_s;
  }
}

// ---- contour fixing ------------------------------------------------------------------------------------
// GetBoundaryList: positions where voicing flips, with the first and last frame taken as unvoiced;
// boundary k is stored as i - (k & 1) (ends point at the last voiced frame).  Block-wide; returns
// the number of boundaries in every thread.
template <typename F>
__device__ inline int hv_boundaries(F voiced, int T, int* bl, int* sh) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (threadIdx.x == 0) sh[4] = 0;
  __syncthreads();
  for (int i0 = 1; i0 < T; i0 += NT) {
    const int i = i0 + threadIdx.x;
    bool flag = false;
    if (i < T) {
      const int a = (i - 1 == 0) ? 0 : (voiced(i - 1) ? 1 : 0);
      const int b = (i == T - 1) ? 0 : (voiced(i) ? 1 : 0);
      flag = a != b;
    }
    const unsigned long long bal = __ballot(flag);
    if (lane == 0) sh[wv] = __popcll(bal);
    __syncthreads();
    if (flag) {
      int off = sh[4];
      for (int q = 0; q < wv; ++q) off += sh[q];
      const int slot = off + __popcll(bal & ((1ull << lane) - 1ull));
      bl[slot] = i - (slot & 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) sh[4] += sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
  }
  return sh[4];
}

// contour scratch of an utterance (doubles): A[T1] B[T1] C[T1] secsum[T1/2+2]; then ints:
// ob[T1+2] (boundaries of the step-2 contour), sb[T1+2] (after extension), ord[T1/2+2], meta[8]
struct HvCtr {
  double *A, *B, *C, *secsum;
  int *ob, *sb, *ord, *meta;
};
__device__ inline HvCtr hv_ctr(double* ctr, const HvUtt& u) {
  HvCtr c;
  const int T = u.T1, h = T / 2 + 2;
  c.A = ctr + u.ctr_off;
  c.B = c.A + T;
  c.C = c.B + T;
  c.secsum = c.C + T;
  c.ob = reinterpret_cast<int*>(c.secsum + h);
  c.sb = c.ob + (T + 2);
  c.ord = c.sb + (T + 2);
  c.meta = c.ord + h;
  return c;
}
static inline int64_t hv_ctr_doubles(int T1) {
  const int64_t h = T1 / 2 + 2;
  const int64_t ints = 2 * ((int64_t)T1 + 2) + h + 8;
  return 3 * (int64_t)T1 + h + (ints + 1) / 2 + 2;
}

// value of section s's channel at frame j: stored for [ob_st - EXT, ob_ed + EXT], zero elsewhere
__device__ __forceinline__ int64_t hv_mc_index(int s, int j) { return (int64_t)s * HV_SECW + j + HV_EXT; }

// SearchF0Base, FixStep1 (0.008), FixStep2 (6 frames), the sections of the result
__global__ __launch_bounds__(NT) void hv_contour1_kernel(const HvUtt* __restrict__ utts, HvParams p,
                                                         const int* __restrict__ ncand,
                                                         const double* __restrict__ cand2,
                                                         const double* __restrict__ score2,
                                                         double* __restrict__ ctr, double* __restrict__ mc) {
  __shared__ int sh[8];
  const HvUtt u = utts[blockIdx.x];
  const HvCtr c = hv_ctr(ctr, u);
  const int T = u.T1, nc = ncand[blockIdx.x] * 7;
  const double* cd = cand2 + u.cand_off;
  const double* sc = score2 + u.cand_off;
  for (int i = threadIdx.x; i < T; i += NT) {
    double bs = 0.0, bv = 0.0;
    for (int j = 0; j < nc; ++j) {
      const double s = sc[(int64_t)i * p.maxc + j];
      if (s > bs) {
        bv = cd[(int64_t)i * p.maxc + j];
        bs = s;
      }
    }
    c.A[i] = bv;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T; i += NT) {
    double v = 0.0;
    if (i >= 2 && c.A[i] != 0.0) {
      const double ref = c.A[i - 1] * 2 - c.A[i - 2];
      v = (fabs((c.A[i] - ref) / ref) > 0.008 && fabs(c.A[i] - c.A[i - 1]) / c.A[i - 1] > 0.008)
              ? 0.0
              : c.A[i];
    }
    c.B[i] = v;
  }
  __syncthreads();
  int nb = hv_boundaries([&](int i) { return c.B[i] > 0; }, T, c.ob, sh);
  for (int s = threadIdx.x; s < nb / 2; s += NT) {
    if (c.ob[2 * s + 1] - c.ob[2 * s] >= 6) continue;
    for (int j = c.ob[2 * s]; j <= c.ob[2 * s + 1]; ++j) c.B[j] = 0.0;
  }
  __syncthreads();
  nb = hv_boundaries([&](int i) { return c.B[i] > 0; }, T, c.ob, sh);
  const int ns = nb / 2;
  if (threadIdx.x == 0) c.meta[0] = ns;
  // channels: section s holds its own frames, zero elsewhere (the store was zero-filled)
  double* m = mc + u.mc_off;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int s = wv; s < ns; s += NT / 64)
    for (int j = c.ob[2 * s] + lane; j <= c.ob[2 * s + 1]; j += 64) m[hv_mc_index(s, j)] = c.B[j];
}

// SelectBestF0 across a wave: the candidate with the smallest relative distance to `ref` within
// `allowed`; among equals the later one (the sequential scan replaces on <=).
__device__ inline double hv_select_best(double ref, const double* __restrict__ row, int nc,
                                        double allowed) {
  const int lane = threadIdx.x & 63;
  double err = allowed;
  int idx = -1;
  for (int j = lane; j < nc; j += 64) {
    const double t = fabs(ref - row[j]) / ref;
    if (t > err) continue;
    err = t;
    idx = j;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double oe = __shfl_xor(err, off, 64);
    const int oi = __shfl_xor(idx, off, 64);
    const bool take = (oi >= 0) && (idx < 0 || oe < err || (oe == err && oi > idx));
    if (take) {
      err = oe;
      idx = oi;
    }
  }
  return idx >= 0 ? row[idx] : 0.0;
}

// Extend: every section grows forwards then backwards through the candidates (threshold 0.18,
// stops after 4 misses or 100 frames).  grid (groups, utterances); one wave per section.
__global__ __launch_bounds__(NT) void hv_extend_kernel(const HvUtt* __restrict__ utts, HvParams p,
                                                       const int* __restrict__ ncand,
                                                       const double* __restrict__ cand2,
                                                       double* __restrict__ ctr, double* __restrict__ mc) {
  const HvUtt u = utts[blockIdx.y];
  const HvCtr c = hv_ctr(ctr, u);
  const int T = u.T1, nc = ncand[blockIdx.y] * 7, ns = c.meta[0];
  const double* cd = cand2 + u.cand_off;
  double* m = mc + u.mc_off;
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), nw = gridDim.x * (NT / 64);
  for (int s = w; s < ns; s += nw) {
    const int st = c.ob[2 * s], ed = c.ob[2 * s + 1];
    double total = 0.0;
    for (int j = st + lane; j <= ed; j += 64) total += m[hv_mc_index(s, j)];
    total = wave_sum(total);
    int bnd[2];
    double edge_val[2];
    for (int dir = 0; dir < 2; ++dir) {
      const int shift = dir == 0 ? 1 : -1;
      const int origin = dir == 0 ? ed : st;
      const int last = dir == 0 ? min(T - 2, ed + 100) : max(1, st - 100);
      const int distance = abs(last - origin);
      double tmp = m[hv_mc_index(s, origin)];
      int shifted = origin, count = 0;
      double at_shifted = tmp;
      for (int i = 0; i <= distance; ++i) {
        const int at = origin + shift * i + shift;
        const double v = hv_select_best(tmp, cd + (int64_t)at * p.maxc, nc, 0.18);
        if (lane == 0) m[hv_mc_index(s, at)] = v;
        if (v == 0.0) {
          count++;
        } else {
          tmp = v;
          count = 0;
          shifted = at;
          at_shifted = v;
          total += v;
        }
        if (count == 4) break;
      }
      bnd[dir] = shifted;
      edge_val[dir] = at_shifted;
    }
    if (lane == 0) {
      c.sb[2 * s] = bnd[1];
      c.sb[2 * s + 1] = bnd[0];
      c.secsum[s] = total - edge_val[0];  // sum over [st', ed')
    }
  }
}

// ExtendSub + MergeF0 + FixStep4, then the sections of the padded contour for the smoothing
__global__ __launch_bounds__(NT) void hv_contour2_kernel(const HvUtt* __restrict__ utts, HvParams p,
                                                         const int* __restrict__ ncand,
                                                         const double* __restrict__ cand2,
                                                         const double* __restrict__ score2,
                                                         double* __restrict__ ctr,
                                                         const double* __restrict__ mc,
                                                         double* __restrict__ dbg_best) {
  __shared__ int sh[8];
  __shared__ int s_cnt, s_b0, s_b1, s_mode, s_o;
  __shared__ double red[8];
  const HvUtt u = utts[blockIdx.x];
  const HvCtr c = hv_ctr(ctr, u);
  const int T = u.T1, nc = ncand[blockIdx.x] * 7, ns = c.meta[0];
  const double* cd = cand2 + u.cand_off;
  const double* sc = score2 + u.cand_off;
  const double* m = mc + u.mc_off;
  // c.ord[k]: original section behind selected channel k (ExtendSub's swaps); c.sb is compacted
  // alongside.  The published loop carries `mean` from section to section.
  if (threadIdx.x == 0) {
    int count = 0;
    double mean = 0.0;
    for (int s = 0; s < ns; ++s) {
      const int st = c.sb[2 * s], ed = c.sb[2 * s + 1];
      mean += c.secsum[s];
      mean /= ed - st;
      if (2200.0 / mean < ed - st) {
        const int t0 = c.sb[2 * count], t1 = c.sb[2 * count + 1];
        c.sb[2 * count] = st;
        c.sb[2 * count + 1] = ed;
        c.sb[2 * s] = t0;
        c.sb[2 * s + 1] = t1;
        c.ord[count] = s;
        count++;
      }
    }
    s_cnt = count;
  }
  __syncthreads();
  const int count = s_cnt;
  auto mcget = [&](int k, int j) -> double {  // selected channel k at frame j
    const int s = c.ord[k];
    if (j < c.ob[2 * s] - HV_EXT || j > c.ob[2 * s + 1] + HV_EXT) return 0.0;
    return m[hv_mc_index(s, j)];
  };
  double* out = c.C;
  if (count == 0) {
    for (int i = threadIdx.x; i < T; i += NT) out[i] = c.B[i];
  } else {
    // MakeSortedOrder (as published) into A's storage reinterpreted as ints
    int* order = reinterpret_cast<int*>(c.A);
    if (threadIdx.x == 0) {
      for (int i = 0; i < count; ++i) order[i] = i;
      for (int i = 1; i < count; ++i)
        for (int j = i - 1; j >= 0; --j)
          if (c.sb[order[j] * 2] > c.sb[order[i] * 2]) {
            const int t = order[i];
            order[i] = order[j];
            order[j] = t;
          } else {
            break;
          }
    }
    for (int i = threadIdx.x; i < T; i += NT) out[i] = mcget(0, i);
    __syncthreads();
    for (int k = 1; k < count; ++k) {
      if (threadIdx.x == 0) {
        const int o = order[k];
        s_o = o;
        s_b0 = c.sb[0];
        s_b1 = c.sb[1];
        s_mode = (c.sb[o * 2] - c.sb[1] > 0) ? 0 : 1;
      }
      __syncthreads();
      const int o = s_o, st2 = c.sb[o * 2], ed2 = c.sb[o * 2 + 1], st1 = s_b0, ed1 = s_b1;
      if (s_mode == 0) {
        for (int j = st2 + threadIdx.x; j <= ed2; j += NT) out[j] = mcget(o, j);
        __syncthreads();
        if (threadIdx.x == 0) {
          c.sb[0] = st2;
          c.sb[1] = ed2;
        }
      } else if (st1 <= st2 && ed1 >= ed2) {
        __syncthreads();
        // boundary_list[1] = ed1: unchanged
      } else {
        double s1 = 0.0, s2 = 0.0;
        for (int i = st2 + threadIdx.x; i <= ed1; i += NT) {
          const double f1 = out[i], f2 = mcget(o, i);
          double a = 0.0, b = 0.0;
          for (int q = 0; q < nc; ++q) {
            const double cv = cd[(int64_t)i * p.maxc + q], sv = sc[(int64_t)i * p.maxc + q];
            if (f1 == cv && a < sv) a = sv;
            if (f2 == cv && b < sv) b = sv;
          }
          s1 += a;
          s2 += b;
        }
        s1 = bsum(s1, red);
        s2 = bsum(s2, red);
        __syncthreads();
        const int from = s1 > s2 ? ed1 : st2;
        for (int i = from + threadIdx.x; i <= ed2; i += NT) out[i] = mcget(o, i);
        __syncthreads();
        if (threadIdx.x == 0) c.sb[1] = ed2;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  // FixStep4: bridge gaps shorter than 9 frames
  int* bl = c.ob;
  const int nb = hv_boundaries([&](int i) { return out[i] > 0; }, T, bl, sh);
  for (int i = threadIdx.x; i < T; i += NT) c.B[i] = out[i];
  __syncthreads();
  for (int s = threadIdx.x; s < nb / 2 - 1; s += NT) {
    const int distance = bl[(s + 1) * 2] - bl[s * 2 + 1] - 1;
    if (distance >= 9) continue;
    const double t0 = out[bl[s * 2 + 1]] + 1;
    const double t1 = out[bl[(s + 1) * 2]] - 1;
    const double co = (t1 - t0) / (distance + 1.0);
    int cnt = 1;
    for (int j = bl[s * 2 + 1] + 1; j <= bl[(s + 1) * 2] - 1; ++j) c.B[j] = t0 + co * cnt++;
  }
  __syncthreads();
  if (dbg_best)
    for (int i = threadIdx.x; i < T; i += NT) dbg_best[i] = c.B[i];
  // sections of the contour padded with HV_LAG zeros either side (positions, not frames)
  const int n = T + 2 * HV_LAG;
  const int nb2 = hv_boundaries(
      [&](int i) { return i >= HV_LAG && i < HV_LAG + T && c.B[i - HV_LAG] > 0; }, n, c.sb, sh);
  if (threadIdx.x == 0) c.meta[1] = nb2 / 2;
  for (int i = threadIdx.x; i < T; i += NT) c.C[i] = c.B[i];  // smoothed contour starts as a copy
}

// SmoothF0Contour: each voiced section, held constant beyond its ends, through the 2nd-order
// low-pass forwards and backwards.  WORLD runs both passes over the whole padded contour from a
// zero state; the filter's poles have radius 0.875, so HV_LAG = 300 samples of the constant
// extension bring the state to its fixed point to 4e-18 relative -- the passes here cover the
// section plus HV_LAG positions either side (always inside the padded contour).  Sections are few
// and long (hundreds to thousands of frames), so one wave takes a section at a time and splits
// each pass into 64 chunks: every lane runs its chunk from a zero state, the chunk states are
// chained through C^L (the recurrence is linear, as in hv_iir3_reversed), and every lane runs its
// chunk again from its true state.
struct HvSt2 { double w0, w1; };

template <typename In, typename Out>
__device__ inline void hv_iir2_wave(In in, Out out, int n, int lane) {
  const double b0 = 0.0078202080334971724, b1 = 0.015640416066994345;
  const double a0 = 1.7347257688092754, a1 = -0.76600660094326412;
  const int L = (n + 63) / 64;
  const int lo = min(n, lane * L), hi = min(n, lo + L);
  // C^L by running the homogeneous recurrence from the two unit states (every lane the same)
  double m00 = 1.0, m10 = 0.0, m01 = 0.0, m11 = 1.0;   // columns: images of (1,0) and (0,1)
  for (int i = 0; i < L; ++i) {
    const double t0 = a0 * m00 + a1 * m10, t1 = a0 * m01 + a1 * m11;
    m10 = m00;
    m11 = m01;
    m00 = t0;
    m01 = t1;
  }
  double w0 = 0.0, w1 = 0.0;
  for (int i = lo; i < hi; ++i) {
    const double wt = in(i) + a0 * w0 + a1 * w1;
    w1 = w0;
    w0 = wt;
  }
  // chain: state entering chunk k = C^L (state entering k-1) + local(k-1); lanes whose chunk is
  // short or empty only exist at the end, where nothing reads their successor
  double s0 = 0.0, s1 = 0.0, i0 = 0.0, i1 = 0.0;
  for (int k = 0; k < 64; ++k) {
    if (lane == k) {
      i0 = s0;
      i1 = s1;
    }
    const double l0 = __shfl(w0, k, 64), l1 = __shfl(w1, k, 64);
    const double n0 = m00 * s0 + m01 * s1 + l0, n1 = m10 * s0 + m11 * s1 + l1;
    s0 = n0;
    s1 = n1;
  }
  w0 = i0;
  w1 = i1;
  for (int i = lo; i < hi; ++i) {
    const double wt = in(i) + a0 * w0 + a1 * w1;
    out(i, b0 * wt + b1 * w0 + b0 * w1);
    w1 = w0;
    w0 = wt;
  }
}

__global__ __launch_bounds__(64) void hv_smooth_kernel(const HvUtt* __restrict__ utts, HvParams p,
                                                       double* __restrict__ ctr,
                                                       double* __restrict__ park) {
  const HvUtt u = utts[blockIdx.x];
  const HvCtr c = hv_ctr(ctr, u);
  const int ns = c.meta[1];
  double* pk = park + u.sm_off;
  const int lane = threadIdx.x;
  for (int s = 0; s < ns; ++s) {
    const int st = c.sb[2 * s], ed = c.sb[2 * s + 1];
    const int len = ed - st + 1, n = len + 2 * HV_LAG;
    const double cst = c.B[st - HV_LAG], ced = c.B[ed - HV_LAG];
    // forward over offsets k - HV_LAG relative to st; everything is parked (k = 0 .. n - 1)
    hv_iir2_wave(
        [&](int k) {
          const int o = k - HV_LAG;
          return o <= 0 ? cst : (o >= len ? ced : c.B[st + o - HV_LAG]);
        },
        [&](int k, double v) { pk[k] = v; }, n - HV_LAG + HV_LAG, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // backward: index k counts down from the last parked value
    hv_iir2_wave([&](int k) { return pk[n - 1 - k]; },
                 [&](int k, double v) {
                   const int o = n - 1 - k - HV_LAG;
                   if (o >= 0 && o < len) c.C[st + o - HV_LAG] = v;
                 },
                 n, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

// Harvest(): the requested frame period picks from the 1 ms contour
__global__ __launch_bounds__(NT) void hv_pick_kernel(const HvUtt* __restrict__ utts, HvParams p,
                                                     const double* __restrict__ ctr,
                                                     double* __restrict__ f0_out) {
  const HvUtt u = utts[blockIdx.y];
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= u.T) return;
  const double* sm = ctr + u.ctr_off + 2 * (int64_t)u.T1;  // HvCtr::C
  const double tp = i * p.frame_period / 1000.0;
  f0_out[u.f_off + i] = sm[min(u.T1 - 1, mround(tp * 1000.0))];
}

}  // namespace itts

using namespace itts;

extern "C" int64_t itts_harvest_num_frames(int64_t n_samples, int fs, double frame_period_ms) {
  return (int64_t)(1000.0 * (double)n_samples / fs / frame_period_ms) + 1;
}

extern "C" int itts_harvest(const double* d_x, const int64_t* h_x_off, const int64_t* h_f_off,
                            int n_utts, int fs, double frame_period_ms, double f0_floor,
                            double f0_ceil, double* d_f0, double* d_dbg_raw, double* d_dbg_cand,
                            double* d_dbg_score, double* d_dbg_best, void* stream) {
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
  ITTS_REQUIRE(n_utts >= 0, "n_utts must be >= 0");
  if (n_utts == 0) return ITTS_OK;
  ITTS_REQUIRE(fs > 0 && frame_period_ms > 0, "fs and frame_period_ms must be positive");
  ITTS_REQUIRE(f0_floor > 0 && f0_ceil > f0_floor, "need 0 < f0_floor < f0_ceil");
  ITTS_REQUIRE(!(d_dbg_raw || d_dbg_cand || d_dbg_score || d_dbg_best) || n_utts == 1,
               "the per-stage outputs are for a single utterance");
  hipStream_t s = as_stream(stream);
  itts::ScratchScope scratch_scope(s);
  DeviceContext* ctx = get_context();
  if (!ctx) return ITTS_E_HIP;

  auto mr = [](double x) { return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5); };
  HvParams p{};
  p.fs = fs;
  p.r = std::max(std::min(mr(fs / 8000.0), 12), 1);
  p.lag = (int)(std::ceil(140.0 / p.r) * p.r);
  p.afs = (double)fs / p.r;
  p.frame_period = frame_period_ms;
  p.f0_floor = f0_floor;
  p.f0_ceil = f0_ceil;
  const double af = f0_floor * 0.9, ac = f0_ceil * 1.1;
  p.nch = 1 + (int)(std::log(ac / af) / std::log(2.0) * 40.0);
  ITTS_REQUIRE(p.nch >= 12 && p.nch <= HV_MAXCH, "f0 range gives an unsupported number of channels");
  p.ntiles = (p.nch + 15) / 16;
  p.nbase = mr(p.nch / 10.0);
  p.maxc = p.nbase * 7;
  for (int k = 0; k < 3; ++k) p.da[k] = DEC_A[p.r][k];
  for (int k = 0; k < 2; ++k) p.db[k] = DEC_B[p.r][k];

  std::vector<HvTables> tabv(1);
  HvTables& tab = tabv[0];
  for (int c = 0; c < p.nch; ++c) {
    tab.bnd[c] = af * std::pow(2.0, (c + 1) / 40.0);
    tab.half[c] = mr(p.afs / tab.bnd[c] * 2.0);
  }
  p.pad = tab.half[0] + 8;
  // refinement window limits (candidates lie in [f0_floor, f0_ceil])
  {
    const int hw = (int)(1.5 * p.afs / f0_floor + 1.0);
    p.bl_max = 2 * hw + 1;
    p.log_fft_max = 2 + (int)(std::log(hw * 2.0 + 1.0) / std::log(2.0));
    p.fft_max = 1 << p.log_fft_max;
    ITTS_REQUIRE(p.log_fft_max <= 12, "f0_floor too low for the refinement window at this rate");
  }
  // taps: per tile [kt][16], centred at hmax
  std::vector<double> wts;
  for (int t = 0; t < p.ntiles; ++t) {
    const int hmax = tab.half[16 * t];
    const int kt = (2 * hmax + 1 + 3) / 4 * 4;
    tab.hmax[t] = hmax;
    tab.kt[t] = kt;
    tab.woff[t] = (int)wts.size();
    wts.resize(wts.size() + (size_t)kt * 16, 0.0);
    double* W = wts.data() + tab.woff[t];
    for (int cc = 0; cc < 16; ++cc) {
      const int c = 16 * t + cc;
      if (c >= p.nch) break;
      const int h = tab.half[c], n = 2 * h + 1;
      for (int d = -h; d <= h; ++d) {
        const double tt = (double)(d + h) / (n - 1.0);
        const double nut = 0.355768 - 0.487396 * std::cos(2.0 * M_PI * tt) +
                           0.144232 * std::cos(4.0 * M_PI * tt) - 0.012604 * std::cos(6.0 * M_PI * tt);
        W[(size_t)(d + hmax) * 16 + cc] = nut * std::cos(2 * M_PI * tab.bnd[c] * d / p.afs);
      }
    }
  }

  int64_t max_xl = 0;
  for (int u = 0; u < n_utts; ++u) {
    const int64_t xl = h_x_off[u + 1] - h_x_off[u];
    ITTS_REQUIRE(xl > 0 && xl < ((int64_t)1 << 30), "utterance length out of range");
    max_xl = std::max(max_xl, xl);
  }
  {
    const int max_yl = (int)((max_xl + p.r - 1) / p.r);
    const double dur = max_yl / p.afs;
    int64_t off = 0;
    for (int c = 0; c < p.nch; ++c) {
      tab.evcap[c] = std::min(max_yl / 2 + 2, (int)(dur * tab.bnd[c] * 3.0) + 16);
      ITTS_REQUIRE(off < ((int64_t)1 << 31), "utterance too long for the event lists");
      tab.evoff[c] = (int)off;
      off += 4 * (int64_t)tab.evcap[c];
    }
    p.evtot = off;
  }

  HvTables* d_tab = nullptr;
  double* d_wt = nullptr;
  int* d_err = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_tab, sizeof(HvTables), s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_wt, wts.size() * 8, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_err, 4, s));
  ITTS_HIP_CHECK(hipMemcpyAsync(d_tab, &tab, sizeof(HvTables), hipMemcpyHostToDevice, s));
  ITTS_HIP_CHECK(hipMemcpyAsync(d_wt, wts.data(), wts.size() * 8, hipMemcpyHostToDevice, s));
  ITTS_HIP_CHECK(hipMemsetAsync(d_err, 0, 4, s));
  ITTS_HIP_CHECK(itts_spin_sync(s));

  const int64_t budget = (int64_t)12 << 30;
  int u0 = 0;
  while (u0 < n_utts) {
    std::vector<HvUtt> utts;
    int64_t y_n = 0, dec_n = 0, sig_n = 0, ev_n = 0, cnt_n = 0, raw_n = 0, base_n = 0, cand_n = 0,
            ctr_n = 0, mc_n = 0, sm_n = 0;
    int max_yl = 0, max_T1 = 0, max_T = 0;
    int u1 = u0;
    while (u1 < n_utts) {
      const int xl = (int)(h_x_off[u1 + 1] - h_x_off[u1]);
      HvUtt d{};
      d.x_off = h_x_off[u1];
      d.f_off = h_f_off[u1];
      d.xl = xl;
      d.yl = (xl + p.r - 1) / p.r;
      d.T1 = (int)itts_harvest_num_frames(xl, fs, 1.0);
      d.T = (int)itts_harvest_num_frames(xl, fs, frame_period_ms);
      {
        const int sample = d.yl + 5 + 2 * (int)(2.0 * p.afs / tab.bnd[0]);
        d.nfft = (int)std::pow(2.0, (int)(std::log((double)sample) / std::log(2.0)) + 1.0);
      }
      ITTS_REQUIRE(h_f_off[u1 + 1] - h_f_off[u1] == d.T, "frame offsets do not match the frame count");
      ITTS_REQUIRE(d.yl >= 4, "utterance too short");
      d.y_off = y_n; d.dec_off = dec_n; d.sig_off = sig_n; d.ev_off = ev_n; d.cnt_off = cnt_n;
      d.raw_off = raw_n; d.base_off = base_n; d.cand_off = cand_n; d.ctr_off = ctr_n; d.mc_off = mc_n;
      d.sm_off = sm_n;
      const int64_t a_y = d.yl + 2 * p.pad, a_dec = 2 * ((int64_t)xl + 2 * p.lag + 18),
                    a_sig = (int64_t)p.nch * d.yl, a_raw = (int64_t)p.nch * d.T1,
                    a_base = (int64_t)d.T1 * p.nbase, a_cand = (int64_t)d.T1 * p.maxc,
                    a_ctr = hv_ctr_doubles(d.T1),
                    a_mc = ((int64_t)d.T1 / 2 + 2) * HV_SECW / 4 + d.T1 + 2 * HV_SECW,
                    a_sm = ((int64_t)d.T1 + 2 * HV_LAG) * 64;
      // sections are >= 7 frames long and >= 1 apart: at most T1 / 8 of them
      const int64_t bytes = 8 * (y_n + dec_n + sig_n + ev_n + raw_n + base_n + 4 * cand_n + ctr_n + mc_n +
                                 sm_n + a_y + a_dec + a_sig + p.evtot + a_raw + a_base + 4 * a_cand +
                                 a_ctr + a_mc + a_sm);
      if (!utts.empty() && bytes > budget) break;
      y_n += a_y; dec_n += a_dec; sig_n += a_sig; ev_n += p.evtot; cnt_n += p.nch * 4; raw_n += a_raw;
      base_n += a_base; cand_n += a_cand; ctr_n += a_ctr; mc_n += a_mc; sm_n += a_sm;
      max_yl = std::max(max_yl, d.yl); max_T1 = std::max(max_T1, d.T1); max_T = std::max(max_T, d.T);
      utts.push_back(d);
      ++u1;
    }
    const int U = (int)utts.size();
    HvUtt* d_utts = nullptr;
    double *d_y = nullptr, *d_dec = nullptr, *d_sig = nullptr, *d_ev = nullptr, *d_raw = nullptr,
           *d_base = nullptr, *d_cand = nullptr, *d_ctr = nullptr, *d_mc = nullptr, *d_sm = nullptr;
    int *d_cnt = nullptr, *d_nc = nullptr;
    double* d_coef = nullptr;
    double4* d_rot = nullptr;
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_rot, (size_t)(p.bl_max / 2 + 1) * sizeof(double4), s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_coef, (size_t)U * p.nch * 3 * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_utts, U * sizeof(HvUtt), s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_y, y_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_dec, std::max<int64_t>(dec_n, 1) * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_sig, sig_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_ev, ev_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_raw, raw_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_base, base_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_cand, 4 * cand_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_ctr, ctr_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_mc, mc_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_sm, sm_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_cnt, cnt_n * 4, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_nc, U * 4, s));
    ITTS_HIP_CHECK(hipMemcpyAsync(d_utts, utts.data(), U * sizeof(HvUtt), hipMemcpyHostToDevice, s));
    ITTS_HIP_CHECK(hipMemsetAsync(d_nc, 0, U * 4, s));
    ITTS_HIP_CHECK(hipMemsetAsync(d_mc, 0, mc_n * 8, s));
    if (d_dbg_cand || d_dbg_score) ITTS_HIP_CHECK(hipMemsetAsync(d_cand, 0, 4 * cand_n * 8, s));
    ITTS_HIP_CHECK(itts_spin_sync(s));
    double* d_score = d_cand + cand_n;
    double* d_cand2 = d_score + cand_n;
    double* d_score2 = d_cand2 + cand_n;

    hipLaunchKernelGGL(hv_decimate_kernel, dim3(U), dim3(NT), 0, s, d_x, d_utts, p, d_dec, d_y);
    ITTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(hv_mirror_kernel, dim3(U), dim3(NT), 0, s, d_utts, p, d_tab, d_wt, d_y, d_coef);
    ITTS_LAUNCH_CHECK();
    {
      const size_t lds = (size_t)(NT + tab.kt[0] + 4) * 8;
      hipLaunchKernelGGL(hv_bandpass_kernel, dim3((max_yl + NT - 1) / NT, p.ntiles, U), dim3(NT), lds, s,
                         d_utts, p, d_tab, d_wt, d_y, d_coef, d_sig);
      ITTS_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(hv_events_kernel, dim3(p.nch, U), dim3(NT), 0, s, d_utts, p, d_tab, d_sig, d_ev,
                       d_cnt, d_err);
    ITTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(hv_raw_kernel, dim3((max_T1 + NT - 1) / NT, p.nch, U), dim3(NT), 0, s, d_utts, p,
                       d_tab, d_ev, d_cnt, d_raw);
    ITTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(hv_detect_kernel, dim3((max_T1 + NT - 1) / NT, U), dim3(NT), 0, s, d_utts, p, d_raw,
                       d_base, d_nc);
    ITTS_LAUNCH_CHECK();
    {
      const int hw_max = p.bl_max / 2;
      hipLaunchKernelGGL(hv_rot_table_kernel, dim3(hw_max / 64 + 1), dim3(64), 0, s, p.afs, hw_max, d_rot);
      ITTS_LAUNCH_CHECK();
      const size_t lds = (size_t)p.fft_max * 16 + (size_t)HV_REFINE_FRAMES * (p.bl_max + 2) * 16;
      ITTS_REQUIRE(lds <= 160 * 1024, "refinement window does not fit the LDS");
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)hv_refine_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(hv_refine_kernel, dim3((max_T1 + HV_REFINE_FRAMES - 1) / HV_REFINE_FRAMES, U),
                         dim3(NT), lds, s, d_utts, p, ctx->tw_compact[p.log_fft_max], d_rot, d_y, d_base, d_nc,
                         d_cand, d_score);
      ITTS_LAUNCH_CHECK();
    }
    {
      const size_t lds = (size_t)(HV_RM_FRAMES + 2) * p.maxc * 8;
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)hv_remove_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(hv_remove_kernel, dim3((max_T1 + HV_RM_FRAMES - 1) / HV_RM_FRAMES, U), dim3(NT),
                         lds, s, d_utts, p, d_nc, d_cand, d_score, d_cand2, d_score2);
      ITTS_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(hv_contour1_kernel, dim3(U), dim3(NT), 0, s, d_utts, p, d_nc, d_cand2, d_score2,
                       d_ctr, d_mc);
    ITTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(hv_extend_kernel, dim3(8, U), dim3(NT), 0, s, d_utts, p, d_nc, d_cand2, d_ctr, d_mc);
    ITTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(hv_contour2_kernel, dim3(U), dim3(NT), 0, s, d_utts, p, d_nc, d_cand2, d_score2,
                       d_ctr, d_mc, d_dbg_best);
    ITTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(hv_smooth_kernel, dim3(U), dim3(64), 0, s, d_utts, p, d_ctr, d_sm);
    ITTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(hv_pick_kernel, dim3((max_T + NT - 1) / NT, U), dim3(NT), 0, s, d_utts, p, d_ctr,
                       d_f0);
    ITTS_LAUNCH_CHECK();
    if (d_dbg_raw)
      ITTS_HIP_CHECK(hipMemcpyAsync(d_dbg_raw, d_raw, raw_n * 8, hipMemcpyDeviceToDevice, s));
    if (d_dbg_cand)
      ITTS_HIP_CHECK(hipMemcpyAsync(d_dbg_cand, d_cand2, cand_n * 8, hipMemcpyDeviceToDevice, s));
    if (d_dbg_score)
      ITTS_HIP_CHECK(hipMemcpyAsync(d_dbg_score, d_score2, cand_n * 8, hipMemcpyDeviceToDevice, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_utts, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_y, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_dec, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_sig, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_ev, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_raw, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_base, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_cand, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_ctr, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_mc, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_sm, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_cnt, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_nc, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_coef, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_rot, s));
    u0 = u1;
  }
  int64_t* slot = pinned_slot(ctx);
  *slot = 0;
  ITTS_HIP_CHECK(hipMemcpyAsync(slot, d_err, 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(itts_spin_sync(s));
  const bool overflow = *reinterpret_cast<int*>(slot) != 0;
  ITTS_HIP_CHECK(itts::scratch_free(d_tab, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_wt, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_err, s));
  if (overflow) {
    set_error("itts_harvest: a band produced more zero crossings than its event list holds");
    return ITTS_E_INVALID;
  }
  return ITTS_OK;
}
