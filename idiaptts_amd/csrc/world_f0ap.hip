// Frame-parallel WORLD kernels for F0 refinement and aperiodicity: StoneMask and D4C(+LoveTrain).
//
// Replaces the StoneMask and D4C stages of pyworld.wav2world
// (src/data_preparation/world/WorldFeatLabelGen.py:792-793; also LF0LabelGen.py:263-264 for
// stonemask) -- WORLD stonemask.cpp (two-stage refinement) and d4c.cpp (LoveTrain V/UV,
// threshold 0.85).  One 256-thread workgroup per frame.
//
// StoneMask only needs the spectra of the two windowed segments at <= 2 + 6 harmonic bins, so
// the FFTs are replaced by direct DFT sums at those bins (twiddles from the same master table)
// -- no LDS FFT buffer, any f0-dependent FFT size.  D4C keeps every spectrum in LDS.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cmath>

#include "context.h"
#include "world_dev.h"
#include "select_largest.h"

namespace itts {
using namespace wd;

// exp(-2 pi i k / n) for 0 <= k < n from the master table exp(+2 pi i j / TW_N), j < TW_N/2.
// For n > TW_N the angle is computed directly.
__device__ __forceinline__ double2 twiddle_neg(const double2* __restrict__ g_tw, int k, int n) {
  if (n <= TW_N) {
    int j = k * (TW_N / n);
    double sgn = 1.0;
    if (j >= TW_N / 2) {
      j -= TW_N / 2;
      sgn = -1.0;
    }
    const double2 w = g_tw[j];
    return make_double2(sgn * w.x, -sgn * w.y);
  }
  double s, c;
  sincospi(2.0 * (double)k / (double)n, &s, &c);
  return make_double2(c, -s);
}

struct SmArgs {
  const double* x;
  const int64_t* x_off;
  const double* f0_in;
  const int64_t* f_off;
  int n_utts;
  int fs;
  double frame_period;
  double* f0_out;
  const double2* g_tw;
  int nmax;  // LDS capacity (samples) per windowed segment
};

__global__ __launch_bounds__(NT) void stonemask_kernel(SmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* sm = reinterpret_cast<double*>(smem);  // x * main window
  double* sd = sm + a.nmax;                       // x * diff window
  double* mw = sd + a.nmax;                       // main window
  double* red = mw + a.nmax;                      // 4 * 32 doubles
  const int64_t g = blockIdx.x;
  const int u = find_utt_wave(a.f_off, a.n_utts, g);
  const double* x = a.x + a.x_off[u];
  const int64_t xl = a.x_off[u + 1] - a.x_off[u];
  const int fs = a.fs;
  const double f0 = a.f0_in[g];
  const double pos = (double)(g - a.f_off[u]) * a.frame_period / 1000.0;
  if (f0 <= 40.0 || f0 > fs / 12.0) {
    if (threadIdx.x == 0) a.f0_out[g] = 0.0;
    return;
  }
  const int half = (int)(1.5 * fs / f0 + 1.0);
  const double wlt = (2.0 * half + 1.0) / fs;
  const int n = 2 * half + 1;
  const int fft = 1 << (2 + (int)(log(half * 2.0 + 1.0) / log(2.0)));
  const int64_t i0 = mround((pos - (double)half / fs) * fs + 0.001);
  for (int i = threadIdx.x; i < n; i += NT) {
    const double tmp = ((double)(i0 + i) - 1.0) / fs - pos;
    mw[i] = 0.42 + 0.5 * cos(2.0 * kPi * tmp / wlt) + 0.08 * cos(4.0 * kPi * tmp / wlt);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += NT) {
    double dw;
    if (i == 0) dw = -mw[1] / 2.0;
    else if (i == n - 1) dw = mw[n - 2] / 2.0;
    else dw = -(mw[i + 1] - mw[i - 1]) / 2.0;
    int64_t idx = i0 + i - 1;
    idx = idx < 0 ? 0 : (idx > xl - 1 ? xl - 1 : idx);
    const double xv = x[idx];
    sm[i] = xv * mw[i];
    sd[i] = xv * dw;
  }
  __syncthreads();

  // spectra (main M, diff D) at `nh` harmonic bins of `base`: direct DFT sums
  auto refine = [&](double base, int nh) -> double {
    int bins[6];
    double acc[24];
#pragma unroll
    for (int q = 0; q < 24; ++q) acc[q] = 0.0;
#pragma unroll
    for (int hq = 0; hq < 6; ++hq) bins[hq] = hq < nh ? mround(base * fft / fs * (hq + 1)) : 0;
    for (int i = threadIdx.x; i < n; i += NT) {
      const double vm = sm[i], vd = sd[i];
#pragma unroll
      for (int hq = 0; hq < 6; ++hq) {
        if (hq < nh) {
          const int k = (int)(((long long)bins[hq] * i) % fft);
          const double2 w = twiddle_neg(a.g_tw, k, fft);
          acc[4 * hq + 0] += vm * w.x;
          acc[4 * hq + 1] += vm * w.y;
          acc[4 * hq + 2] += vd * w.x;
          acc[4 * hq + 3] += vd * w.y;
        }
      }
    }
    // block reduction of 4*nh values
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 24; ++q) {
      if (q < 4 * nh) {
        const double v = wave_sum(acc[q]);
        if (lane == 0) red[wv * 32 + q] = v;
      }
    }
    __syncthreads();
    double num_sum = 0.0, den_sum = 0.0;
    for (int hq = 0; hq < nh; ++hq) {
      double v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        v[c] = (red[4 * hq + c] + red[32 + 4 * hq + c]) + (red[64 + 4 * hq + c] + red[96 + 4 * hq + c]);
      const double Mr = v[0], Mi = v[1], Dr = v[2], Di = v[3];
      const double numer = Mr * Di - Mi * Dr;
      const double ps = Mr * Mr + Mi * Mi;
      const int idx = bins[hq];
      const double inst = ps == 0.0 ? 0.0 : (double)idx * fs / fft + numer / ps * fs / 2.0 / kPi;
      const double amp = sqrt(ps);
      num_sum += amp * inst;
      den_sum += amp * (hq + 1);
    }
    return num_sum / (den_sum + kEps);
  };

  const double t = refine(f0, 2);
  double mean = 0.0;
  if (!(t <= 0.0 || t > f0 * 2)) {
    int nh = (int)(fs / 2.0 / f0);
    if (nh > 6) nh = 6;
    mean = refine(t, nh);
  }
  if (threadIdx.x == 0) a.f0_out[g] = fabs(mean - f0) > f0 * 0.2 ? f0 : mean;
}

// One WAVE per frame (SM_WAVES frames per workgroup, no workgroup barriers): the windowed and
// derivative-windowed samples go to the wave's LDS block once (window by one sincos per lane and
// rotations), then both refinement passes regroup the lanes as harmonics x sample phases
// (2 x 32 for the first pass, 8 x 8 for the second) so that the spectra at the harmonic bins need
// one short butterfly reduction instead of a workgroup-wide one.  Same sums as stonemask_kernel in
// a different order.
__device__ __forceinline__ double sm_xor_sum(double v, int mask) { return v + __shfl_xor(v, mask, 64); }

// f0_from / f0_to: the launch takes the frames with f0_from < f0 <= f0_to (their windows fit its LDS blocks: a.nmax is
// sized for f0_from); frames outside (40 Hz, fs / 12] get their zero from the launch with f0_to = infinity.
__global__ __launch_bounds__(NT) void stonemask_wave_kernel(SmArgs a, int waves, double f0_from, double f0_to) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (wv >= waves) return;
  const int64_t g = (int64_t)blockIdx.x * waves + wv;
  if (g >= a.f_off[a.n_utts]) return;
  double* sm = reinterpret_cast<double*>(smem) + (size_t)wv * 2 * a.nmax;  // x * main window
  double* sd = sm + a.nmax;                                                  // x * diff window
  const int u = find_utt_wave(a.f_off, a.n_utts, g);
  const double* x = a.x + a.x_off[u];
  const int64_t xl = a.x_off[u + 1] - a.x_off[u];
  const int fs = a.fs;
  const double f0 = a.f0_in[g];
  const double pos = (double)(g - a.f_off[u]) * a.frame_period / 1000.0;
  if (f0 <= 40.0 || f0 > fs / 12.0) {
    if (lane == 0 && f0_to > 1e300) a.f0_out[g] = 0.0;
    return;
  }
  if (!(f0 > f0_from && f0 <= f0_to)) return;      // the other launch's frame
  const int half = (int)(1.5 * fs / f0 + 1.0);
  const double wlt = (2.0 * half + 1.0) / fs;
  const int n = 2 * half + 1;
  const int fft = 1 << (2 + ilog2(n));   // n is odd: floor(log2) is exact
  const int64_t i0 = mround((pos - (double)half / fs) * fs + 0.001);
  {
    const double tmp = ((double)(i0 + lane) - 1.0) / fs - pos;
    double sn, cs, rs, rc, ds1, dc1;
    sincospi(2.0 * tmp / wlt, &sn, &cs);
    sincospi(2.0 * 64.0 / (fs * wlt), &rs, &rc);
    sincospi(2.0 / (fs * wlt), &ds1, &dc1);
    auto win = [](double c) { return 0.42 + 0.5 * c + 0.08 * (2.0 * c * c - 1.0); };
    // four rows of 64 samples per trip, their loads in flight together
    for (int ib = lane; ib < n; ib += 256) {
      double xq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int64_t idx = i0 + (ib + 64 * q) - 1;
        idx = idx < 0 ? 0 : (idx > xl - 1 ? xl - 1 : idx);
        xq[q] = x[idx];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = ib + 64 * q;
        if (i < n) {
          const double xv = xq[q];
          const double mw = win(cs);
          const double up = win(cs * dc1 - sn * ds1), dn = win(cs * dc1 + sn * ds1);
          double dw;
          if (i == 0) dw = -up / 2.0;
          else if (i == n - 1) dw = dn / 2.0;
          else dw = -(up - dn) / 2.0;
          sm[i] = xv * mw;
          sd[i] = xv * dw;
          const double c2 = cs * rc - sn * rs;
          sn = sn * rc + cs * rs;
          cs = c2;
        }
      }
    }
  }
  // spectra (main M, diff D) at `nh` harmonic bins of `base`; lanes = 2^hb harmonics x 2^(6-hb) phases
  auto refine = [&](double base, int nh, int hb) -> double {
    const int pbits = 6 - hb, phases = 1 << pbits;
    const int hq = lane >> pbits, ph = lane & (phases - 1);
    const int bin = hq < nh ? mround(base * fft / fs * (hq + 1)) : 0;
    double mr = 0.0, mi = 0.0, dr = 0.0, di = 0.0;
    if (hq < nh) {
      const int step = (int)(((long long)bin * phases) % fft);
      int k = (int)(((long long)bin * ph) % fft);
      // four samples per trip: their factors (a gather from the 128-KB master table: a trip to the L2
      // each) and samples are requested together; the sums keep their order in i
      int i = ph;
      for (; i + 3 * phases < n; i += 4 * phases) {
        double2 w[4];
        double vm[4], vd[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          w[q] = twiddle_neg(a.g_tw, k, fft);
          vm[q] = sm[i + q * phases];
          vd[q] = sd[i + q * phases];
          k += step;
          if (k >= fft) k -= fft;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          mr += vm[q] * w[q].x;
          mi += vm[q] * w[q].y;
          dr += vd[q] * w[q].x;
          di += vd[q] * w[q].y;
        }
      }
      for (; i < n; i += phases) {
        const double2 w = twiddle_neg(a.g_tw, k, fft);
        const double vm = sm[i], vd = sd[i];
        mr += vm * w.x;
        mi += vm * w.y;
        dr += vd * w.x;
        di += vd * w.y;
        k += step;
        if (k >= fft) k -= fft;
      }
    }
    for (int mask = 1; mask < phases; mask <<= 1) {
      mr = sm_xor_sum(mr, mask);
      mi = sm_xor_sum(mi, mask);
      dr = sm_xor_sum(dr, mask);
      di = sm_xor_sum(di, mask);
    }
    double num = 0.0, den = 0.0;
    if (hq < nh) {
      const double numer = mr * di - mi * dr;
      const double ps = mr * mr + mi * mi;
      const double inst = ps == 0.0 ? 0.0 : (double)bin * fs / fft + numer / ps * fs / 2.0 / kPi;
      const double amp = sqrt(ps);
      num = amp * inst;
      den = amp * (hq + 1);
    }
    for (int mask = phases; mask < 64; mask <<= 1) {
      num = sm_xor_sum(num, mask);
      den = sm_xor_sum(den, mask);
    }
    return num / (den + kEps);
  };
  const double t = refine(f0, 2, 1);
  double mean = 0.0;
  if (!(t <= 0.0 || t > f0 * 2)) {
    int nh = (int)(fs / 2.0 / f0);
    if (nh > 6) nh = 6;
    mean = refine(t, nh, 3);
  }
  if (lane == 0) a.f0_out[g] = fabs(mean - f0) > f0 * 0.2 ? f0 : mean;
}

// ------------------------------------------------------------------------------------------------
struct D4cArgs {
  const double* x;
  const int64_t* x_off;
  const double* f0;
  const int64_t* f_off;
  int n_utts;
  int fs;
  double frame_period;
  int fft_size;   // output resolution (CheapTrick fft size)
  int fftd, logfftd;
  int fftl, logfftl;
  double threshold;
  int nap;
  double* ap;       // [Ttot, K] or nullptr
  double* bap_f64;  // [Ttot, nap] or nullptr
  float* bap_f32;   // [Ttot, ld_bap] or nullptr
  int64_t ld_bap;
  const double2* g_tw;    // compact table of the largest transform (DeviceContext::tw_compact)
  int bmax;
  const int* order;       // frames with f0 != 0 first (one workgroup each), order[t_total] = how many
  int64_t t_total;
  const double* nwin;     // [wl] Nuttall window of the coarse-aperiodicity bands (d4c_nuttall_kernel)
};

// Voiced frames to the front of the launch order, unvoiced ones (which only write constants) to the
// back: interleaved as they come, the instantly finishing unvoiced workgroups leave the CUs
// half empty between dispatches.  cnt[0] / cnt[1] count from both ends; the order inside each
// part does not matter (frames are independent).
__global__ void d4c_order_kernel(const double* __restrict__ f0, int64_t T, int* __restrict__ order,
                                 int* __restrict__ cnt) {
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < T; g += (int64_t)gridDim.x * blockDim.x) {
    if (f0[g] != 0.0)
      order[atomicAdd(&cnt[0], 1)] = (int)g;
    else
      order[T - 1 - atomicAdd(&cnt[1], 1)] = (int)g;
  }
}

struct D4cLds {
  const double2* tw;
  double2* z;
  double* A;   // [h+1]
  double* B;
  double* C;
  double* D;
  double* mir;
  double* red;
};

// The analysis window of a frame, kept in registers: thread t owns the samples t + 256 j.  D4C windows
// a frame's segment six times (LoveTrain, four times for the centroid, once for the power spectrum)
// and every pass needs the window twice (the segment, then the mean removal): evaluated on the spot
// that was 21 n fp64 cosines per frame (n = 4 fs / f0 + 1 samples) -- as many instructions as the seven
// transforms.  Now each distinct window is evaluated once (4.5 n) with the short cosine of fastmath.h.
template <int PER>
struct Win {
  double w[PER];
  int half, n;
};
template <int PER>
__device__ __forceinline__ void win_make(Win<PER>& W, int fs, double f0, int blackman, double ratio) {
  W.half = mround(ratio * fs / f0 / 2.0);
  W.n = 2 * W.half + 1;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = (int)threadIdx.x + j * NT;
    const int b = i - W.half;
    const double p = (2.0 * b / ratio) / fs;
    double w = 0.0;
    if (i < W.n)
      // (|pi p f0| <= pi inside the window: the short cosine of fastmath.h directly -- cos_mid's out-of-line
      // fallback for |x| > 1e5 is a call this kernel never takes)
      w = blackman ? 0.42 + 0.5 * fm::fcos(kPi * p * f0) + 0.08 * fm::fcos(kPi * p * f0 * 2)
                   : 0.5 * fm::fcos(kPi * p * f0) + 0.5;
    W.w[j] = w;
  }
}

// windowed segment written to zr[0..n) (zero padded to `pad`): x*win - win*mean; v keeps this thread's
// values.  normalise: divide by sqrt(sum of squares) (D4C centroid).  `ramp`: multiply sample i by (i+1).
template <int PER>
__device__ inline void windowed_to(const double* __restrict__ x, int64_t xl, int fs, double pos, const Win<PER>& W,
                                   double* zr, int pad, bool normalise, bool ramp, double* red, double (&v)[PER],
                                   double* zr_ramped = nullptr) {
  const int half = W.half, n = W.n;
  const int64_t c = mround(pos * fs + 0.001);
  double swf = 0.0, sw = 0.0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = (int)threadIdx.x + j * NT;
    v[j] = 0.0;
    if (i < n) {
      int64_t idx = c + i - half;
      idx = idx < 0 ? 0 : (idx > xl - 1 ? xl - 1 : idx);
      v[j] = x[idx] * W.w[j];
      swf += v[j];
      sw += W.w[j];
    }
  }
  bsum2(swf, sw, red);
  const double mean = swf / sw;
  double pw = 0.0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = (int)threadIdx.x + j * NT;
    if (i < n) {
      const double u = v[j] - W.w[j] * mean;
      v[j] = u;
      pw += u * u;
    }
  }
  if (normalise) {
    pw = sqrt(bsum(pw, red));
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = (int)threadIdx.x + j * NT;
      if (i < n) v[j] = v[j] / pw;
    }
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = (int)threadIdx.x + j * NT;
    if (i < pad) {
      zr[i] = (ramp && i < n) ? v[j] * (i + 1.0) : v[j];
      if (zr_ramped) zr_ramped[i] = i < n ? v[j] * (i + 1.0) : v[j];      // what ramped_to would write
    }
  }
  for (int i = (int)threadIdx.x + PER * NT; i < pad; i += NT) {
    zr[i] = 0.0;
    if (zr_ramped) zr_ramped[i] = 0.0;
  }
  __syncthreads();
}

// the same segment again, multiplied by the ramp (i + 1): what a second windowed_to(..., ramp = true)
// would recompute from scratch
template <int PER>
__device__ inline void ramped_to(const Win<PER>& W, const double (&v)[PER], double* zr, int pad) {
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = (int)threadIdx.x + j * NT;
    if (i < pad) zr[i] = i < W.n ? v[j] * (i + 1.0) : v[j];
  }
  for (int i = (int)threadIdx.x + PER * NT; i < pad; i += NT) zr[i] = 0.0;
  __syncthreads();
}

// AREG: the centroid accumulates in registers and only two bin arrays live in LDS (needed to get
// two workgroups per CU with the 4096-point transforms of 48 kHz; costs ~60 VGPRs, so the
// 2048-point case keeps three arrays and stays at three workgroups per CU).
// The Nuttall window of the coarse-aperiodicity bands only depends on the transform size and the
// sampling rate: tabulated once per call (three fp64 cosines per sample and band and frame less).
__global__ void d4c_nuttall_kernel(int wl, double* __restrict__ nwin) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < wl; i += gridDim.x * blockDim.x) {
    const double tt = (double)i / (wl - 1.0);
    nwin[i] = 0.355768 - 0.487396 * cos(2.0 * kPi * tt) + 0.144232 * cos(4.0 * kPi * tt) - 0.012604 * cos(6.0 * kPi * tt);
  }
}

// CFFT: log2 of BOTH transform sizes (fftd = fftl = 2^CFFT: 11 at 16 .. 24 kHz) as a compile-time constant, 0: any sizes.
#ifndef D4C_OCC
#define D4C_OCC 3
#endif
template <bool AREG, int CFFT = 0>
__global__ __launch_bounds__(NT, AREG ? 2 : D4C_OCC) void d4c_kernel(D4cArgs a) {
  constexpr int CTS = CFFT ? CFFT - 1 : 0;       // log2 of the twiddle table's size - 1
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PER = AREG ? 16 : 8;           // window samples per thread: n < fftd = 256 PER
  const int fmax = max(a.fftd, a.fftl);
  const int hmax = fmax / 2;
  D4cLds L;
  char* p = smem;
  // LDS is handed out in coarse blocks, so the footprint decides the workgroups per CU in steps:
  // twiddles are read through the cache from a compact global table, D lives in A's storage (A is
  // dead by then), with AREG also C in B's, and the smoothing passes use the idle FFT buffer z for
  // their mirrored copy and scan scratch.  41 KB for the 2048-point transforms (16 kHz: three
  // workgroups per CU), 66 KB for 4096 with AREG (48 kHz: two instead of one).
  L.tw = a.g_tw;
  L.z = reinterpret_cast<double2*>(p);
  p += max((size_t)(hmax + 1) * 16, (size_t)(hmax + 2 * a.bmax + 2) * 8 + (size_t)(NT + 8) * 8);
  L.A = reinterpret_cast<double*>(p); p += (size_t)(hmax + 2) * 8;
  L.B = reinterpret_cast<double*>(p); p += (size_t)(hmax + 2) * 8;
  if (AREG) {
    L.C = L.B;
  } else {
    L.C = reinterpret_cast<double*>(p); p += (size_t)(hmax + 2) * 8;
  }
  L.D = L.A;
  L.mir = reinterpret_cast<double*>(L.z);
  L.red = reinterpret_cast<double*>(p);

  const int64_t g = a.order[blockIdx.x];
  const int u = find_utt_wave(a.f_off, a.n_utts, g);
  const double* x = a.x + a.x_off[u];
  const int64_t xl = a.x_off[u + 1] - a.x_off[u];
  const int fs = a.fs;
  const double f0raw = a.f0[g];
  const double pos = (double)(g - a.f_off[u]) * a.frame_period / 1000.0;
  const int K = a.fft_size / 2 + 1;
  const int nap = a.nap;
  double* zr = reinterpret_cast<double*>(L.z);

  bool voiced = false;
  double coarse[8];
  if (f0raw != 0.0) {
    // ---- LoveTrain: energy ratio 100 Hz..4 kHz over 100 Hz..7.9 kHz
    {
      const int fft = a.fftl;
      const int b0 = (int)ceil(100.0 * fft / fs), b1 = (int)ceil(4000.0 * fft / fs),
                b2 = (int)ceil(7900.0 * fft / fs);
      Win<PER> W;
      double seg[PER];
      win_make(W, fs, f0raw > 40.0 ? f0raw : 40.0, 1, 3.0);
      windowed_to(x, xl, fs, pos, W, zr, fft + 2, false, false, L.red, seg);
      rfft_lds<true, CFFT, CTS>(L.z, fft, a.logfftl, L.tw, fmax);
      double s1 = 0.0, s2 = 0.0;
      for (int k = threadIdx.x; k <= b2; k += NT) {
        if (k > b0) {
          const double2 v = L.z[k];
          const double ps = v.x * v.x + v.y * v.y;
          s2 += ps;
          if (k <= b1) s1 += ps;
        }
      }
      bsum2(s1, s2, L.red);
      voiced = (s1 / s2) > a.threshold;
      __syncthreads();
    }
  }
  if (voiced) {
    const int fft = a.fftd, logfft = a.logfftd, h = fft / 2;
    const double f0 = f0raw > 47.0 ? f0raw : 47.0;  // kFloorF0D4C
    // twiddles in LDS are those of the largest transform (fmax); smaller ones stride through it
    auto rfft = [&](void) { rfft_lds<true, CFFT, CTS>(L.z, fft, logfft, L.tw, fmax); };
    // --- static centroid (two time-shifted analyses): the first spectrum of a side waits in
    // B (re) and C (im) -- with AREG in B and A, the centroid then accumulates in registers (a thread
    // always owns the bins tid + 256 i) and moves to A when both sides are done
    constexpr int APER = 9;                     // (4096 / 2 + 1) / 256 rounded up
    double acc_a[AREG ? APER : 1];
    double* im1 = AREG ? L.A : L.C;
    Win<PER> W;
    double seg[PER];
    win_make(W, fs, f0, 1, 4.0);               // one Blackman window serves all four centroid transforms
    for (int side = 0; side < 2; ++side) {
      const double cpos = side == 0 ? pos - 0.25 / f0 : pos + 0.25 / f0;
      if (!AREG && h <= FFT_PAIR_MAX) {
        // the segment and its ramped copy transformed in LOCKSTEP (the ramped one in B / C's storage,
        // which only held the first spectrum until it was multiplied in): one set of barriers, index
        // arithmetic and twiddle loads for two transforms; same values bit for bit
        double2* z2 = reinterpret_cast<double2*>(L.B);
        windowed_to(x, xl, fs, cpos, W, zr, fft + 2, true, false, L.red, seg, reinterpret_cast<double*>(z2));
        rfft_lds_pair<CFFT, CTS>(L.z, z2, fft, logfft, L.tw, fmax);
        for (int k = threadIdx.x; k <= h; k += NT) {
          const double v = z2[k].x * L.z[k].x + L.z[k].y * z2[k].y;
          L.A[k] = side == 0 ? v : L.A[k] + v;
        }
        __syncthreads();
        continue;
      }
      windowed_to(x, xl, fs, cpos, W, zr, fft + 2, true, false, L.red, seg);
      rfft();
      for (int k = threadIdx.x; k <= h; k += NT) {
        L.B[k] = L.z[k].x;
        im1[k] = L.z[k].y;
      }
      __syncthreads();
      ramped_to(W, seg, zr, fft + 2);
      rfft();
      if (AREG) {
#pragma unroll
        for (int i = 0; i < APER; ++i) {
          const int k = threadIdx.x + i * NT;
          if (k <= h) {
            const double v = L.z[k].x * L.B[k] + im1[k] * L.z[k].y;
            acc_a[i] = side == 0 ? v : acc_a[i] + v;
          }
        }
      } else {
        for (int k = threadIdx.x; k <= h; k += NT) {
          const double v = L.z[k].x * L.B[k] + im1[k] * L.z[k].y;
          L.A[k] = side == 0 ? v : L.A[k] + v;
        }
      }
      __syncthreads();
    }
    if (AREG) {
#pragma unroll
      for (int i = 0; i < APER; ++i) {
        const int k = threadIdx.x + i * NT;
        if (k <= h) L.A[k] = acc_a[i];
      }
      __syncthreads();
    }
    dc_correction(L.A, f0, fs, fft);
    // --- smoothed power spectrum
    win_make(W, fs, f0, 0, 4.0);
    windowed_to(x, xl, fs, pos, W, zr, fft + 2, false, false, L.red, seg);
    rfft();
    for (int k = threadIdx.x; k <= h; k += NT) {
      const double2 v = L.z[k];
      L.B[k] = v.x * v.x + v.y * v.y;
    }
    __syncthreads();
    dc_correction(L.B, f0, fs, fft);
    double* scan = L.mir + (hmax + 2 * a.bmax + 2);     // behind the mirrored copy, inside z
    linear_smoothing(L.B, f0, fs, fft, L.B, L.mir, scan);
    // --- static group delay
    for (int k = threadIdx.x; k <= h; k += NT) L.C[k] = L.A[k] / L.B[k];
    __syncthreads();
    linear_smoothing(L.C, f0 / 2.0, fs, fft, L.C, L.mir, scan);
    linear_smoothing(L.C, f0, fs, fft, L.D, L.mir, scan);
    for (int k = threadIdx.x; k <= h; k += NT) L.C[k] -= L.D[k];
    __syncthreads();
    // --- coarse aperiodicity per 3 kHz band
    const int wl = (int)(3000.0 * fft / fs) * 2 + 1;
    const int boundary = mround(fft * 8.0 / wl);
    const int halfw = wl / 2;
    for (int b = 0; b < nap; ++b) {
      const int center = (int)(3000.0 * (b + 1) * fft / fs);
      for (int i = threadIdx.x; i < fft + 2; i += NT) {
        double v = 0.0;
        if (i <= halfw * 2) {
          v = L.C[center - halfw + i] * a.nwin[i];      // Nuttall window of the band, tabulated per call
        }
        zr[i] = v;
      }
      __syncthreads();
      rfft();
      // power spectrum: each thread keeps its bins (sorted); the boundary + 1 largest of the frame are left
      // out of `rest` (d4c.cpp, D4CGeneralBody's sort)
      constexpr int MPER = AREG ? 9 : 5;  // (fft / 2 + 1) / 256 rounded up
      double mine[MPER];
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < MPER; ++i) {
        const int k = (int)threadIdx.x + i * NT;
        double v = -1.0;
        if (k <= h) {
          const double2 zz = L.z[k];
          v = zz.x * zz.x + zz.y * zz.y;
          ++cnt;
        }
        mine[i] = v;
      }
      // sort descending (tiny insertion sort in registers; the -1 fillers stay behind)
#pragma unroll
      for (int i = 1; i < MPER; ++i) {
#pragma unroll
        for (int j = i; j > 0; --j) {
          if (mine[j] > mine[j - 1]) {
            const double tswap = mine[j];
            mine[j] = mine[j - 1];
            mine[j - 1] = tswap;
          }
        }
      }
      double total = 0.0;
#pragma unroll
      for (int i = 0; i < MPER; ++i)
        if (i < cnt) total += mine[i];
      total = bsum(total, L.red);          // (its barriers: everybody has read z, which now holds the lists)
      const double rest = d4c_rest_without_largest<MPER>(mine, cnt, boundary + 1, zr, L.red);
      const double ca = 10.0 * log10(rest / total);
      const double cv = ca + (f0 - 100.0) / 50.0;
      coarse[b] = cv < 0.0 ? cv : 0.0;
      __syncthreads();
    }
  }
  // ---- outputs: aperiodicity row and / or coded band aperiodicity
  double cfa[8], cap[8];
  for (int b = 0; b <= nap; ++b) cfa[b] = b * 3000.0;
  cfa[nap + 1] = fs / 2.0;
  cap[0] = -60.0;
  for (int b = 0; b < nap; ++b) cap[b + 1] = voiced ? coarse[b] : 0.0;
  cap[nap + 1] = -kEps;
  auto ap_at = [&](int k) -> double {
    if (!voiced) return 1.0 - kEps;
    const double f = (double)k * fs / a.fft_size;
    int kk = 1;
    while (kk < nap + 1 && f >= cfa[kk]) ++kk;
    const double s = (f - cfa[kk - 1]) / (cfa[kk] - cfa[kk - 1]);
    return pow(10.0, (cap[kk - 1] + s * (cap[kk] - cap[kk - 1])) / 20.0);
  };
  if (a.ap)
    for (int k = threadIdx.x; k < K; k += NT) a.ap[g * K + k] = ap_at(k);
  if ((a.bap_f64 || a.bap_f32) && (int)threadIdx.x < nap) {
    const int b = threadIdx.x;
    const double cf = 3000.0 * (b + 1);
    int k = (int)(cf * a.fft_size / fs);
    while (k + 1 < K && (double)(k + 1) * fs / a.fft_size <= cf) ++k;
    while (k > 0 && (double)k * fs / a.fft_size > cf) --k;
    if (k > K - 2) k = K - 2;
    const double x0 = (double)k * fs / a.fft_size, x1 = (double)(k + 1) * fs / a.fft_size;
    // the aperiodicity of the two bins in dB: ap_at's exponent itself (20 log10(10^(y / 20)) went through the
    // library's pow and log10 for every band and frame on one wave while the other three waited: 600 instructions)
    auto db_at = [&](int kb) -> double {
      if (!voiced) return 20.0 * log10(1.0 - kEps);
      const double f = (double)kb * fs / a.fft_size;
      int kk = 1;
      while (kk < nap + 1 && f >= cfa[kk]) ++kk;
      const double sfr = (f - cfa[kk - 1]) / (cfa[kk] - cfa[kk - 1]);
      return cap[kk - 1] + sfr * (cap[kk] - cap[kk - 1]);
    };
    const double y0 = db_at(k), y1 = db_at(k + 1);
    const double v = y0 + (cf - x0) / (x1 - x0) * (y1 - y0);
    if (a.bap_f64) a.bap_f64[g * nap + b] = v;
    if (a.bap_f32) a.bap_f32[g * a.ld_bap + b] = (float)v;
  }
}

// The frames with f0 = 0 -- seven in ten of a speech-like batch --: aperiodicity 1 - eps in every bin, its coded form
// 20 log10(1 - eps) in every band (what d4c_kernel writes for a frame it finds unvoiced).  They used to get a workgroup
// of d4c_kernel each, 41 KB of LDS held for a few loads: over a millisecond of dispatching and retiring at 16 kHz.
__global__ __launch_bounds__(256) void d4c_unvoiced_kernel(D4cArgs a) {
  const int K = a.fft_size / 2 + 1;
  const int n_active = a.order[a.t_total];
  const int64_t n_rest = a.t_total - n_active;
  const double apv = 1.0 - kEps;
  const double bv = 20.0 * log10(1.0 - kEps);
  if (a.ap) {
    // a wave per frame: rows of K values
    const int lane = threadIdx.x & 63;
    for (int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n_rest; r += (int64_t)gridDim.x * 4) {
      const int64_t g = a.order[n_active + r];
      for (int k = lane; k < K; k += 64) a.ap[g * K + k] = apv;
    }
  }
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rest; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = a.order[n_active + r];
    for (int b = 0; b < a.nap; ++b) {
      if (a.bap_f64) a.bap_f64[g * a.nap + b] = bv;
      if (a.bap_f32) a.bap_f32[g * a.ld_bap + b] = (float)bv;
    }
  }
}

}  // namespace itts

using namespace itts;

static int ilog2h(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}

static int check_offsets(const int64_t* h_x_off, const int64_t* h_f_off, int n_utts, int fs,
                         double frame_period_ms) {
  for (int u = 0; u < n_utts; ++u) {
    const int64_t n = h_x_off[u + 1] - h_x_off[u];
    ITTS_REQUIRE(n > 0, "empty utterance");
    ITTS_REQUIRE(h_f_off[u + 1] - h_f_off[u] == itts_world_num_frames(n, fs, frame_period_ms),
                 "frame offsets do not match int(1000*n/fs/frame_period)+1");
  }
  return ITTS_OK;
}

extern "C" int itts_stonemask(const double* d_x, const int64_t* h_x_off, const double* d_f0_in,
                              const int64_t* h_f_off, int n_utts, int fs, double frame_period_ms,
                              double* d_f0_out, void* stream) {
  ITTS_REQUIRE(h_x_off && h_f_off && (n_utts == 0 || (d_x && d_f0_in && d_f0_out)), "null pointer");
  ITTS_REQUIRE(n_utts >= 0 && fs > 0 && frame_period_ms > 0, "bad sizes");
  if (n_utts == 0 || h_f_off[n_utts] == 0) return ITTS_OK;
  int rc = check_offsets(h_x_off, h_f_off, n_utts, fs, frame_period_ms);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  itts::ScratchScope scratch_scope(s);
  DeviceContext* ctx = get_context();
  if (!ctx) return ITTS_E_HIP;
  int64_t *d_xo = nullptr, *d_fo = nullptr;
  if ((rc = upload_i64(h_x_off, n_utts + 1, &d_xo, s))) return rc;
  if ((rc = upload_i64(h_f_off, n_utts + 1, &d_fo, s))) return rc;
  SmArgs a{d_x, d_xo, d_f0_in, d_fo, n_utts, fs, frame_period_ms, d_f0_out, ctx->twiddles, 0};
  a.nmax = 2 * (int)(1.5 * fs / 40.0 + 1.0) + 1 + 3;
  const size_t per_wave = (size_t)a.nmax * 2 * 8;
  if (per_wave <= 80 * 1024) {
    // A wave's LDS block holds the window of its frame: 3 periods of f0.  The interface admits f0 down to 40 Hz
    // (3 606 samples at 48 kHz: 58 KB per wave, one wave per workgroup, two per CU), DIO's floor is 71 Hz: the main
    // launch is sized for f0 > 70 Hz (33 KB per wave at 48 kHz, 11 KB at 16 kHz: twice / 1.5 times the waves per CU)
    // and a second launch with the long blocks takes the frames below -- none, after DIO; its waves then leave at once.
    const double kSplit = 70.0;
    const int64_t T = h_f_off[n_utts];
    static std::atomic<int> lds_attr{0};
    auto launch = [&](double from, double to) -> int {
      SmArgs b = a;
      b.nmax = 2 * (int)(1.5 * fs / from + 1.0) + 1 + 3;
      const size_t pw = (size_t)b.nmax * 2 * 8;
      // as many frames per workgroup as keep two workgroups on a CU
      const int waves = (int)std::max<size_t>(1, std::min<size_t>(4, (80 * 1024) / pw));
      const size_t lds = pw * waves;
      if ((int)lds > lds_attr.load()) {
        ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)stonemask_wave_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(80 * 1024)));
        lds_attr.store(80 * 1024);
      }
      hipLaunchKernelGGL(stonemask_wave_kernel, dim3((unsigned)((T + waves - 1) / waves)), dim3(64 * waves), lds,
                         s, b, waves, from, to);
      ITTS_LAUNCH_CHECK();
      return ITTS_OK;
    };
    if ((rc = launch(kSplit, 1e308))) return rc;
    if ((rc = launch(40.0, kSplit))) return rc;
  } else {
    const size_t lds = (size_t)a.nmax * 3 * 8 + 128 * 8;
    ITTS_REQUIRE(lds <= 160 * 1024, "sampling rate too high for the StoneMask LDS budget");
    ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)stonemask_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(stonemask_kernel, dim3((unsigned)h_f_off[n_utts]), dim3(NT), lds, s, a);
    ITTS_LAUNCH_CHECK();
  }
  ITTS_HIP_CHECK(itts::scratch_free(d_xo, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_fo, s));
  return ITTS_OK;
}

extern "C" int itts_d4c(const double* d_x, const int64_t* h_x_off, const double* d_f0,
                        const int64_t* h_f_off, int n_utts, int fs, double frame_period_ms,
                        int fft_size, double threshold, double* d_ap, double* d_bap_f64,
                        float* d_bap_f32, int64_t ld_bap, void* stream) {
  ITTS_REQUIRE(h_x_off && h_f_off && (n_utts == 0 || (d_x && d_f0)), "null pointer");
  ITTS_REQUIRE(n_utts == 0 || d_ap || d_bap_f64 || d_bap_f32, "nothing to compute");
  ITTS_REQUIRE(n_utts >= 0 && fs >= 15800 && frame_period_ms > 0, "bad sizes (fs >= 15.8 kHz)");
  ITTS_REQUIRE(fft_size > 0 && (fft_size & (fft_size - 1)) == 0, "fft_size must be a power of 2");
  const int nap = itts_num_aperiodicities(fs);
  ITTS_REQUIRE(nap >= 1 && nap <= 5, "unsupported number of aperiodicity bands");
  ITTS_REQUIRE(!d_bap_f32 || ld_bap >= nap, "ld_bap too small");
  if (n_utts == 0 || h_f_off[n_utts] == 0) return ITTS_OK;
  int rc = check_offsets(h_x_off, h_f_off, n_utts, fs, frame_period_ms);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  itts::ScratchScope scratch_scope(s);
  DeviceContext* ctx = get_context();
  if (!ctx) return ITTS_E_HIP;
  D4cArgs a{};
  a.fftd = 1 << (1 + (int)std::log2(4.0 * fs / 47.0 + 1));
  a.fftl = 1 << (1 + (int)std::log2(3.0 * fs / 40.0 + 1));
  ITTS_REQUIRE(std::max(a.fftd, a.fftl) <= 4096, "sampling rate too high for the D4C LDS budget");
  a.logfftd = ilog2h(a.fftd);
  a.logfftl = ilog2h(a.fftl);
  int64_t *d_xo = nullptr, *d_fo = nullptr;
  if ((rc = upload_i64(h_x_off, n_utts + 1, &d_xo, s))) return rc;
  if ((rc = upload_i64(h_f_off, n_utts + 1, &d_fo, s))) return rc;
  a.x = d_x; a.x_off = d_xo; a.f0 = d_f0; a.f_off = d_fo; a.n_utts = n_utts; a.fs = fs;
  a.frame_period = frame_period_ms; a.fft_size = fft_size; a.threshold = threshold; a.nap = nap;
  a.ap = d_ap; a.bap_f64 = d_bap_f64; a.bap_f32 = d_bap_f32; a.ld_bap = ld_bap;
  const int fmax = std::max(a.fftd, a.fftl), hmax = fmax / 2;
  a.g_tw = ctx->tw_compact[ilog2h(fmax)];
  a.bmax = (int)(1200.0 * fmax / fs) + 2;
  // z doubles as the smoothing passes' mirrored copy + scan scratch: it has to hold them
  const size_t z_bytes = std::max((size_t)(hmax + 1) * 16,
                                  (size_t)(hmax + 2 * a.bmax + 2) * 8 + (size_t)(NT + 8) * 8);
  const bool areg = hmax > 1024;            // 4096-point transforms: two bin arrays, see the kernel
  size_t lds = z_bytes + (areg ? 2 : 3) * (size_t)(hmax + 2) * 8 + 32 * 8;
#if D4C_OCC != 3
  if (const char* e = getenv("ITTS_D4C_LAB_LDS")) lds = (size_t)atol(e);   // lab only: timing with a smaller claim (results invalid)
#endif
  ITTS_REQUIRE(lds <= 160 * 1024, "LDS budget exceeded");
  // both transforms of the usual sizes (2048 at 16 .. 24 kHz, 4096 at 44.1 / 48 kHz) have a kernel with the size
  // compiled in (ITTS_D4C_GENERIC=1: the any-size kernel, for the A/B test)
  const bool sized = a.fftd == a.fftl && a.logfftd == (areg ? 12 : 11) && !getenv("ITTS_D4C_GENERIC");
  const void* kern = areg ? (sized ? (const void*)d4c_kernel<true, 12> : (const void*)d4c_kernel<true>)
                          : (sized ? (const void*)d4c_kernel<false, 11> : (const void*)d4c_kernel<false>);
  ITTS_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t t_total = h_f_off[n_utts];
  ITTS_REQUIRE(t_total < ((int64_t)1 << 31), "too many frames in one call");
  int* d_order = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_order, (size_t)(t_total + 2) * sizeof(int), s));
  ITTS_HIP_CHECK(hipMemsetAsync(d_order + t_total, 0, 2 * sizeof(int), s));
  hipLaunchKernelGGL(d4c_order_kernel, dim3((unsigned)std::min<int64_t>((t_total + 255) / 256, 1024)), dim3(256),
                     0, s, d_f0, t_total, d_order, d_order + t_total);
  ITTS_LAUNCH_CHECK();
  a.order = d_order;
  const int wl = (int)(3000.0 * a.fftd / fs) * 2 + 1;
  double* d_nwin = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_nwin, (size_t)wl * 8, s));
  hipLaunchKernelGGL(d4c_nuttall_kernel, dim3((wl + 255) / 256), dim3(256), 0, s, wl, d_nwin);
  ITTS_LAUNCH_CHECK();
  a.nwin = d_nwin;
  a.t_total = t_total;
  // how many frames have f0 != 0: the host asks (a page-locked slot, awaited by polling while the kernel of the other
  // frames and whatever the caller queued on its other streams run) and launches a workgroup for exactly those
  int64_t* h_active = pinned_slot(ctx);
  {
    hipEvent_t ev = nullptr;
    ITTS_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    // (cnt[0] and cnt[1] are ints next to each other: one 8-byte copy, the low word is the count)
    ITTS_HIP_CHECK(hipMemcpyAsync(h_active, d_order + t_total, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    ITTS_HIP_CHECK(hipEventRecord(ev, s));
    hipLaunchKernelGGL(d4c_unvoiced_kernel, dim3((unsigned)std::min<int64_t>((t_total + 255) / 256, 2048)), dim3(256), 0, s, a);
    hipError_t e;
    while ((e = hipEventQuery(ev)) == hipErrorNotReady) {
    }
    (void)hipEventDestroy(ev);
    ITTS_HIP_CHECK(e);
    ITTS_LAUNCH_CHECK();
  }
  const int64_t n_active = (int64_t)(uint32_t)(*h_active & 0xffffffff);
  ITTS_REQUIRE(n_active >= 0 && n_active <= t_total, "corrupt frame count");
  if (n_active > 0) {
    if (areg && sized)
      hipLaunchKernelGGL((d4c_kernel<true, 12>), dim3((unsigned)n_active), dim3(NT), lds, s, a);
    else if (areg)
      hipLaunchKernelGGL(d4c_kernel<true>, dim3((unsigned)n_active), dim3(NT), lds, s, a);
    else if (sized)
      hipLaunchKernelGGL((d4c_kernel<false, 11>), dim3((unsigned)n_active), dim3(NT), lds, s, a);
    else
      hipLaunchKernelGGL(d4c_kernel<false>, dim3((unsigned)n_active), dim3(NT), lds, s, a);
    ITTS_LAUNCH_CHECK();
  }
  ITTS_HIP_CHECK(itts::scratch_free(d_nwin, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_order, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_xo, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_fo, s));
  return ITTS_OK;
}
