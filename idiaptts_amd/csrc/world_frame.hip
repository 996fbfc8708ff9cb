// Frame-parallel WORLD / SPTK kernels: CheapTrick spectral envelope, SPTK mel-cepstral analysis
// (mcep), mel-cepstrum -> amplitude spectrum (mgc2sp), code / decode aperiodicity.
//
// Replaces (reference call sites, /root/reference/idiaptts):
//   pyworld.cheaptrick inside pyworld.wav2world   src/data_preparation/world/WorldFeatLabelGen.py:792
//   pysptk.mcep                                    src/data_preparation/audio/AudioProcessing.py:146-152
//   pysptk.mgc2sp + exp(real)                      AudioProcessing.py:252-256
//   pyworld.code_aperiodicity / decode_aperiodicity  WorldFeatLabelGen.py:805, :940-941
//
// One 256-thread workgroup per frame; everything between the waveform samples read and the
// feature row written stays in LDS.  Roofline: HBM-bound by contract --
// analysis reads 80 samples*8 B and writes K*8 B (+60*4 B mcep) per frame; in practice the
// kernels are fp64-FFT/LDS bound, so achieved fp64 FLOP/s is reported beside GB/s.
#include <algorithm>
#include <cmath>

#include "context.h"
#include "wave_fft.h"
#include "world_dev.h"

namespace itts {
using namespace wd;

// ------------------------------------------------------------------------------------------------
// CheapTrick for one frame. Result (power spectral envelope, fft/2+1 bins) is left in P.
// LDS: z [fft/2+1] complex, P [fft/2+1], mir [fft/2 + 2*bmax + 1], red [NT+8], tw [fft/2] cplx
struct CtLds {
  const double2* tw;
  double2* z;
  double* P;
  double* mir;
  double* red;
};

//
// rn = this frame's slice of the call's randn() stream (as the integers behind the normals, see
// context.h): WORLD adds 1e-12 * randn to every windowed sample and eps * |randn| to every bin of
// the smoothed spectrum, consuming one stream per CheapTrick() call frame after frame
// (cheaptrick.cpp GetWindowedWaveform / AddInfinitesimalNoise).  Irrelevant next to speech, but
// it is what keeps the envelope of digitally silent frames finite, so it is reproduced exactly.
__device__ __forceinline__ double randn_of(uint32_t r) { return r / 268435456.0 - 6.0; }

__device__ inline void cheaptrick_frame(const double* __restrict__ x, int64_t xl, int fs, double f0,
                                        double pos, int fft, int logfft, double q1, const CtLds& L,
                                        const uint32_t* __restrict__ rn) {
  const int h = fft / 2;
  const int half = mround(1.5 * fs / f0);
  const int n = 2 * half + 1;
  const int64_t c = mround(pos * fs + 0.001);
  double* zr = reinterpret_cast<double*>(L.z);  // real view, fft + 2 doubles
  double* win = L.mir;                          // n <= fft-3 doubles of scratch
  // window and its energy
  double e = 0.0;
  for (int i = threadIdx.x; i < n; i += NT) {
    const int b = i - half;
    const double w = 0.5 * cos_mid(kPi * ((double)b / 1.5 / fs) * f0) + 0.5;
    win[i] = w;
    e += w * w;
  }
  e = sqrt(bsum(e, L.red));
  double swf = 0.0, sw = 0.0;
  for (int i = threadIdx.x; i < fft + 2; i += NT) {
    double v = 0.0;
    if (i < n) {
      const double w = win[i] / e;
      win[i] = w;
      int64_t idx = c + i - half;
      idx = idx < 0 ? 0 : (idx > xl - 1 ? xl - 1 : idx);
      v = x[idx] * w;
      const double nz = randn_of(rn[i]) * 1e-12;
      v = v + nz;
      swf += v;
      sw += w;
    }
    zr[i] = v;
  }
  bsum2(swf, sw, L.red);
  const double mean = swf / sw;
  for (int i = threadIdx.x; i < n; i += NT) zr[i] -= win[i] * mean;
  __syncthreads();
  rfft_lds(L.z, fft, logfft, L.tw, fft);
  for (int k = threadIdx.x; k <= h; k += NT) {
    const double2 v = L.z[k];
    L.P[k] = v.x * v.x + v.y * v.y;
  }
  __syncthreads();
  dc_correction(L.P, f0, fs, fft);
  linear_smoothing(L.P, f0 * 2.0 / 3.0, fs, fft, L.P, L.mir, L.red);
  // smoothing with recovery (cepstral liftering)
  for (int k = threadIdx.x; k <= h; k += NT) {
    const double nz = fabs(randn_of(rn[n + k])) * 2.2204460492503131e-16;
    const double lp = log_pos(L.P[k] + nz);
    zr[k] = lp;
    if (k > 0 && k < h) zr[fft - k] = lp;
  }
  if (threadIdx.x == 0) {
    zr[fft] = 0.0;
    zr[fft + 1] = 0.0;
  }
  __syncthreads();
  rfft_lds(L.z, fft, logfft, L.tw, fft);
  for (int k = threadIdx.x; k <= h; k += NT) {
    double sl = 1.0, cl = 1.0;
    if (k > 0) {
      const double q = (double)k / fs;
      sl = sin_mid(kPi * f0 * q) / (kPi * f0 * q);
      cl = (1.0 - 2.0 * q1) + 2.0 * q1 * cos_mid(2.0 * kPi * q * f0);
    }
    L.z[k] = make_double2(L.z[k].x * sl * cl, 0.0);
  }
  __syncthreads();
  irfft_lds(L.z, fft, logfft, L.tw, fft);
  for (int k = threadIdx.x; k <= h; k += NT) L.P[k] = exp(zr[k]);
  __syncthreads();
}

struct FrameArgs {
  const double* x;          // concatenated waveforms
  const int64_t* x_off;     // [U+1] sample offsets (device)
  const double* f0;         // [Ttot]
  const int64_t* f_off;     // [U+1] frame offsets (device)
  int n_utts;
  int fs;
  double frame_period;      // ms
  int fft, logfft;
  double q1;
  double* sp;               // [Ttot, K] power envelope or nullptr
  // mcep (optional)
  int do_mcep;
  FreqtTables ft;
  int m;
  double alpha, eps;
  int itr1, itr2;
  double dd;
  float* mc_f32;            // [Ttot, ld_mc] or nullptr
  double* mc_f64;           // [Ttot, m+1] or nullptr
  int64_t ld_mc;
  int* iters;               // [Ttot] or nullptr
  const double2* g_tw;
  const double2* g_tw_compact;   // DeviceContext::tw_compact of this fft size
  int bmax;                 // bound on the smoothing boundary (LDS carve)
  const uint32_t* rn;       // safeguard-noise streams, utterance u at rn + f_off[u] * rn_pitch
  const int64_t* rn_pos;    // [Ttot] position of each frame inside its utterance's stream
  int64_t rn_pitch;         // bound on the normals one frame consumes
  int64_t t_total;          // frames of the call
  int far_only;             // cheaptrick_kernel: only the frames the wave kernel leaves (ct_far)
};

__device__ __forceinline__ double ct_frame_f0(double f0, int fs, int fft) {
  const double floor_f0 = 3.0 * fs / (fft - 3.0);
  return f0 > floor_f0 ? f0 : 500.0;  // WORLD kDefaultF0
}

// F0 values at or beyond the Nyquist frequency (an input error: WORLD's own DC correction reads past
// its arrays there) stay with the workgroup kernel: the wave kernel's scratch and its short sine /
// cosine reductions (arguments up to 2 pi * 512 * f0 / fs, valid to 1e5) are sized for F0 below it.
__device__ __forceinline__ bool ct_far(double f0, int fs) { return !(f0 < 0.5 * fs); }

// Stream positions: frame t of an utterance starts where frames 0..t-1 stopped; each consumes its
// window length (2 * round(1.5 fs / f0) + 1) plus fft/2+1 normals.  One workgroup per utterance,
// blocked exclusive scan with a running carry.  Also writes the stream's offset and length.
__global__ __launch_bounds__(256) void ct_noise_pos_kernel(const double* __restrict__ f0,
                                                           const int64_t* __restrict__ f_off, int fs,
                                                           int fft, int64_t pitch,
                                                           int64_t* __restrict__ pos,
                                                           int64_t* __restrict__ r_off,
                                                           int64_t* __restrict__ r_len) {
  __shared__ int64_t wsum[4];
  __shared__ int64_t carry_s;
  const int u = blockIdx.x;
  const int64_t g0 = f_off[u];
  const int64_t T = f_off[u + 1] - g0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < T; base += 256) {
    const int64_t t = base + threadIdx.x;
    int64_t cnt = 0;
    if (t < T) cnt = 2 * (int64_t)mround(1.5 * fs / ct_frame_f0(f0[g0 + t], fs, fft)) + 1 + fft / 2 + 1;
    int64_t inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int64_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int64_t before = carry_s;
    for (int w = 0; w < wv; ++w) before += wsum[w];
    if (t < T) pos[g0 + t] = before + inc - cnt;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = before + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    r_off[u] = g0 * pitch;
    r_len[u] = carry_s;
  }
}

__device__ inline void carve_ct(char*& p, int fft, int bmax, CtLds& L, bool tw_in_lds) {
  L.tw = reinterpret_cast<double2*>(p);
  if (tw_in_lds) p += (size_t)(fft / 2) * sizeof(double2);
  L.z = reinterpret_cast<double2*>(p); p += (size_t)(fft / 2 + 1) * sizeof(double2);
  L.P = reinterpret_cast<double*>(p); p += (size_t)(fft / 2 + 2) * sizeof(double);
  L.mir = reinterpret_cast<double*>(p); p += (size_t)(fft + 2 * bmax + 2) * sizeof(double);
  L.red = reinterpret_cast<double*>(p); p += (size_t)(NT + 8) * sizeof(double);
}
static size_t ct_lds_bytes(int fft, int bmax, bool tw_in_lds) {
  return (tw_in_lds ? (size_t)(fft / 2) * 16 : 0) + (size_t)(fft / 2 + 1) * 16 + (size_t)(fft / 2 + 2) * 8 +
         (size_t)(fft + 2 * bmax + 2) * 8 + (size_t)(NT + 8) * 8;
}
// CheapTrick -- one workgroup per frame; the twiddles come through the cache from the compact table
// (8 KB of LDS less: six workgroups per CU instead of five).
__global__ __launch_bounds__(NT) void cheaptrick_kernel(FrameArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* p = smem;
  CtLds L;
  carve_ct(p, a.fft, a.bmax, L, false);
  L.tw = a.g_tw_compact;
  if (a.far_only) {
    // normally no frame is left: the workgroup's F0 values are looked at side by side, once
    bool any = false;
    for (int64_t g = blockIdx.x + (int64_t)threadIdx.x * gridDim.x; g < a.t_total; g += (int64_t)NT * gridDim.x)
      any = any || ct_far(ct_frame_f0(a.f0[g], a.fs, a.fft), a.fs);
    if (!__syncthreads_or(any)) return;
  }
  for (int64_t g = blockIdx.x; g < a.t_total; g += gridDim.x) {
    const double f0 = ct_frame_f0(a.f0[g], a.fs, a.fft);
    if (a.far_only && !ct_far(f0, a.fs)) continue;       // workgroup-uniform
    const int u = find_utt_wave(a.f_off, a.n_utts, g);
    const int64_t t = g - a.f_off[u];
    const double* x = a.x + a.x_off[u];
    const int64_t xl = a.x_off[u + 1] - a.x_off[u];
    const double pos = (double)t * a.frame_period / 1000.0;
    cheaptrick_frame(x, xl, a.fs, f0, pos, a.fft, a.logfft, a.q1, L,
                     a.rn + a.f_off[u] * a.rn_pitch + a.rn_pos[g]);
    const int K = a.fft / 2 + 1;
    for (int k = threadIdx.x; k < K; k += NT) a.sp[g * K + k] = L.P[k];
    __syncthreads();
  }
}

// CheapTrick with one WAVE per frame (fft_size 1024: 16 / 22.05 kHz; wave_fft.h).  The three real
// transforms run in registers (eight complex points per lane); everything between them that needs
// random access -- the DC correction's interpolation, the mirrored running sum of the smoothing --
// lives in the wave's OWN 8.5 KB of exchange rows, which are idle between transforms: sixteen frames
// in flight per CU, no workgroup barrier.  Same expressions per sample and per bin as
// cheaptrick_frame; the sums (window energy, mean, running sum) associate differently -- the
// envelope agrees with it to rounding (tests: 1e-8 relative against the oracle, as before).
// The cosines and sines take fastmath.h's short forms directly: their arguments are bounded by
// 2 pi * 512 * f0 / fs here, far inside the reduction's range, and the out-of-line fallback of
// cos_mid / sin_mid -- a call the kernel never takes -- would cost the register allocation of the whole kernel.
#ifndef CTW_THREADS_N
#define CTW_THREADS_N 512     // eight frames in flight per CU, 218 registers per lane (twelve at 168 spill 49 of them: 5.1 against 4.4 ms)
#endif
constexpr int CTW_THREADS = CTW_THREADS_N;
// running-sum elements per lane: the smoothing's scratch is the wave's exchange rows (64 * 17 = 1088 doubles at
// 1024 points, 64 * 33 = 2112 at 2048)
template <int R> constexpr int ctw_scan() { return wf::lds_bytes<R>() / 8 / 64; }

__device__ __forceinline__ double ct_bcast0(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// TH: threads per workgroup = 64 x frames in flight per CU.  2048-point transforms (R = 16: 44.1 / 48 kHz) hold twice
// the bins per lane: with eight waves per CU (256 registers) 146 of them live in scratch; four waves per CU (one per
// SIMD, 256 + 98 registers, no scratch) measured SLOWER (round 5: 3.48 against 3.05 ms for 64 utterances): the second
// wave of a SIMD covers more than the scratch traffic costs.
#ifndef CTW_DEAL
#define CTW_DEAL 2      // frames a wave takes from the counter at a time (0: dealt by stride, as until late in round 5: 4.42 -> 4.00 ms at 16 kHz)
#endif
template <int R, int TH = CTW_THREADS>
__global__ __launch_bounds__(TH) void cheaptrick_wave_kernel(FrameArgs a, int64_t t_total, int* __restrict__ next) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FFT = 128 * R, H = 64 * R, K = H + 1, NW = TH / 64, CTW_SCAN = ctw_scan<R>();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), l0 = wf::lane_id();
  typename wf::PlanOf<R>::type P;
  wf::table_init<R>(smem, a.g_tw_compact);
  char* rows = smem + wf::table_bytes<R>() + (size_t)wv * wf::lds_bytes<R>();
  wf::plan_init(P, a.g_tw_compact, rows, smem);
  double* S = reinterpret_cast<double*>(rows);       // the exchange rows as 1088 doubles of scratch
#if CTW_DEAL
  // frames from a counter, CTW_DEAL neighbours at a time (as the pulses of syn_pulse_wave_kernel)
  int64_t g = 0, g_end = 0;
  for (;; ++g) {
    if (g >= g_end) {
      int base = 0;
      if (l0 == 0) base = atomicAdd(next, CTW_DEAL);
      g = __builtin_amdgcn_readfirstlane(base);
      if (g >= t_total) break;
      g_end = min(g + (int64_t)CTW_DEAL, t_total);
    }
#else
  (void)next;
  for (int64_t g = (int64_t)blockIdx.x * NW + wv; g < t_total; g += (int64_t)gridDim.x * NW) {
#endif
    // opaque per frame: what derives from the lane number, the sampling rate and q1 is a handful of
    // integer and fp64 operations -- hoisted out of the loop it is sixty registers held for the
    // whole kernel
    int l = l0, fs = a.fs;
    double q1 = a.q1;
    asm volatile("" : "+v"(l), "+s"(fs), "+s"(q1));
    const int u = __builtin_amdgcn_readfirstlane(find_utt_wave(a.f_off, a.n_utts, g));
    const int64_t fo = a.f_off[u];
    const double* x = a.x + a.x_off[u];
    const int64_t xl = a.x_off[u + 1] - a.x_off[u];
    const double f0 = ct_frame_f0(a.f0[g], fs, FFT);
    if (ct_far(f0, fs)) continue;                   // wave-uniform; cheaptrick_kernel(far_only) takes the frame
    const double pos = (double)(g - fo) * a.frame_period / 1000.0;
    const uint32_t* rn = a.rn + fo * a.rn_pitch + a.rn_pos[g];
    const int half = __builtin_amdgcn_readfirstlane(mround(1.5 * fs / f0));
    const int n = 2 * half + 1;
    const int64_t c = mround(pos * fs + 0.001);
    // ---- windowed waveform, two samples per complex point: sample i = 2 (l + 64 q) + s
    double2 z[R], xh;
    {
      // the window of sample i waits in S[l + 64 j] between the three passes (energy, mean, removal):
      // sixteen doubles per lane that the registers do not have beside the samples
      double e = 0.0;
#pragma unroll
      for (int j = 0; j < 2 * R; ++j) {
        const int i = 2 * (l + 64 * (j >> 1)) + (j & 1);
        if (128 * (j >> 1) < n) {                   // wave-uniform: registers beyond the window are skipped
          double wj = 0.0;
          if (i < n) {
            const int b = i - half;
            wj = 0.5 * fm::fcos(kPi * ((double)b / 1.5 / fs) * f0) + 0.5;
            e += wj * wj;
          }
          S[l + 64 * j] = wj;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      e = sqrt(wave_sum_at(e, l));
      // the windowed samples are formed twice -- for the mean, then for the transform -- from the same
      // loads and the same expressions (same bits): kept in registers between the passes they were the
      // kernel's largest array
      auto sample = [&](int j, double wn) -> double {
        const int i = 2 * (l + 64 * (j >> 1)) + (j & 1);
        int64_t idx = c + i - half;
        idx = idx < 0 ? 0 : (idx > xl - 1 ? xl - 1 : idx);
        double vj = x[idx] * wn;
        const double nz = randn_of(rn[i]) * 1e-12;
        vj = vj + nz;
        return vj;
      };
      double swf = 0.0, sw = 0.0;
#pragma unroll
      for (int j = 0; j < 2 * R; ++j) {
        const int i = 2 * (l + 64 * (j >> 1)) + (j & 1);
        if (128 * (j >> 1) < n) {
          if (i < n) {
            const double wn = S[l + 64 * j] / e;
            S[l + 64 * j] = wn;
            swf += sample(j, wn);
            sw += wn;
          }
        }
      }
      swf = wave_sum_at(swf, l);
      sw = wave_sum_at(sw, l);
      const double mean = swf / sw;
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const int i = 2 * (l + 64 * q);
        double a0 = 0.0, a1 = 0.0;
        if (128 * q < n) {
          if (i < n) {
            const double wn = S[l + 64 * (2 * q)];
            a0 = sample(2 * q, wn) - wn * mean;
          }
          if (i + 1 < n) {
            const double wn = S[l + 64 * (2 * q + 1)];
            a1 = sample(2 * q + 1, wn) - wn * mean;
          }
        }
        z[q] = make_double2(a0, a1);
      }
      wf::wave_sync();
    }
    wf::rfft<R>(z, xh, P);
    double pw[R], p512;
#pragma unroll
    for (int q = 0; q < R; ++q) pw[q] = z[q].x * z[q].x + z[q].y * z[q].y;
    p512 = ct_bcast0(xh.x * xh.x + xh.y * xh.y);
    // ---- DC correction (wd::dc_correction): the replica is interpolated from the uncorrected bins
#pragma unroll
    for (int q = 0; q < R; ++q) S[l + 64 * q] = pw[q];
    if (l == 0) S[H] = p512;
    wf::wave_sync();
    {
      const int upper = 2 + (int)(f0 * FFT / fs);
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const int ii = l + 64 * q;
        if (64 * q < upper - 1) {          // wave-uniform: one register of bins at speech F0
          if (ii < upper - 1) pw[q] += interp1q(f0, -(double)fs / FFT, S, upper + 1, (double)ii * fs / FFT);
        }
      }
      if (H < upper - 1) p512 += interp1q(f0, -(double)fs / FFT, S, upper + 1, (double)H * fs / FFT);
    }
    wf::wave_sync();
    // ---- linear smoothing over f0 * 2 / 3 (wd::linear_smoothing): mirrored copy, running sum, two
    // interpolated reads per bin
    const double width = f0 * 2.0 / 3.0;
    const int boundary = __builtin_amdgcn_readfirstlane((int)(width * FFT / fs) + 1);
    const int ml = H + boundary * 2 + 1;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int k = l + 64 * q;
      const double val = pw[q] * fs / FFT;
      S[boundary + k] = val;
      if (k >= 1 && k <= boundary) S[boundary - k] = val;
      if (k >= H - boundary) S[H + boundary + (H - k)] = val;
    }
    if (l == 0) S[H + boundary] = p512 * fs / FFT;
    wf::wave_sync();
    {
      const int chunk = (ml + 63) >> 6;            // <= CTW_SCAN
      const int lo = l * chunk;
      double cs[CTW_SCAN];
      double run = 0.0;
#pragma unroll
      for (int j = 0; j < CTW_SCAN; ++j) {
        double t = 0.0;
        if (j < chunk && lo + j < ml) t = S[lo + j];
        run += t;
        cs[j] = run;
      }
      double incl = run;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double o = lane_shfl(incl, l - off);       // lanes below `off` fetch from a wrapped lane and drop it
        if (l >= off) incl += o;
      }
      const double excl = incl - run;
      wf::wave_sync();
#pragma unroll
      for (int j = 0; j < CTW_SCAN; ++j)
        if (j < chunk && lo + j < ml) S[lo + j] = excl + cs[j];
      wf::wave_sync();
    }
    {
      const double org = -((double)boundary - 0.5) * fs / FFT;
      const double dfi = (double)fs / FFT;
#pragma unroll
      for (int q = 0; q <= R; ++q) {
        const int k = q < R ? l + 64 * q : H;
        const double fa = (double)k / FFT * fs - width / 2.0;
        const double lo = interp1q(org, dfi, S, ml, fa);
        const double hi = interp1q(org, dfi, S, ml, fa + width);
        const double o = (hi - lo) / width;
        if (q < R) pw[q < R ? q : 0] = o; else p512 = o;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    wf::wave_sync();
    // ---- smoothing with recovery (cepstral liftering)
    {
      double lp[R], lp512;
#pragma unroll
      for (int q = 0; q <= R; ++q) {
        const int k = q < R ? l + 64 * q : H;
        const double nz = fabs(randn_of(rn[n + k])) * 2.2204460492503131e-16;
        const double v = log_pos((q < R ? pw[q < R ? q : 0] : p512) + nz);
        if (q < R) lp[q < R ? q : 0] = v; else lp512 = v;
        __builtin_amdgcn_sched_barrier(0);
      }
      wf::pack_real_r<R>(lp, lp512, z, P, true);
    }
    wf::rfft<R>(z, xh, P);
#pragma unroll
    for (int q = 0; q <= R; ++q) {
      const int k = q < R ? l + 64 * q : H;
      double sl = 1.0, cl = 1.0;
      if (k > 0) {
        const double qq = (double)k / fs;
        sl = fm::fsin(kPi * f0 * qq) / (kPi * f0 * qq);
        cl = (1.0 - 2.0 * q1) + 2.0 * q1 * fm::fcos(2.0 * kPi * qq * f0);
      }
      if (q < R) z[q < R ? q : 0] = make_double2(z[q < R ? q : 0].x * sl * cl, 0.0);
      else xh = make_double2(xh.x * sl * cl, 0.0);     // X[512]: lane 0's value is the one irfft reads
      __builtin_amdgcn_sched_barrier(0);
    }
    wf::irfft<R>(z, xh, P);
    double* out = a.sp + g * K;
#pragma unroll
    for (int q = 0; q < R / 2; ++q) {
      const int k = 2 * (l + 64 * q);
      out[k] = fm::fexp(z[q].x);
      out[k + 1] = fm::fexp(z[q].y);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (l == 0) out[H] = fm::fexp(z[R / 2].x);
  }
}

// mgc2sp(gamma = 0): c = freqt(mc, -alpha) to order fftlen/2, FFT, real part; optional exp.
struct Mgc2spArgs {
  const double* cep;  // [T, ld_cep] de-warped cepstra, f2 + 1 per frame
  int64_t ld_cep;
  int64_t T;
  int m, fftlen, logfft;
  float* out_f32;     // exp(float(real))  (AudioProcessing.mcep_to_amp_sp :252-256)
  double* out_f64;    // raw log amplitude
  double* out_pow;    // double(exp(float(real)))^2: what world_features_to_raw feeds WORLD (:925)
  const double2* g_tw;
};

__global__ __launch_bounds__(NT) void mgc2sp_kernel(Mgc2spArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double2* tw = reinterpret_cast<double2*>(smem);
  double2* z = tw + a.fftlen / 2;
  double* zr = reinterpret_cast<double*>(z);
  const int64_t g = blockIdx.x;
  const int f2 = a.fftlen / 2;
  load_twiddles(tw, a.g_tw, a.fftlen);
  // the de-warped cepstrum c = freqt(mc, m -> f2, -alpha) of all frames comes from one fp64-MFMA
  // GEMM (per-frame matrix-vector products streamed the 250 KB warping matrix through every
  // workgroup: L2-bandwidth bound)
  const double* c = a.cep + g * a.ld_cep;
  for (int i = tid(); i < a.fftlen + 2; i += NT) zr[i] = i <= f2 ? c[i] : 0.0;
  __syncthreads();
  rfft_lds(z, a.fftlen, a.logfft, tw, a.fftlen);
  for (int k = tid(); k <= f2; k += NT) {
    const double re = z[k].x;
    if (a.out_f64) a.out_f64[g * (f2 + 1) + k] = re;
    const float amp = expf((float)re);
    if (a.out_f32) a.out_f32[g * (f2 + 1) + k] = amp;
    if (a.out_pow) a.out_pow[g * (f2 + 1) + k] = (double)amp * (double)amp;
  }
}


// ---- aperiodicity coding -----------------------------------------------------------------------
// WORLD interp1 (with histc index semantics) on a short monotone knot vector, one query.
__device__ __forceinline__ double interp1_small(const double* xk, const double* yk, int n, double xi) {
  // histc: index k (1-based) with xk[k-1] <= xi < xk[k], clamped to [1, n-1]
  int k = 1;
  while (k < n - 1 && xi >= xk[k]) ++k;
  const double h = xk[k] - xk[k - 1];
  const double s = (xi - xk[k - 1]) / h;
  return yk[k - 1] + s * (yk[k] - yk[k - 1]);
}

__global__ void code_aperiodicity_kernel(const double* __restrict__ ap, int64_t T, int fft_size, int fs,
                                         int nap, double* __restrict__ bap_f64, float* __restrict__ bap_f32) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= T * nap) return;
  const int64_t t = i / nap;
  const int b = (int)(i - t * nap);
  const int K = fft_size / 2 + 1;
  const double cf = 3000.0 * (b + 1);
  // frequency axis fa[k] = k*fs/fft_size; bracket cf (histc semantics, clamped)
  int k = (int)(cf * fft_size / fs);
  while (k + 1 < K && (double)(k + 1) * fs / fft_size <= cf) ++k;
  while (k > 0 && (double)k * fs / fft_size > cf) --k;
  if (k > K - 2) k = K - 2;
  const double x0 = (double)k * fs / fft_size, x1 = (double)(k + 1) * fs / fft_size;
  const double y0 = 20.0 * log10(ap[t * K + k]), y1 = 20.0 * log10(ap[t * K + k + 1]);
  const double s = (cf - x0) / (x1 - x0);
  const double v = y0 + s * (y1 - y0);
  if (bap_f64) bap_f64[i] = v;
  if (bap_f32) bap_f32[i] = (float)v;
}

// A workgroup owns DA_FRAMES consecutive frames (contiguous in the output); element e of its range is
// bin k of its frame tl with e = tl (2^lg + 1) + k -- a shift and a correction instead of the 64-bit
// division a flat element index needs (the pass was VALU-bound on it: 0.54 ms for 1.3 GB out).
// (32 frames a workgroup: with 8 the launch of 39 000 workgroups -- 11 ns each -- was what the pass took, 0.43 ms whether
// it wrote every row or a third of them)
constexpr int DA_FRAMES = 32;
// f0 (or null): only the rows a VOICED pulse of the synthesis can read are decoded -- a pulse is voiced when the V/UV
// flags of the two frames around it interpolate above one half, so one of them has f0 > 0, and it reads exactly those
// two rows: row t is needed iff one of f0[t - 1], f0[t], f0[t + 1] is positive (a superset at the seams between
// utterances).  The other rows are left as they are.  (The bench's speech-like signals: 30 % of the frames voiced.)
__global__ __launch_bounds__(256) void decode_aperiodicity_kernel(const double* __restrict__ bap, int64_t T, int fs,
                                                                  int fft_size, int lg, int nap,
                                                                  double* __restrict__ ap, const double* __restrict__ f0) {
  const int K = fft_size / 2 + 1;
  const int64_t t0 = (int64_t)blockIdx.x * DA_FRAMES;
  const int nfr = (int)(T - t0 < DA_FRAMES ? T - t0 : DA_FRAMES);
  unsigned need = 0xffffffffu;    // bit tl: frame t0 + tl has to be decoded
  if (f0) {
    // the flags f0[t0 - 1 .. t0 + DA_FRAMES] > 0 by as many lanes at once (a thread walking them was dozens of dependent
    // trips to memory in front of everything a workgroup does)
    __shared__ unsigned s_need;
    if (threadIdx.x < 64) {
      const int64_t t = t0 - 1 + (int)threadIdx.x;
      const bool v = (int)threadIdx.x < nfr + 2 && t >= 0 && t < T && f0[t] > 0.0;
      const unsigned long long vb = __ballot(v);           // bit i: frame t0 - 1 + i is voiced
      if (threadIdx.x == 0) s_need = (unsigned)((vb | (vb >> 1) | (vb >> 2)) & ((1ull << nfr) - 1ull));
    }
    __syncthreads();
    need = s_need;
    if (need == 0) return;
  }
  for (int e = threadIdx.x; e < nfr * K; e += 256) {
  int tl = e >> lg;
  int k = (e & ((1 << lg) - 1)) - tl;
  if (k < 0) { --tl; k += K; }
  if (!((need >> tl) & 1u)) continue;
  const int64_t t = t0 + tl;
  const int64_t i = t * K + k;
  double cfa[8], cap[8];
  double mean = 0.0;
  for (int b = 0; b < nap; ++b) {
    cfa[b] = b * 3000.0;
    cap[b + 1] = bap[t * nap + b];
    mean += cap[b + 1];
  }
  mean /= nap;
  cfa[nap] = nap * 3000.0;
  cfa[nap + 1] = fs / 2.0;
  cap[0] = -60.0;
  cap[nap + 1] = -kEps;
  double v = 1.0 - kEps;
  if (!(mean > -0.5)) {  // WORLD codec.cpp CheckVUV: mean band aperiodicity > -0.5 dB => unvoiced
    const double f = (double)fs / fft_size * k;
    // 10^(y / 20), y in [-60, 0] dB, as exp(y ln 10 / 20) with fastmath.h's exp: the library's pow is
    // ~150 instructions per bin, which made this pass (one value per bin, 1.3 GB out) VALU-bound at
    // 0.69 ms; relative difference to pow below 2e-15
    v = fm::fexp(interp1_small(cfa, cap, nap + 2, f) * (2.30258509299404568402 / 20.0));
  }
  ap[i] = v;
  }
}

// mcep_lockstep.hip
int mcep_lockstep(DeviceContext* ctx, const double* d_in, int in_is_power, int64_t T, int K, int order,
                  double alpha, double eps, int miniter, int maxiter, double threshold, float* d_mc_f32,
                  int64_t ld_mc, double* d_mc_f64, int* d_iters, hipStream_t s);

static int smoothing_bmax(int fs, int fft, double max_width) { return (int)(max_width * fft / fs) + 2; }

}  // namespace itts

using namespace itts;

static int ilog2_host(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}
static bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

extern "C" int itts_cheaptrick_mcep(const double* d_x, const int64_t* h_x_off, const double* d_f0,
                                    const int64_t* h_f_off, int n_utts, int fs, double frame_period_ms,
                                    int fft_size, double q1, double* d_sp, int order, double alpha,
                                    double eps, int miniter, int maxiter, double threshold,
                                    float* d_mc_f32, int64_t ld_mc, double* d_mc_f64, int* d_iters,
                                    void* stream) {
  ITTS_REQUIRE(h_x_off && h_f_off && (n_utts == 0 || (d_x && d_f0)), "null pointer");
  ITTS_REQUIRE(n_utts >= 0 && fs > 0 && frame_period_ms > 0, "bad sizes");
  ITTS_REQUIRE(is_pow2(fft_size) && fft_size >= 512 && fft_size <= 8192, "fft_size must be 2^k in [512, 8192]");
  const bool do_mcep = (d_mc_f32 != nullptr) || (d_mc_f64 != nullptr);
  ITTS_REQUIRE(n_utts == 0 || d_sp || do_mcep, "nothing to compute");
  if (do_mcep) {
    ITTS_REQUIRE(order >= 1 && order < fft_size / 2 && order <= 127, "bad mcep order");
    ITTS_REQUIRE(fft_size <= 2048, "fused mcep supports fft_size <= 2048");
    ITTS_REQUIRE(!d_mc_f32 || ld_mc >= order + 1, "ld_mc too small");
  }
  if (n_utts == 0) return ITTS_OK;
  const int64_t t_total = h_f_off[n_utts];
  if (t_total == 0) return ITTS_OK;
  for (int u = 0; u < n_utts; ++u) {
    const int64_t n = h_x_off[u + 1] - h_x_off[u];
    ITTS_REQUIRE(n > 0, "empty utterance");
    ITTS_REQUIRE(h_f_off[u + 1] - h_f_off[u] == itts_world_num_frames(n, fs, frame_period_ms),
                 "frame offsets do not match int(1000*n/fs/frame_period)+1");
  }
  hipStream_t s = as_stream(stream);
  itts::ScratchScope scratch_scope(s);
  DeviceContext* ctx = get_context();
  if (!ctx) return ITTS_E_HIP;
  FrameArgs a{};
  // lockstep mcep needs the envelope in memory: use the caller's buffer or a temporary
  double* sp_buf = d_sp;
  if (do_mcep && !sp_buf)
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&sp_buf, (size_t)t_total * (fft_size / 2 + 1) * 8, s));
  int64_t *d_xo = nullptr, *d_fo = nullptr;
  int rc = upload_i64(h_x_off, n_utts + 1, &d_xo, s);
  if (rc) return rc;
  rc = upload_i64(h_f_off, n_utts + 1, &d_fo, s);
  if (rc) return rc;
  a.x = d_x; a.x_off = d_xo; a.f0 = d_f0; a.f_off = d_fo; a.n_utts = n_utts; a.fs = fs;
  a.frame_period = frame_period_ms; a.fft = fft_size; a.logfft = ilog2_host(fft_size); a.q1 = q1;
  a.sp = sp_buf; a.do_mcep = 0; a.m = order; a.alpha = alpha; a.eps = eps;
  a.itr1 = miniter; a.itr2 = maxiter; a.dd = threshold; a.mc_f32 = d_mc_f32; a.mc_f64 = d_mc_f64;
  a.ld_mc = ld_mc; a.iters = d_iters; a.g_tw = ctx->twiddles;
  a.g_tw_compact = ctx->tw_compact[a.logfft];
  a.bmax = smoothing_bmax(fs, fft_size, 1000.0);
  // safeguard-noise streams: positions (scan over the frames' window lengths), then the normals
  const int K = fft_size / 2 + 1;
  const int64_t rn_pitch = 2 * (int64_t)std::lround((fft_size - 3.0) / 2.0) + 1 + K;
  int64_t t_max = 0;
  for (int u = 0; u < n_utts; ++u) t_max = std::max<int64_t>(t_max, h_f_off[u + 1] - h_f_off[u]);
  int64_t* d_rpos = nullptr;   // [t_total] + r_off [U] + r_len [U]
  uint32_t* d_rn = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_rpos, (size_t)(t_total + 2 * n_utts + 1) * sizeof(int64_t), s));
  int* d_next = reinterpret_cast<int*>(d_rpos + t_total + 2 * n_utts);      // the wave kernel's frame counter
  ITTS_HIP_CHECK(hipMemsetAsync(d_next, 0, sizeof(int64_t), s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_rn, (size_t)t_total * rn_pitch * sizeof(uint32_t), s));
  int64_t* d_roff = d_rpos + t_total;
  int64_t* d_rlen = d_roff + n_utts;
  hipLaunchKernelGGL(ct_noise_pos_kernel, dim3(n_utts), dim3(256), 0, s, d_f0, d_fo, fs, fft_size,
                     rn_pitch, d_rpos, d_roff, d_rlen);
  ITTS_LAUNCH_CHECK();
  rc = launch_randn_u32(ctx, d_roff, d_rlen, n_utts, t_max * rn_pitch, d_rn, s);
  if (rc) return rc;
  a.rn = d_rn; a.rn_pos = d_rpos; a.rn_pitch = rn_pitch; a.t_total = t_total; a.far_only = 0;
  // the smoothing boundary of the wave kernel is bounded by its scratch: 513 + 2 b <= 1088 doubles
  const int wave_r = fft_size == 1024 ? 8 : (fft_size == 2048 ? 16 : 0);
  if (wave_r != 0 && fft_size / 2 + 1 + 2 * ((int)(1000.0 * 2.0 / 3.0 * fft_size / fs) + 1) <=
                         64 * (wave_r == 8 ? ctw_scan<8>() : ctw_scan<16>())) {
    int dev = 0, n_cu = 256;
    ITTS_HIP_CHECK(hipGetDevice(&dev));
    ITTS_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    constexpr int TH16 = CTW_THREADS;          // (R = 16 at one wave per SIMD -- 256 threads, no scratch -- took 3.48 ms against 3.05)
    const int NW = (wave_r == 8 ? CTW_THREADS : TH16) / 64;
    const dim3 grid((unsigned)std::min<int64_t>((t_total + NW - 1) / NW, n_cu));
    if (wave_r == 8) {
      const size_t lds = wf::table_bytes<8>() + (size_t)NW * wf::lds_bytes<8>();
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)cheaptrick_wave_kernel<8>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(cheaptrick_wave_kernel<8>, grid, dim3(CTW_THREADS), lds, s, a, t_total, d_next);
    } else {
      const size_t lds = wf::table_bytes<16>() + (size_t)NW * wf::lds_bytes<16>();
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)cheaptrick_wave_kernel<16, TH16>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((cheaptrick_wave_kernel<16, TH16>), grid, dim3(TH16), lds, s, a, t_total, d_next);
    }
    ITTS_LAUNCH_CHECK();
    a.far_only = 1;     // frames with an F0 at or beyond fs / 2: a pass that reads the F0 values and normally finds none
  }
  {
    size_t lds = ct_lds_bytes(fft_size, a.bmax, false);
    ITTS_REQUIRE(lds <= 160 * 1024, "LDS budget exceeded");
    ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)cheaptrick_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(cheaptrick_kernel, dim3((unsigned)(a.far_only ? std::min<int64_t>(t_total, 256) : t_total)),
                       dim3(NT), lds, s, a);
  }
  ITTS_LAUNCH_CHECK();
  ITTS_HIP_CHECK(itts::scratch_free(d_xo, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_fo, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_rpos, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_rn, s));
  if (do_mcep) {
    rc = mcep_lockstep(ctx, sp_buf, 1, t_total, fft_size / 2 + 1, order, alpha, eps, miniter, maxiter,
                       threshold, d_mc_f32, ld_mc, d_mc_f64, d_iters, s);
    if (sp_buf != d_sp) ITTS_HIP_CHECK(itts::scratch_free(sp_buf, s));
    if (rc) return rc;
  }
  return ITTS_OK;
}

extern "C" int itts_mcep(const double* d_amp_sp, int64_t T, int K, int order, double alpha, double eps,
                         int miniter, int maxiter, double threshold, float* d_mc_f32, int64_t ld_mc,
                         double* d_mc_f64, int* d_iters, void* stream) {
  ITTS_REQUIRE(d_amp_sp && (d_mc_f32 || d_mc_f64), "null pointer");
  const int flng = (K - 1) * 2;
  ITTS_REQUIRE(T >= 0 && is_pow2(flng) && flng >= 64 && flng <= 2048, "K must be 2^k/2+1, K <= 1025");
  ITTS_REQUIRE(order >= 1 && order < flng / 2 && order <= 127, "bad mcep order");
  ITTS_REQUIRE(!d_mc_f32 || ld_mc >= order + 1, "ld_mc too small");
  if (T == 0) return ITTS_OK;
  DeviceContext* ctx = get_context();
  if (!ctx) return ITTS_E_HIP;
  return mcep_lockstep(ctx, d_amp_sp, 0, T, K, order, alpha, eps, miniter, maxiter, threshold,
                       d_mc_f32, ld_mc, d_mc_f64, d_iters, as_stream(stream));
}

extern "C" int itts_mgc2sp(const double* d_mc, int64_t T, int order, double alpha, int fftlen,
                           float* d_amp_f32, double* d_logamp_f64, double* d_pow_f64, void* stream) {
  ITTS_REQUIRE(d_mc && (d_amp_f32 || d_logamp_f64 || d_pow_f64), "null pointer");
  ITTS_REQUIRE(T >= 0 && is_pow2(fftlen) && fftlen >= 64 && fftlen <= 8192, "bad fftlen");
  ITTS_REQUIRE(order >= 0 && order <= fftlen / 2 && order <= 1023, "bad order");
  if (T == 0) return ITTS_OK;
  DeviceContext* ctx = get_context();
  if (!ctx) return ITTS_E_HIP;
  hipStream_t s = as_stream(stream);
  itts::ScratchScope scratch_scope(s);
  const int K = fftlen / 2 + 1;
  // Orders up to 63 at transform sizes up to 2048: de-warping, transform and real part are ONE linear map
  // (FreqtTables::specT), so the log amplitude is one fp64-MFMA product and exp / square ride in its
  // epilogue -- no cepstrum array, no transform kernel (round 4; until then a GEMM to [T x K] cepstra
  // and one transform per frame).
  if (order + 1 <= 64 && fftlen <= 2048) {
    const FreqtTables* fts = get_freqt(ctx, order, fftlen / 2, alpha, false, false, true);
    if (!fts) return ITTS_E_HIP;
    return launch_gemm_f64_mgc2sp(d_mc, order + 1, fts->specT, K, T, K, order + 1, d_amp_f32, d_logamp_f64,
                                  d_pow_f64, s);
  }
  const FreqtTables* ft = get_freqt(ctx, order, fftlen / 2, alpha, false);
  if (!ft) return ITTS_E_HIP;
  const int64_t ld_cep = (K + 1) & ~1;
  double* d_cep = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_cep, (size_t)T * ld_cep * 8, s));
  int rc = launch_gemm_f64(d_mc, order + 1, ft->invT, K, d_cep, ld_cep, T, K, order + 1, nullptr, s,
                           /*a_has_slack=*/false);
  if (rc) return rc;
  Mgc2spArgs a{d_cep, ld_cep, T, order, fftlen, ilog2_host(fftlen), d_amp_f32, d_logamp_f64, d_pow_f64,
               ctx->twiddles};
  {
    size_t lds = (size_t)(fftlen / 2) * 16 + (size_t)(fftlen / 2 + 1) * 16;
    ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)mgc2sp_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mgc2sp_kernel, dim3((unsigned)T), dim3(NT), lds, s, a);
  }
  ITTS_LAUNCH_CHECK();
  ITTS_HIP_CHECK(itts::scratch_free(d_cep, s));
  return ITTS_OK;
}

extern "C" int itts_code_aperiodicity(const double* d_ap, int64_t T, int fft_size, int fs,
                                      double* d_bap_f64, float* d_bap_f32, void* stream) {
  ITTS_REQUIRE(d_ap && (d_bap_f64 || d_bap_f32), "null pointer");
  const int nap = itts_num_aperiodicities(fs);
  ITTS_REQUIRE(T >= 0 && nap >= 1 && nap <= 5 && is_pow2(fft_size), "bad sizes");
  if (T == 0) return ITTS_OK;
  const int64_t n = T * nap;
  hipLaunchKernelGGL(code_aperiodicity_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     as_stream(stream), d_ap, T, fft_size, fs, nap, d_bap_f64, d_bap_f32);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

static int decode_aperiodicity_impl(const double* d_bap, const double* d_f0, int64_t T, int fs, int fft_size,
                                    double* d_ap, void* stream) {
  ITTS_REQUIRE(d_bap && d_ap, "null pointer");
  const int nap = itts_num_aperiodicities(fs);
  ITTS_REQUIRE(T >= 0 && nap >= 1 && nap <= 5 && is_pow2(fft_size), "bad sizes");
  if (T == 0) return ITTS_OK;
  ITTS_REQUIRE(fft_size >= 32 && (T + DA_FRAMES - 1) / DA_FRAMES < ((int64_t)1 << 31), "bad sizes");
  hipLaunchKernelGGL(decode_aperiodicity_kernel, dim3((unsigned)((T + DA_FRAMES - 1) / DA_FRAMES)), dim3(256), 0,
                     as_stream(stream), d_bap, T, fs, fft_size, ilog2_host(fft_size / 2), nap, d_ap, d_f0);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_decode_aperiodicity(const double* d_bap, int64_t T, int fs, int fft_size,
                                        double* d_ap, void* stream) {
  return decode_aperiodicity_impl(d_bap, nullptr, T, fs, fft_size, d_ap, stream);
}

extern "C" int itts_decode_aperiodicity_voiced(const double* d_bap, const double* d_f0, int64_t T, int fs,
                                               int fft_size, double* d_ap, void* stream) {
  ITTS_REQUIRE(d_f0, "null pointer");
  return decode_aperiodicity_impl(d_bap, d_f0, T, fs, fft_size, d_ap, stream);
}
