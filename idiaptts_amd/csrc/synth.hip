// WORLD waveform synthesis (synthesis.cpp) for batches of utterances + de-pre-emphasis.
// Replaces pyworld.synthesize and the lfilter de-pre-emphasis in
// WorldFeatLabelGen.world_features_to_raw (src/data_preparation/world/WorldFeatLabelGen.py:943-945,
// src/data_preparation/audio/AudioProcessing.py:329-331).
//
// WORLD walks glottal pulses sequentially and draws noise from ONE xorshift128 stream.  Here:
//   1. per-sample F0 / VUV interpolation is parallel; the phase sum keeps WORLD's sequential
//      rounding (see syn_phase_seq_kernel);
//   2. pulse positions are compacted in order (ballot + block offsets);
//   3. the random stream is generated out of order with GF(2) jump-ahead: pulse q consumes
//      stream positions [idx_q - idx_0, idx_{q+1} - idx_0), so stream position == sample offset
//      and every 64-sample chunk of normals is produced by one lane from a jumped state --
//      bit-identical to the sequential generator;
//   4. one 256-thread workgroup per pulse builds the periodic + aperiodic responses in LDS
//      (4 real FFTs for the two minimum-phase spectra, 1 for the noise, 2 inverse) and
//      overlap-adds them with f64 atomics.
// PARITY: the reference has no golden waveform; checked against oracle/c/synth.c.
#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "context.h"
#include "wave_fft.h"
#include "world_dev.h"

namespace itts {
using namespace wd;

constexpr int CHUNK = 2048;       // samples per block in the scan kernels (8 per thread)
constexpr double kDefaultF0 = 500.0;
constexpr int RCHUNK = RNG_CHUNK;  // normals per lane in the RNG kernel (context.h)
constexpr int NJUMP = RNG_NJUMP;

struct SynUtt {
  int64_t f_off;   // frames offset
  int T;
  int64_t y_off;   // output samples offset
  int yl;
  int64_t s_off;   // offset into per-sample scratch
  int64_t b_off;   // offset into per-block scratch
  int nblk;
};

struct SynParams {
  int fs, fft, logfft;
  double fp;        // frame period in seconds
  double lowest_f0;
  int n_utts;
};

// WORLD interp1 on the uniform knots cta[j] = j*fp, j = 0..T (T+1 knots), histc semantics.
__device__ __forceinline__ double interp_coarse(const double* __restrict__ f0, int T, double fp, double t,
                                                double lowest, bool want_vuv) {
  int j = (int)(t / fp);
  while ((double)(j + 1) * fp <= t) ++j;
  while (j > 0 && (double)j * fp > t) --j;
  int k = j + 1;  // count of knots <= t
  if (k < 1) k = 1;
  if (k > T) k = T;
  auto knot = [&](int q) -> double {  // value at knot q (0..T)
    auto base = [&](int r) -> double {
      const double v = f0[r] < lowest ? 0.0 : f0[r];
      return want_vuv ? (v == 0.0 ? 0.0 : 1.0) : v;
    };
    if (q < T) return base(q);
    return base(T - 1) * 2 - base(T - 2);
  };
  const double x0 = (double)(k - 1) * fp, x1 = (double)k * fp;
  const double y0 = knot(k - 1), y1 = knot(k);
  const double s = (t - x0) / (x1 - x0);
  return y0 + s * (y1 - y0);
}

// K1: phase increments + V/UV per sample
__global__ __launch_bounds__(NT) void syn_inc_kernel(const double* __restrict__ f0, const SynUtt* __restrict__ utts,
                                                     SynParams p, double* __restrict__ inc,
                                                     uint8_t* __restrict__ vuv, double* __restrict__ y) {
  const SynUtt u = utts[blockIdx.y];
  if ((int)blockIdx.x >= u.nblk) return;
  const double* f = f0 + u.f_off;
  // (a wave's lanes take consecutive samples: eight per thread in a row made every store instruction touch 64 lines --
  // the PMC pass showed 1.14 GB of traffic for a 0.23-GB output)
  const int i0 = blockIdx.x * CHUNK + threadIdx.x;
  for (int r = 0; r < CHUNK / NT; ++r) {
    const int i = i0 + r * NT;
    if (i < u.yl) {
      const double t = i / (double)p.fs;
      const double v = interp_coarse(f, u.T, p.fp, t, p.lowest_f0, true) > 0.5 ? 1.0 : 0.0;
      double fi = interp_coarse(f, u.T, p.fp, t, p.lowest_f0, false);
      if (v == 0.0) fi = kDefaultF0;
      inc[u.s_off + i] = 2.0 * kPi * fi / p.fs;
      vuv[u.s_off + i] = (uint8_t)(v != 0.0);
      y[u.y_off + i] = 0.0;       // (the overlap-add's target: a fill of its own was 64 us in front of this kernel)
    }
  }
}

// exclusive scan of per-block values of each utterance (double or int payload in double)
__global__ void syn_scan_blocks_kernel(const SynUtt* __restrict__ utts, double* __restrict__ vals,
                                       double* __restrict__ totals) {
  const SynUtt u = utts[blockIdx.x];
  if (threadIdx.x != 0) return;
  double run = 0.0;
  for (int b = 0; b < u.nblk; ++b) {
    const double v = vals[u.b_off + b];
    vals[u.b_off + b] = run;
    run += v;
  }
  if (totals) totals[blockIdx.x] = run;
}

// K3: total phase -> wrapped phase.  WORLD accumulates the phase sample by sample; in unvoiced
// regions (default 500 Hz) at fs = 16 kHz the running sum lands on multiples of 2 pi up to
// rounding, so the pulse positions depend on the exact sequential rounding.  The chain
// total += inc[i] is therefore evaluated strictly in order by one lane per utterance (the loads
// and the fmod are done by the whole wave); ~8 cycles per sample, utterances in parallel.
constexpr int SEQ = 1024;
__global__ __launch_bounds__(64) void syn_phase_seq_kernel(const SynUtt* __restrict__ utts,
                                                           double* __restrict__ inc_wrap) {
  __shared__ __attribute__((aligned(16))) double buf[SEQ];
  const SynUtt u = utts[blockIdx.x];
  double* a = inc_wrap + u.s_off;
  double total = 0.0;
  for (int base = 0; base < u.yl; base += SEQ) {
    const int n = min(SEQ, u.yl - base);
    for (int i = threadIdx.x; i < n; i += 64) buf[i] = a[base + i];
    __syncthreads();
    if (threadIdx.x == 0) {
      // (explicitly batched / software-pipelined variants of this loop measured 10-100 % slower
      // than letting hipcc unroll it: the chain, not the LDS latency, is the bound)
#pragma unroll 8
      for (int i = 0; i < n; ++i) {
        total = __dadd_rn(total, buf[i]);
        buf[i] = total;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 64) a[base + i] = fmod(buf[i], 2.0 * kPi);
    __syncthreads();
  }
}

// K3': the same running sum, bit for bit, as a parallel scan.
// While the total stays inside one binade [2^e, 2^(e+1)) it is a multiple of u = 2^(e-52) and
// fl(total + d) is again a multiple of u: total + u * R, where R is d / u rounded to an integer --
// ties going to whichever neighbour makes the SUM even.  In units of u the chain is therefore
// exact integer arithmetic:  S -> S + I + [f > 1/2]            (d / u = I + f, no tie)
//                            S -> S + I + ((S + I) mod 2)      (f = 1/2: round half to even)
// Both are maps of the form S -> S + a[S mod 2]; such pairs (a[0], a[1]) compose associatively,
// (a o b)[p] = a[p] + b[(p + a[p]) mod 2], so a workgroup scans 4096 samples at once.  The scan stops at
// the first sample whose sum leaves the binade (S >= 2^53); that one addition is done in floating
// point (its rounding grid is the next binade's), then the scan resumes there.  A 10 s utterance
// crosses ~20 binades.  Verified bit-identical to syn_phase_seq_kernel (tests/test_gpu_world.py).
struct PhaseMap {
  unsigned long long a0, a1;   // addend for even / odd S (unsigned: sums past the cut may wrap)
};
__device__ __forceinline__ PhaseMap phase_compose(const PhaseMap f, const PhaseMap g) {   // g after f
  PhaseMap r;
  r.a0 = f.a0 + ((f.a0 & 1) ? g.a1 : g.a0);
  r.a1 = f.a1 + (((1 + f.a1) & 1) ? g.a1 : g.a0);
  return r;
}
constexpr unsigned long long kBinadeTop = 1ull << 53;

// fmod(t, y) for 0 <= t < 2^20, y = double(2 pi): the quotient estimate is off by at most one and
// every candidate remainder t - q y is a multiple of ulp(y) below 8, hence exactly representable:
// the fma and the one-step correction are exact, the result is the exact remainder like fmod's.
__device__ __forceinline__ double fmod_2pi(double t) {
  constexpr double y = 2.0 * kPi;
  const double q = floor(t * (1.0 / y));
  double r = fma(-q, y, t);
  if (r < 0.0) r += y;
  else if (r >= y) r -= y;
  return r;
}

// One workgroup of PST threads per utterance, PST x 16 samples per round: the round's increments are
// staged in LDS by coalesced loads (the next round's are already in flight), thread t owns the 16
// consecutive samples 16 t .. 16 t + 15.  An utterance's rounds follow one another (the running sum is carried), so the
// kernel's time is rounds x the latency of a round: 512 threads (two waves per SIMD, 8 192 samples a round) instead of
// 256 halve the rounds for nearly the same round (round 5, with the conflict-free LDS layout below: 1.20 -> 0.91 ms at
// 48 kHz, where an utterance had 71 rounds of 4 096 and only 64 of the 256 CUs have an utterance; unchanged at 16 kHz:
// 0.59 ms).
constexpr int PST = 512, PSW = PST / 64;
// The pulses -- sample i with |wrap[i + 1] - wrap[i]| > pi -- leave with the round that wraps them: counted per thread
// over its sixteen samples, numbered by one scan over the workgroup, written in order to the utterance's list (what
// syn_pulse_count_kernel / syn_scan_blocks_kernel / syn_pulse_emit_kernel did in three more passes over the array:
// 0.29 of the 2 ms in front of the pulse kernel).
__global__ __launch_bounds__(PST) void syn_phase_scan_kernel(const SynUtt* __restrict__ utts,
                                                            double* __restrict__ inc_wrap, int* __restrict__ pidx_all,
                                                            double* __restrict__ ptot) {
  constexpr int PER = 16, BLK = PST * PER;
  // sample i of the round sits at PH(i) = i + i / 16: a thread's sixteen consecutive samples are read and written 17
  // doubles apart from its neighbour's (at 16 apart -- 128 bytes -- all 64 lanes of a wave hit the same bank: every such
  // access took 64 turns)
  __shared__ double buf[BLK + BLK / 16 + 2];
  auto PH = [](int i) { return i + (i >> 4); };
  __shared__ PhaseMap wagg[PSW];
  __shared__ int wcut[PSW];
  __shared__ int wpc[PSW];
  __shared__ double s_total;
  const SynUtt u = utts[blockIdx.x];
  double* a = inc_wrap + u.s_off;
  int* pidx = pidx_all + u.s_off;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double total = 0.0;      // uniform running sum (value after sample `pos - 1`)
  int pos = 0;
  int pc = 0;              // pulses so far (uniform)
  double wprev = 0.0;      // wrapped phase of sample pos - 1 (uniform)
  // a sample wrapped by the whole workgroup at once (the plain-addition paths): its pair with the sample before it
  auto single = [&](double wnew) {
    if (pos >= 1 && fabs(wnew - wprev) > kPi) {
      if (tid == 0) pidx[pc] = pos - 1;
      ++pc;
    }
    wprev = wnew;
  };
  int have = -1;           // first sample of the increments held in `pre` (prefetched), or -1
  double pre[PER];
  while (pos < u.yl) {
    // stage samples pos .. pos + BLK (one extra: the sample a binade crossing would need)
    if (have == pos) {
#pragma unroll
      for (int r = 0; r < PER; ++r) buf[PH(r * PST + tid)] = pre[r];
    } else {
#pragma unroll
      for (int r = 0; r < PER; ++r) {
        const int i = pos + r * PST + tid;
        buf[PH(r * PST + tid)] = i < u.yl ? a[i] : 0.0;
      }
    }
    // next round's increments (used if this round ends without a crossing)
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int i = pos + BLK + r * PST + tid;
      pre[r] = i < u.yl ? a[i] : 0.0;
    }
    have = pos + BLK;
    __syncthreads();

    int n_ok;
    bool crossing_sample = false;        // the round was cut by a sample that leaves the binade ..
    double crossing_value = 0.0;         // .. with this increment
    if (pos == 0) {
      // The first round in sequence, by one lane (eight cycles a sample: 30 us): the total starts at zero and doubles
      // every few samples at first -- thirteen binade crossings in the first 5 000 samples of an utterance that begins
      // unvoiced, each of them a round of its own cut short, a third of the kernel's rounds for 3 % of its samples.
      n_ok = min(BLK, u.yl);
      if (tid == 0) {
        double t = 0.0;
#pragma unroll 8
        for (int i = 0; i < n_ok; ++i) {
          t = __dadd_rn(t, buf[PH(i)]);
          buf[PH(i)] = t;
        }
        s_total = t;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < PER; ++r) {
        const int li = tid * PER + r;
        if (li < n_ok) {
          const double t = buf[PH(li)];
          buf[PH(li)] = t < 1048576.0 ? fmod_2pi(t) : fmod(t, 2.0 * kPi);
        }
      }
      __syncthreads();
    } else {
    int ex;
    frexp(total, &ex);                       // total = m * 2^ex, m in [0.5, 1)
    const int e = ex - 1;
    const double first = buf[PH(0)];
    if (!(total > 0.0) || !(first < ldexp(1.0, e + 2)) || !(first >= 0.0) || e < -900) {
      // a total of zero or a sample that does not fit the integer picture: plain addition
      total = __dadd_rn(total, first);
      const double wnew = fmod(total, 2.0 * kPi);
      if (tid == 0) a[pos] = wnew;
      single(wnew);
      ++pos;
      have = -1;
      __syncthreads();
      continue;
    }
    const double up = ldexp(1.0, 52 - e), down = ldexp(1.0, e - 52);           // exact scalings by u
    const double dmax = ldexp(1.0, e + 2);
    const unsigned long long S0 = (unsigned long long)(total * up);            // in [2^52, 2^53)
    PhaseMap m[PER];
    PhaseMap agg = {0, 0};
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const double d = buf[PH(tid * PER + r)];
      const double x = d * up;                                      // exact: d / u
      // (x < 2^51 as well: the integer part then comes out of the bits of fl + 2^52 -- two instructions where the
      // conversion of a double to a 64-bit integer is twenty; an increment that large against the total does not occur,
      // and if it did the crossing path takes it)
      const bool fits = d >= 0.0 && d < dmax && x < 2251799813685248.0;      // else: forces the crossing path
      const double fl = floor(x);
      const double f = x - fl;                                      // exact
      const unsigned long long I = fits ? (unsigned long long)__double_as_longlong(fl + 4503599627370496.0) - 0x4330000000000000ull
                                        : kBinadeTop;
      if (f == 0.5 && fits) {
        m[r].a0 = I + (I & 1);
        m[r].a1 = I + ((I + 1) & 1);
      } else {
        m[r].a0 = m[r].a1 = I + (f > 0.5 ? 1 : 0);
      }
      agg = r == 0 ? m[0] : phase_compose(agg, m[r]);
    }
    // inclusive scan of the thread aggregates inside the wave, then across the four waves
    PhaseMap inc = agg;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      PhaseMap o;
      o.a0 = (unsigned long long)__shfl_up((long long)inc.a0, off);
      o.a1 = (unsigned long long)__shfl_up((long long)inc.a1, off);
      if (lane >= off) inc = phase_compose(o, inc);
    }
    if (lane == 63) wagg[wv] = inc;
    PhaseMap ex_map;
    ex_map.a0 = (unsigned long long)__shfl_up((long long)inc.a0, 1);
    ex_map.a1 = (unsigned long long)__shfl_up((long long)inc.a1, 1);
    if (lane == 0) ex_map.a0 = ex_map.a1 = 0;
    __syncthreads();
    PhaseMap wpre = {0, 0};
    for (int w = 0; w < wv; ++w) wpre = phase_compose(wpre, wagg[w]);
    ex_map = phase_compose(wpre, ex_map);
    // walk the thread's samples from its entry value; find the first that leaves the binade
    unsigned long long S = S0 + ((S0 & 1) ? ex_map.a1 : ex_map.a0);
    unsigned long long vals[PER];
    int first_out = PER;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      S += (S & 1) ? m[r].a1 : m[r].a0;
      vals[r] = S;
      if (first_out == PER && (S >= kBinadeTop || pos + tid * PER + r >= u.yl)) first_out = r;
    }
    // (a thread past the cut may hold wrapped garbage, which is fine: it is past the cut)
    int cut = first_out < PER ? tid * PER + first_out : BLK;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cut = min(cut, __shfl_xor(cut, off));
    if (lane == 0) wcut[wv] = cut;
    __syncthreads();
    cut = wcut[0];
#pragma unroll
    for (int w = 1; w < PSW; ++w) cut = min(cut, wcut[w]);
    n_ok = min(cut, min(BLK, u.yl - pos));               // samples pos .. pos+n_ok-1 are final
    const double crossing_inc = buf[PH(n_ok < BLK ? n_ok : 0)];        // read before buf is overwritten
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int li = tid * PER + r;
      if (li < n_ok) {
        // (a final value lies in [2^52, 2^53): as a double its bits are the value plus a constant)
        const double t = __longlong_as_double((long long)(vals[r] + 0x4320000000000000ull)) * down;                   // exact
        buf[PH(li)] = t < 1048576.0 ? fmod_2pi(t) : fmod(t, 2.0 * kPi);
        if (li == n_ok - 1) s_total = t;
      }
    }
    __syncthreads();
    crossing_sample = n_ok < BLK;
    crossing_value = crossing_inc;
    }
    {
      // the round's pulses: pair (li - 1, li) of this thread's samples, the first one against its neighbour's last
      // (or the round before's)
      unsigned mask = 0;
      int myc = 0;
      double pw = tid > 0 ? buf[PH(tid * PER - 1)] : wprev;
#pragma unroll
      for (int r = 0; r < PER; ++r) {
        const int li = tid * PER + r;
        if (li < n_ok) {
          const double cw = buf[PH(li)];
          const bool pl = (pos + li >= 1) && fabs(cw - pw) > kPi;
          mask |= (pl ? 1u : 0u) << r;
          myc += pl ? 1 : 0;
          pw = cw;
        }
      }
      int incl = myc;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      if (lane == 63) wpc[wv] = incl;
      __syncthreads();
      int before = pc + incl - myc, round_total = 0;
#pragma unroll
      for (int w = 0; w < PSW; ++w) {
        const int c = wpc[w];
        if (w < wv) before += c;
        round_total += c;
      }
#pragma unroll
      for (int r = 0; r < PER; ++r)
        if ((mask >> r) & 1u) pidx[before++] = pos + tid * PER + r - 1;
      pc += round_total;
      if (n_ok > 0) wprev = buf[PH(n_ok - 1)];
    }
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int li = r * PST + tid;
      if (li < n_ok) a[pos + li] = buf[PH(li)];
    }
    if (n_ok > 0) total = s_total;
    pos += n_ok;
    if (n_ok < BLK) {
      have = -1;                         // the prefetched round no longer lines up
      if (crossing_sample && pos < u.yl) {          // the sample that crosses the binade: one real addition
        total = __dadd_rn(total, crossing_value);
        const double wnew = fmod(total, 2.0 * kPi);
        if (tid == 0) a[pos] = wnew;
        single(wnew);
        ++pos;
      }
    }
    __syncthreads();
  }
  if (tid == 0) ptot[blockIdx.x] = (double)pc;
}

// K4: pulses per block (pulse at sample i when |wrap[i+1]-wrap[i]| > pi, i <= yl-2)
__global__ __launch_bounds__(NT) void syn_pulse_count_kernel(const SynUtt* __restrict__ utts,
                                                             const double* __restrict__ wrap,
                                                             double* __restrict__ pcnt) {
  __shared__ double red[8];
  const SynUtt u = utts[blockIdx.y];
  if ((int)blockIdx.x >= u.nblk) return;
  const double* w = wrap + u.s_off;
  double c = 0.0;
  for (int r = 0; r < CHUNK / NT; ++r) {
    const int i = blockIdx.x * CHUNK + r * NT + threadIdx.x;
    if (i < u.yl - 1 && fabs(w[i + 1] - w[i]) > kPi) c += 1.0;
  }
  c = bsum(c, red);
  if (threadIdx.x == 0) pcnt[u.b_off + blockIdx.x] = c;
}

// K6: ordered compaction of pulse indices
__global__ __launch_bounds__(NT) void syn_pulse_emit_kernel(const SynUtt* __restrict__ utts,
                                                            const double* __restrict__ wrap,
                                                            const double* __restrict__ poff,
                                                            int* __restrict__ pidx, SynParams p) {
  __shared__ int wcnt[4];
  __shared__ int base_s;
  const SynUtt u = utts[blockIdx.y];
  if ((int)blockIdx.x >= u.nblk) return;
  const double* w = wrap + u.s_off;
  if (threadIdx.x == 0) base_s = (int)poff[u.b_off + blockIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int r = 0; r < CHUNK / NT; ++r) {
    const int i = blockIdx.x * CHUNK + r * NT + threadIdx.x;
    const bool pulse = (i < u.yl - 1) && fabs(w[i + 1] - w[i]) > kPi;
    const unsigned long long bal = __ballot(pulse);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wv] = __popcll(bal);
    __syncthreads();
    int off = base_s;
    for (int q = 0; q < wv; ++q) off += wcnt[q];
    if (pulse) pidx[u.s_off + off + before] = i;
    __syncthreads();
    if (threadIdx.x == 0) base_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
}

// flat pulse numbering over the batch: gpoff[u] = sum_{u'<u} P_u', gpoff[U] = total
// exclusive sums of n values by ONE wave, 64 at a time (a lone thread walking 256 utterances took 45-70 us right in
// front of the pulse kernels: a dependent load and store per utterance)
template <class Load>
__device__ __forceinline__ void wave_offsets(int n, Load load, int64_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  int64_t carry = 0;
  for (int base = 0; base < n; base += 64) {
    const int u = base + lane;
    const int64_t v = u < n ? load(u) : 0;
    int64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int64_t o = __shfl_up(inc, off);
      if (lane >= off) inc += o;
    }
    if (u < n) out[u] = carry + inc - v;
    carry += __shfl(inc, 63);
  }
  if (lane == 0) out[n] = carry;
}
__device__ void syn_dcr_table(int fft, double* __restrict__ dcr);
// one launch of two workgroups between the phase scan and the split of the pulses by kind: block 0 numbers the pulses over
// the batch (its first wave) and clears the split's counts, block 1 tabulates the DC remover's window (three launches
// before: 24 + 13 + 6 us on the way to the pulse kernels)
__global__ __launch_bounds__(NT) void syn_pulse_offsets_kernel(const double* __restrict__ ptot, int n_utts,
                                                               int64_t* __restrict__ gpoff, int* __restrict__ kcnt,
                                                               int kcnt_ints, int fft, double* __restrict__ dcr) {
  if (blockIdx.x == 0) {
    if (threadIdx.x < 64) wave_offsets(n_utts, [&](int u) { return (int64_t)ptot[u]; }, gpoff);
    for (int i = threadIdx.x; i < kcnt_ints; i += NT) kcnt[i] = 0;
  } else if (blockIdx.x == 1) {
    syn_dcr_table(fft, dcr);
  }
}

// ---- WORLD randn stream with jump-ahead ---------------------------------------------------------
__device__ __forceinline__ void xs_step(uint32_t& x, uint32_t& y, uint32_t& z, uint32_t& w) {
  const uint32_t t = x ^ (x << 11);
  x = y; y = z; z = w;
  w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
}

__global__ __launch_bounds__(NT) void syn_randn_kernel(const SynUtt* __restrict__ utts,
                                                       const JumpTable* __restrict__ jt,
                                                       double* __restrict__ R) {
  const SynUtt u = utts[blockIdx.y];
  const int chunk = blockIdx.x * NT + threadIdx.x;
  const int n0 = chunk * RCHUNK;
  if (n0 >= u.yl) return;
  const uint4 st = rng_chunk_state(jt, chunk);
  uint32_t x = st.x, y = st.y, z = st.z, w = st.w;
  double* out = R + u.s_off + n0;
  const int cnt = min(RCHUNK, u.yl - n0);
  for (int i = 0; i < cnt; ++i) {
    xs_step(x, y, z, w);
    uint32_t tmp = w >> 4;
    for (int q = 0; q < 11; ++q) {
      xs_step(x, y, z, w);
      tmp += w >> 4;
    }
    out[i] = tmp / 268435456.0 - 6.0;
  }
}

// ---- one workgroup per pulse ------------------------------------------------------------------------
struct PulseArgs {
  const double* f0;
  const double* sp;
  const double* ap;
  const SynUtt* utts;
  const int64_t* gpoff;
  const double* ptot;
  const int* pidx;
  const double* wrap;
  const uint8_t* vuv;
  const double* R;
  double* y;
  SynParams p;
  const double2* g_tw;
  const double* dcr;   // [h + 1]: the Hann half window of RemoveDCComponent, dcr[h] = its doubled sum
  const int* kq;           // syn_pulse_wave_kernel: the pulses of its kind, per utterance (see syn_pulse_split_kernel)
  const int64_t* kgpoff;   //                        [n_utts + 1] flat numbering of that kind over the batch
  int* next;               //                        the kind's pulses handed out so far (zero at launch)
};

// The pulses by kind (syn_pulse_wave_kernel has a kernel per kind).  kq holds, in the utterance's own stretch of a
// per-sample int array (yl entries; an utterance has fewer pulses than samples), the numbers -- positions in the
// utterance's pulse list -- of its unvoiced pulses from the front and of its voiced ones from the back; kcnt[u],
// kcnt[n_utts + u] count them.  A wave's records go behind one atomic; the order inside a kind is that of the appends
// (the overlap-add is a sum of atomics either way).
__global__ __launch_bounds__(NT) void syn_pulse_split_kernel(const SynUtt* __restrict__ utts, const double* __restrict__ ptot,
                                                             const int* __restrict__ pidx, const uint8_t* __restrict__ vuv,
                                                             int n_utts, int* __restrict__ kq, int* __restrict__ kcnt) {
  const int ui = blockIdx.y;
  const SynUtt u = utts[ui];
  const int Pn = (int)ptot[ui];
  const int lane = threadIdx.x & 63;
  for (int q0 = blockIdx.x * NT; q0 < Pn; q0 += gridDim.x * NT) {
    const int qi = q0 + (int)threadIdx.x;
    const bool live = qi < Pn;
    const bool voiced = live && vuv[u.s_off + pidx[u.s_off + qi]] != 0;
    for (int kind = 0; kind < 2; ++kind) {
      const bool mine = live && (voiced == (kind == 1));
      const unsigned long long bal = __ballot(mine);
      if (bal == 0) continue;
      int base = 0;
      if (lane == 0) base = atomicAdd(kcnt + kind * n_utts + ui, __popcll(bal));
      base = __shfl(base, 0) + __popcll(bal & ((1ull << lane) - 1ull));
      if (mine) kq[u.s_off + (kind == 0 ? base : u.yl - 1 - base)] = qi;
    }
  }
}

// flat numbering of each kind over the batch: kgp[kind][u] = sum over u' < u, kgp[kind][n_utts] = total
__global__ __launch_bounds__(128) void syn_kind_offsets_kernel(const int* __restrict__ kcnt, int n_utts,
                                                               int64_t* __restrict__ kgp) {
  const int kind = threadIdx.x >> 6;      // a wave per kind
  if (blockIdx.x == 0 && kind < 2)
    wave_offsets(n_utts, [&](int u) { return (int64_t)kcnt[kind * n_utts + u]; }, kgp + kind * (n_utts + 1));
}

// minimum phase spectrum of the log-amplitude lg[0..h] (in z.x of the first h+1 entries is NOT
// assumed): input array `lg`, output mp[0..h] complex. Uses z as FFT scratch.  lg is consumed by
// the first loop, so it may live inside mp's storage.
template <int CFFT = 0>      // CFFT: log2 of the transform size as a compile-time constant (0: any)
__device__ inline void min_phase(const double* lg, int fft_rt, int logfft, double2* z, const double2* tw,
                                 double2* mp) {
  const int fft = CFFT ? (1 << CFFT) : fft_rt;
  const int h = fft / 2;
  double* zr = reinterpret_cast<double*>(z);
  for (int k = tid(); k <= h; k += NT) {
    const double v = lg[k];
    zr[k] = v;
    if (k > 0 && k < h) zr[fft - k] = v;
  }
  __syncthreads();
  rfft_lds<true, CFFT, CFFT ? CFFT - 1 : 0>(z, fft, logfft, tw, fft);  // real even input -> real spectrum = fft * cepstrum
  // fold: c[0], 2 c[1..h-1], c[h], zeros; keep the (real) values, build the real sequence
  for (int k = tid(); k <= h; k += NT) mp[k].x = z[k].x * ((k == 0 || k == h) ? 1.0 : 2.0);
  __syncthreads();
  for (int k = tid(); k <= h; k += NT) zr[k] = mp[k].x;
  for (int k = h + 1 + tid(); k < fft + 2; k += NT) zr[k] = 0.0;
  __syncthreads();
  rfft_lds<true, CFFT, CFFT ? CFFT - 1 : 0>(z, fft, logfft, tw, fft);
  for (int k = tid(); k <= h; k += NT) {
    const double t = exp(z[k].x / fft);
    double sn, cs;
    sincos_mid(z[k].y / fft, &sn, &cs);
    mp[k] = make_double2(t * cs, t * sn);
  }
  __syncthreads();
}

// WORLD's dc_remover window depends on the transform size only: one workgroup tabulates
// hann(i) = 0.5 - 0.5 cos(2 pi (i + 1) / (1 + fft)), i < fft / 2, and the normaliser (the doubled sum,
// reduced in the order the pulse kernel used when every pulse recomputed it: same bits).  Three
// fp64 cosines per bin and pulse less.
__device__ void syn_dcr_table(int fft, double* __restrict__ dcr) {
  __shared__ double red[16];
  const int h = fft / 2;
  double dsum = 0.0;
  for (int i = tid(); i < h; i += NT) {
    const double w = 0.5 - 0.5 * cos(2.0 * kPi * (i + 1.0) / (1.0 + fft));
    dcr[i] = w;
    dsum += w * 2.0;
  }
  dsum = bsum(dsum, red);
  if (tid() == 0) dcr[h] = dsum;
}

// 28 KB of LDS at fft 1024: five workgroups fit a CU when a wave needs <= 102 VGPRs
// UNV: the kernel of the unvoiced pulses (see syn_pulse_wave_kernel: a kernel per kind): no periodic response, so no
// `per` array, and the log amplitudes go straight to min_phase's input inside mp -- 33 KB instead of 49 at fft 2048, four
// workgroups per CU instead of three for nine pulses in ten
constexpr size_t syn_pulse_lds_bytes(int h, bool unv) {
  return 2 * (size_t)(h + 1) * 16 + (unv ? 0 : (size_t)(h + 2) * 8 + (size_t)h * 8) + 16 * 8;
}
template <int CFFT, bool UNV>      // CFFT: log2 of the transform size as a compile-time constant (11 at 32 .. 48 kHz), 0: any size
__global__ __launch_bounds__(NT, 5) void syn_pulse_kernel(PulseArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int fft = CFFT ? (1 << CFFT) : a.p.fft, logfft = CFFT ? CFFT : a.p.logfft, h = fft / 2, K = h + 1;
  char* q = smem;
  const double2* tw = a.g_tw;          // compact table read through the cache: 8 KB of LDS less
  double2* z = reinterpret_cast<double2*>(q); q += (size_t)(h + 1) * 16;
  double2* mp = reinterpret_cast<double2*>(q); q += (size_t)(h + 1) * 16;
  double* lg = reinterpret_cast<double*>(mp);      // input of min_phase, dead before mp is written
  // the aperiodic response's log amplitudes, formed together with the periodic ones from ONE read of the four
  // spectrum rows (rounds 1-4 kept the interpolated envelope and aperiodicity ratio in two LDS arrays instead:
  // 8 KB more at fft 2048, i.e. two workgroups per CU where three fit now)
  double* lgs = lg;
  double* per = nullptr;
  if (!UNV) {
    lgs = reinterpret_cast<double*>(q); q += (size_t)(K + 1) * 8;
    per = reinterpret_cast<double*>(q); q += (size_t)h * 8;
  }
  double* red = reinterpret_cast<double*>(q);
  double* zr = reinterpret_cast<double*>(z);

  const int64_t total = a.kgpoff[a.p.n_utts];
  const int64_t g = blockIdx.x;           // one pulse of this kind per workgroup (the host reads the counts)
  if (g < total) {
    // utterance of pulse g of this kind: every wave reads the offsets 64 at a time and counts (a binary search is
    // log2(U) dependent trips to memory in front of everything else the pulse does)
    const int lo = find_utt_wave(a.kgpoff, a.p.n_utts, g);
    const SynUtt u = a.utts[lo];
    const int P = (int)a.ptot[lo];
    const int kn = (int)(g - a.kgpoff[lo]);
    const int qi = a.kq[u.s_off + (UNV ? kn : u.yl - 1 - kn)];
    const int* pidx = a.pidx + u.s_off;
    const int idx = pidx[qi];
    const int idx_next = pidx[min(P - 1, qi + 1)];
    const int noise_size = idx_next - idx;
    const double* wrap = a.wrap + u.s_off;
    const double y1 = wrap[idx] - 2.0 * kPi, y2 = wrap[idx + 1];
    const double tshift = (-y1 / (y2 - y1)) / a.p.fs;
    const double t = idx / (double)a.p.fs;
    const double vuv = UNV ? 0.0 : 1.0;            // (the kind of the list)
    const int T = u.T;
    int fl = (int)floor(t / a.p.fp), ce = (int)ceil(t / a.p.fp);
    if (fl > T - 1) fl = T - 1;
    if (ce > T - 1) ce = T - 1;
    const double al = t / a.p.fp - fl;
    const double* sp0 = a.sp + (u.f_off + fl) * K;
    const double* sp1 = a.sp + (u.f_off + ce) * K;
    const double* ap0 = a.ap + (u.f_off + fl) * K;
    const double* ap1 = a.ap + (u.f_off + ce) * K;
    // spectral envelope and aperiodicity ratio of bin k at the pulse
    auto se_ar = [&](int k, double& se, double& ar) {
      const double s0 = fabs(sp0[k]);
      double a0 = ap0[k];
      a0 = a0 > 0.999999999999 ? 0.999999999999 : a0;
      a0 = a0 < 0.001 ? 0.001 : a0;
      if (fl == ce) {
        se = s0;
        ar = a0 * a0;                    // pow(x, 2.0): compilers fold it to x * x
      } else {
        const double s1 = fabs(sp1[k]);
        double a1 = ap1[k];
        a1 = a1 > 0.999999999999 ? 0.999999999999 : a1;
        a1 = a1 < 0.001 ? 0.001 : a1;
        se = (1.0 - al) * s0 + al * s1;
        ar = (1.0 - al) * (a0 * a0) + al * (a1 * a1);
      }
    };
    // ---- periodic response
    bool has_per = false;
    if (vuv != 0.0) {
      double se0, ar0;
      se_ar(0, se0, ar0);
      has_per = !(ar0 > 0.999);
      for (int k = tid(); k < K; k += NT) {
        double se, ar;
        se_ar(k, se, ar);
        if (has_per) lg[k] = log_pos(se * (1.0 - ar) + kEps) / 2.0;
        lgs[k] = log_pos(se * ar) / 2.0;
      }
    } else {
      // an unvoiced pulse: the envelope alone (the aperiodicity rows are not read)
      for (int k = tid(); k < K; k += NT) {
        const double s0 = fabs(sp0[k]);
        lgs[k] = log_pos(fl == ce ? s0 : (1.0 - al) * s0 + al * fabs(sp1[k])) / 2.0;
      }
    }
    __syncthreads();
    double per_dc = 0.0, per_dsum = 1.0;
    if (has_per) {
      min_phase<CFFT>(lg, fft, logfft, z, tw, mp);
      const double coef = 2.0 * kPi * tshift * a.p.fs / fft;
      for (int k = tid(); k < K; k += NT) {
        const double re2 = cos_mid(coef * k);
        const double im2 = sqrt(1.0 - re2 * re2);
        const double2 m = mp[k];
        z[k] = make_double2(m.x * re2 + m.y * im2, m.y * re2 - m.x * im2);
      }
      __syncthreads();
      irfft_lds<true, CFFT, CFFT ? CFFT - 1 : 0>(z, fft, logfft, tw, fft);
      // fftshift + DC removal
      double dc = 0.0;
      for (int i = tid(); i < h; i += NT) dc += zr[i];  // shifted index i+h <- zr[i]
      dc = bsum(dc, red);
      // dc_remover[i] = hann(i) / sum, symmetric (tabulated per transform size)
      const double dsum = a.dcr[h];
      // shifted response minus its DC share: y[i] = x[i+h] - dc * dcr (i < h; x is zero there, so
      // that half is recomputed at the overlap-add), y[i] = x[i-h] - dc * dcr (i >= h; kept)
      for (int i = tid(); i < h; i += NT) {
        const int m = h - 1 - i;                  // = fft - 1 - (i + h)
        const double dcr = a.dcr[m] / dsum;
        per[i] = zr[i] - dc * dcr;
      }
      per_dc = dc;
      per_dsum = dsum;
      __syncthreads();
    }
    // ---- aperiodic response: minimum-phase spectrum first (into mp), then the noise spectrum in z,
    // multiplied in place
    min_phase<CFFT>(lgs, fft, logfft, z, tw, mp);
    {
      const double* R = a.R + u.s_off + (idx - pidx[0]);
      double s = 0.0;
      for (int i = tid(); i < fft + 2; i += NT) {
        double v = 0.0;
        if (i < noise_size && i < fft) {
          v = R[i];
          s += v;
        }
        zr[i] = v;
      }
      s = bsum(s, red);
      if (noise_size > 0) {
        const double avg = s / noise_size;
        for (int i = tid(); i < noise_size && i < fft; i += NT) zr[i] -= avg;
      }
      __syncthreads();
      rfft_lds<true, CFFT, CFFT ? CFFT - 1 : 0>(z, fft, logfft, tw, fft);
    }
    for (int k = tid(); k < K; k += NT) {
      const double2 m = mp[k], n = z[k];
      z[k] = make_double2(m.x * n.x - m.y * n.y, m.x * n.y + m.y * n.x);
    }
    __syncthreads();
    irfft_lds<true, CFFT, CFFT ? CFFT - 1 : 0>(z, fft, logfft, tw, fft);
    // ---- overlap-add
    const double sq = sqrt((double)noise_size);
    const int off = idx - h + 1;
    double* y = a.y + u.y_off;
    for (int j = tid(); j < fft; j += NT) {
      const int tgt = j + off;
      if (tgt >= 0 && tgt < u.yl) {
        const double apv = (j < h) ? zr[j + h] : zr[j - h];  // fftshift
        double pv = 0.0;
        if (has_per) {
          if (j < h) {
            const double dcr = a.dcr[j] / per_dsum;
            pv = -per_dc * dcr;
          } else {
            pv = per[j - h];
          }
        }
        const double v = (has_per ? pv * sq : 0.0) + apv;
        atomicAdd(&y[tgt], v);
      }
    }
    __syncthreads();
  }
}

// ---- one WAVE per pulse (fft = 1024: 16 .. 24 kHz) ------------------------------------------------
// The same pulse as syn_pulse_kernel, by one wavefront: the 513 bins of a spectrum live eight per
// lane in registers (bin lane + 64 q in register q; bin 512 is carried by every lane), the seven real
// transforms are wf::rfft1024 / irfft1024 (wave_fft.h: register passes, the wave's own 8.5 KB of LDS
// for the transposes, no workgroup barrier), the responses stay in registers up to the overlap-add.
// Persistent: 8 or 12 waves per CU take the pulses from a counter, a few neighbours at a time (SYN_WAVE_DEAL_*), so the
// host no longer needs the pulse count.
// Spectra and responses are those of syn_pulse_kernel bit for bit; the two wave-wide sums (DC of the
// periodic response, mean of the noise) add up in another order.
// wave-uniform values said so (the pulse's scalars then live in SGPRs instead of one VGPR each)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni(int64_t v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double uni(double v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__double2loint(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)__double2hiint(v));
  return __hiloint2double((int)hi, (int)lo);
}
template <class T>
__device__ __forceinline__ const T* uni(const T* p) { return reinterpret_cast<const T*>(uni((int64_t)reinterpret_cast<uintptr_t>(p))); }

__device__ __forceinline__ double wave_bcast0(double v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__double2loint(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)__double2hiint(v));
  return __hiloint2double((int)hi, (int)lo);
}

// minimum phase spectrum of the log-amplitude v (layout A; v[8] = bin 512, the same in every lane) ->
// mp (same layout).  The two real transforms run as two trips through one copy of the code.
__device__ __forceinline__ void min_phase_wave(double (&v)[9], const wf::Plan512& P, double2 (&mp)[9]) {
  constexpr int fft = 1024;
  double2 z[8], x512;
#pragma unroll 1
  for (int r = 0; r < 2; ++r) {
    double v8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v8[q] = v[q];
    wf::pack_real(v8, v[8], z, P, r == 0);
    wf::rfft1024(z, x512, P);            // r = 0: real even input -> real spectrum = fft * cepstrum
    if (r == 0) {
      // fold: c[0], 2 c[1 .. h-1], c[h], zeros
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = z[q].x * ((q == 0 && wf::lane_id() == 0) ? 1.0 : 2.0);
      v[8] = wave_bcast0(x512.x) * 1.0;
    }
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const double zr = q < 8 ? z[q].x : wave_bcast0(x512.x), zi = q < 8 ? z[q].y : wave_bcast0(x512.y);
    const double t = fm::fexp(zr / fft);      // (the log amplitude of a spectrum: within +-373)
    double sn, cs;
    // The short reduction of fastmath.h directly: sincos_mid's out-of-line fallback for |x| > 1e5 is a CALL,
    // and one call in the kernel costs its whole register allocation (75 -> 19 spilled registers
    // without it).  The phase cannot get there: it is the conjugate function of the log amplitude lg,
    // |phase| <= (4 / pi) (1 + 1/3 + ... + 1/511) max|lg| < 5.3 max|lg|, and |lg| <= 373 for anything a
    // double holds -- below 2 000 rad.  (A NaN or an infinite phase gives NaN here as it does there.)
    fm::fsincos(zi / fft, &sn, &cs);
    mp[q] = make_double2(t * cs, t * sn);
    if (q % 3 == 2) __builtin_amdgcn_sched_barrier(0);     // three bins in flight are enough; nine do not fit the registers
  }
}

#ifndef SYN_WAVE_OCC
#define SYN_WAVE_OCC 2
#endif
#ifndef SYN_WAVE_DIAG
#define SYN_WAVE_DIAG 0
#endif
// pulses a wave takes from its kind's list at a time (0: dealt by stride, as until late in round 5)
#ifndef SYN_WAVE_DEAL_UNV
#define SYN_WAVE_DEAL_UNV 4
#endif
#ifndef SYN_WAVE_DEAL_VOI
#define SYN_WAVE_DEAL_VOI 2
#endif
// A kernel per kind of pulse (round 5): UNV = the unvoiced ones -- nine in ten for speech: a pulse every 2 ms in
// unvoiced frames, one per pitch period in voiced ones --, which have no periodic response: without its registers (and
// the DC remover's table and the stash of log amplitudes in LDS) the kernel fits THREE waves per SIMD (168 registers,
// 47 KB a workgroup) where the general one spills 124 registers at that size; the voiced pulses keep the kernel as it
// was, at two.  The pulses of a kind come as a list (syn_pulse_split_kernel).
constexpr int syn_wave_lds_bytes(bool unv) {
  return wf::WF_TABLE_BYTES + (unv ? 0 : 512 * 8) + (NT / 64) * (wf::WF_LDS_BYTES + (unv ? 0 : (512 + 4) * 8));
}
template <bool UNV>
__global__ __launch_bounds__(NT, UNV ? 3 : SYN_WAVE_OCC) void syn_pulse_wave_kernel(PulseArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int fft = 1024, h = 512, K = 513;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int l0 = wf::lane_id();
  wf::Plan512 P;
  // the DC remover's window over its sum (what every pulse divides out again and again), once per workgroup
  double* dcrn = reinterpret_cast<double*>(smem + wf::WF_TABLE_BYTES);
  if (!UNV) {
    const double dsum = a.dcr[h];
    for (int i = threadIdx.x; i < h; i += NT) dcrn[i] = a.dcr[i] / dsum;
  }
  wf::table512_init(smem, a.g_tw);
  constexpr int kWaveLds = wf::WF_LDS_BYTES + (UNV ? 0 : (h + 4) * 8);      // exchange buffer + the stash of 513 log amplitudes
  char* wave_lds = smem + wf::WF_TABLE_BYTES + (UNV ? 0 : h * 8) + (size_t)wv * kWaveLds;
  double* lgs = reinterpret_cast<double*>(wave_lds + wf::WF_LDS_BYTES);      // (not there, and not touched, with UNV)
  wf::plan512_init(P, a.g_tw, wave_lds, smem);
  const double2* dcrn2 = reinterpret_cast<const double2*>(dcrn);
  const int64_t total = a.kgpoff[a.p.n_utts];
  const int64_t nw = (int64_t)gridDim.x * (NT / 64);
  int lo = 0;
  constexpr int kDeal = UNV ? SYN_WAVE_DEAL_UNV : SYN_WAVE_DEAL_VOI;
#if SYN_WAVE_DEAL_UNV && SYN_WAVE_DEAL_VOI
  // The pulses are handed out kDeal at a time behind a counter (a wave's pulses still only grow).  Dealt by stride
  // -- pulse g to wave g mod nw -- the kernel lasted as long as its last workgroup to START: one that finds its CU
  // taken by the tail of another stream's kernel (the caller's decode_aperiodicity) starts when some workgroup has
  // FINISHED its share, and does its own share behind everybody else's (7.2 instead of 5.4 ms when the two met,
  // profiles/r5ap_timeline_noise_on_side_stream.txt).
  (void)nw;
  int64_t g = 0, g_end = 0;
  for (;; ++g) {
    if (g >= g_end) {
      int base = 0;
      if (l0 == 0) base = atomicAdd(a.next, kDeal);
      g = uni(base);
      if (g >= total) break;
      g_end = min(g + kDeal, total);
    }
#else
  (void)kDeal;
  for (int64_t g = (int64_t)blockIdx.x * (NT / 64) + wv; g < total; g += nw) {
#endif
    // opaque per pulse: what derives from the lane number is a few integer operations; hoisted out of the
    // loop it is a dozen registers held for the whole kernel
    int l = l0;
    asm volatile("" : "+v"(l));
    while (uni(a.kgpoff[lo + 1]) <= g) ++lo;          // utterance of pulse g of this kind (g only grows)
    const int64_t u_foff = uni(a.utts[lo].f_off), u_yoff = uni(a.utts[lo].y_off), u_soff = uni(a.utts[lo].s_off);
    const int T = uni(a.utts[lo].T), u_yl = uni(a.utts[lo].yl);
    const int Pn = uni((int)a.ptot[lo]);
    const int kn = (int)(g - uni(a.kgpoff[lo]));
    const int qi = uni(a.kq[u_soff + (UNV ? kn : u_yl - 1 - kn)]);
    const int* pidx = a.pidx + u_soff;
    const int idx = uni(pidx[qi]);
    const int idx_next = uni(pidx[min(Pn - 1, qi + 1)]);
    const int idx_first = uni(pidx[0]);
    const int noise_size = idx_next - idx;
    const double* wrap = a.wrap + u_soff;
    const double y1 = uni(wrap[idx]) - 2.0 * kPi, y2 = uni(wrap[idx + 1]);
    const double tshift = uni((-y1 / (y2 - y1)) / a.p.fs);
    const double t = idx / (double)a.p.fs;
    const double vuv = UNV ? 0.0 : 1.0;                 // (the kind of the list)
    int fl = uni((int)floor(t / a.p.fp)), ce = uni((int)ceil(t / a.p.fp));
    if (fl > T - 1) fl = T - 1;
    if (ce > T - 1) ce = T - 1;
    const double al = uni(t / a.p.fp - fl);
    const double* sp0 = a.sp + (u_foff + fl) * K;
    const double* sp1 = a.sp + (u_foff + ce) * K;
    const double* ap0 = a.ap + (u_foff + fl) * K;
    const double* ap1 = a.ap + (u_foff + ce) * K;
    // spectral envelope and aperiodicity ratio of bin k at the pulse
    auto se_ar = [&](int k, double& se, double& ar) {
      const double s0 = fabs(sp0[k]);
      if (vuv == 0.0) {
        // an unvoiced pulse: the envelope alone (its aperiodicity ratio is never used: the rows are not read)
        se = fl == ce ? s0 : (1.0 - al) * s0 + al * fabs(sp1[k]);
        ar = 1.0;
        return;
      }
      double a0 = ap0[k];
      a0 = a0 > 0.999999999999 ? 0.999999999999 : a0;
      a0 = a0 < 0.001 ? 0.001 : a0;
      if (fl == ce) {
        se = s0;
        ar = a0 * a0;
      } else {
        const double s1 = fabs(sp1[k]);
        double a1 = ap1[k];
        a1 = a1 > 0.999999999999 ? 0.999999999999 : a1;
        a1 = a1 < 0.001 ? 0.001 : a1;
        se = (1.0 - al) * s0 + al * s1;
        ar = (1.0 - al) * (a0 * a0) + al * (a1 * a1);
      }
    };
    double se0, ar0;
    se_ar(0, se0, ar0);
    const bool has_per = UNV ? false : !(ar0 > 0.999);
    const double coef = uni(2.0 * kPi * tshift * a.p.fs / fft);
    double2 per[4] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
    double per_dc = 0.0;
    double2 z[8], x512;
    // two trips through one copy of the code: the periodic response (skipped for unvoiced pulses), then
    // the aperiodic one -- [log amplitude -> minimum phase] x [time shift | spectrum of the noise] -> response
#pragma unroll 1
    for (int half = has_per ? 0 : 1; half < 2; ++half) {
      double v[9];
      if (half == 1 && has_per) {
        // the aperiodic log amplitudes were formed on the first trip (one read of the four rows)
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = lgs[l + 64 * q];
        v[8] = lgs[h];
      } else {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          double se, ar;
          se_ar(q < 8 ? l + 64 * q : h, se, ar);
          const double xa = (vuv != 0.0) ? se * ar : se;
          if (half == 0) {
            v[q] = log_pos(se * (1.0 - ar) + kEps) / 2.0;
            const double w = log_pos(xa) / 2.0;
            if (q < 8) lgs[l + 64 * q] = w; else lgs[h] = w;
          } else {
            v[q] = log_pos(xa) / 2.0;
          }
          if (q % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
        wf::wave_sync();
      }
      double2 mp[9];
      min_phase_wave(v, P, mp);
      if (half == 0) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const int k = q < 8 ? l + 64 * q : h;
          const double re2 = fm::fcos(coef * k);        // |coef * k| <= pi: the time shift is below one sample
          const double im2 = sqrt(1.0 - re2 * re2);
          const double2 m = mp[q];
          const double2 r = make_double2(m.x * re2 + m.y * im2, m.y * re2 - m.x * im2);
          if (q < 8) z[q] = r; else x512 = r;
          if (q % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        const double* R = a.R + u_soff + (idx - idx_first);
        double sacc = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int i = 2 * (l + 64 * q);
          const double v0 = (i < noise_size) ? R[i] : 0.0;
          const double v1 = (i + 1 < noise_size) ? R[i + 1] : 0.0;
          z[q] = make_double2(v0, v1);
          sacc += v0 + v1;
        }
        sacc = wave_sum(sacc);
        if (noise_size > 0) {
          const double avg = sacc / noise_size;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int i = 2 * (l + 64 * q);
            if (i < noise_size) z[q].x -= avg;
            if (i + 1 < noise_size) z[q].y -= avg;
          }
        }
        wf::rfft1024(z, x512, P);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const double2 m = mp[q], n = z[q];
          z[q] = make_double2(m.x * n.x - m.y * n.y, m.x * n.y + m.y * n.x);
        }
        const double nx = wave_bcast0(x512.x), ny = wave_bcast0(x512.y);
        x512 = make_double2(mp[8].x * nx - mp[8].y * ny, mp[8].x * ny + mp[8].y * nx);
      }
      wf::irfft1024(z, x512, P);       // z[q] = (x[2m], x[2m+1]), m = lane + 64 q
      if (half == 0) {
        // fftshift + DC removal: the kept half of the shifted response is x[0 .. h)
        double dc = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) dc += z[q].x + z[q].y;
        dc = wave_sum(dc);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double2 d = dcrn2[255 - l - 64 * q];         // (dcr[h - 2 - i], dcr[h - 1 - i]) / sum, i = 2 (l + 64 q)
          per[q] = make_double2(z[q].x - dc * d.y, z[q].y - dc * d.x);
        }
        per_dc = dc;
      }
    }
    // ---- overlap-add: sample i of a response lands at j = (i + h) mod fft (fftshift).  The sums go
    // through the wave's LDS once so that an atomic instruction covers 64 consecutive samples.
    const double sq = sqrt((double)noise_size);
    const int off = idx - h + 1;
    double* y = a.y + u_yoff;
    {
      double2* s2 = reinterpret_cast<double2*>(wave_lds);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        double2 pv = make_double2(0.0, 0.0);
        if (has_per) {
          if (q >= 4) {
            const double2 d = dcrn2[l + 64 * (q - 4)];      // j = i - h = 2 (l + 64 (q - 4))
            pv = make_double2(-per_dc * d.x, -per_dc * d.y);
          } else {
            pv = per[q];
          }
        }
        const double2 v = make_double2((has_per ? pv.x * sq : 0.0) + z[q].x, (has_per ? pv.y * sq : 0.0) + z[q].y);
        s2[l + 64 * ((q + 4) & 7)] = v;                     // slot j / 2
      }
      wf::wave_sync();
      const double* s1 = reinterpret_cast<const double*>(wave_lds);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 64 * r + l;
        const int tgt = j + off;
        const double v = s1[j];
        if (tgt >= 0 && tgt < u_yl) {
#if SYN_WAVE_DIAG == 1
          if (v == 1.2345e300) y[tgt] = v;
#else
          atomicAdd(&y[tgt], v);
#endif
        }
      }
      wf::wave_sync();
    }
  }
}

// y f64 -> out (f32 and/or f64) through float32 rounding (utterances are stored back to back: one
// flat pass)
__global__ void syn_cast_kernel(const double* __restrict__ y, int64_t n, float* __restrict__ out_f32,
                                double* __restrict__ out_f64) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = (float)y[i];
    if (out_f32) out_f32[i] = v;
    if (out_f64) out_f64[i] = (double)v;
  }
}

// ... and 1st-order de-pre-emphasis lfilter([1], [1, -pre]) evaluated in f64 on the f32-rounded
// samples (as scipy does): v[i] = x[i] + pre * v[i-1].  The recurrence forgets: a sample `warm`
// steps back contributes pre^warm <= 2^-64 of its value, below the rounding of the running sum, so
// every thread owns `seg` consecutive samples and starts `warm` samples earlier from a zero state
// (from the true start of the utterance when that is closer).  grid (ceil(max_yl/seg/64), U).
__global__ __launch_bounds__(64) void syn_deemph_kernel(const double* __restrict__ y,
                                                        const SynUtt* __restrict__ utts, double pre,
                                                        int seg, int warm, float* __restrict__ out_f32,
                                                        double* __restrict__ out_f64) {
  const SynUtt u = utts[blockIdx.y];
  const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t lo = c * seg;
  if (lo >= u.yl) return;
  const int64_t hi = min((int64_t)u.yl, lo + seg);
  const double* x = y + u.y_off;
  double prev = 0.0;
  // (product and sum in separate statements: lfilter rounds twice, an fma would round once)
  for (int64_t i = max((int64_t)0, lo - warm); i < lo; ++i) {
    const double fb = pre * prev;
    prev = (double)(float)x[i] + fb;
  }
  for (int64_t i = lo; i < hi; ++i) {
    const double fb = pre * prev;
    const double v = (double)(float)x[i] + fb;
    prev = v;
    if (out_f32) out_f32[u.y_off + i] = (float)v;
    if (out_f64) out_f64[u.y_off + i] = v;
  }
}

}  // namespace itts

using namespace itts;

// spectra_ready / ap_ready: events behind which d_sp / d_ap are complete (or null: it is, on `stream`).  Everything up
// to the pulse kernels -- the per-sample phase, the pulse positions, the noise -- reads d_f0 only (a third of a 16 kHz
// synthesis by launches, latency-bound); the stream waits for the envelope right in front of the kernel of the unvoiced
// pulses and for the aperiodicity, which only a voiced pulse reads, in front of the kernel of the voiced ones, so a
// caller can produce the two arrays (mgc2sp, decode_aperiodicity) on another stream meanwhile.
static int world_synthesize_impl(const double* d_f0, const double* d_sp, const double* d_ap,
                                 const int64_t* h_f_off, const int64_t* h_y_off, int n_utts, int fs,
                                 double frame_period_ms, int fft_size, double preemphasis,
                                 float* d_y_f32, double* d_y_f64, void* stream, hipEvent_t spectra_ready,
                                 hipEvent_t ap_ready) {
  ITTS_REQUIRE(h_f_off && h_y_off && (n_utts == 0 || (d_f0 && d_sp && d_ap && (d_y_f32 || d_y_f64))), "null pointer");
  ITTS_REQUIRE(n_utts >= 0 && fs > 0 && frame_period_ms > 0, "bad sizes");
  ITTS_REQUIRE(fft_size >= 256 && fft_size <= 4096 && (fft_size & (fft_size - 1)) == 0,
               "fft_size must be 2^k in [256, 4096]");
  if (n_utts == 0) return ITTS_OK;
  hipStream_t s = as_stream(stream);
  itts::ScratchScope scratch_scope(s);
  DeviceContext* ctx = get_context();
  if (!ctx) return ITTS_E_HIP;
  const JumpTable* jt = get_jump_table(ctx);
  if (!jt) {
    set_error("could not create the RNG jump table");
    return ITTS_E_HIP;
  }
  SynParams p{};
  p.fs = fs; p.fft = fft_size; p.logfft = 0;
  while ((1 << p.logfft) < fft_size) ++p.logfft;
  p.fp = frame_period_ms / 1000.0;
  p.lowest_f0 = (double)(fs / fft_size) + 1.0;  // integer division as in WORLD
  p.n_utts = n_utts;
  std::vector<SynUtt> utts(n_utts);
  int64_t s_n = 0, b_n = 0;
  int max_nblk = 0, max_yl = 0;
  for (int u = 0; u < n_utts; ++u) {
    const int64_t T = h_f_off[u + 1] - h_f_off[u];
    const int64_t yl = itts_world_synth_length(T, fs, frame_period_ms);
    ITTS_REQUIRE(T >= 2 && yl >= 2 && yl < ((int64_t)1 << 30), "each utterance needs >= 2 frames");
    ITTS_REQUIRE(h_y_off[u + 1] - h_y_off[u] == yl, "output offsets do not match int(T*frame_period*fs/1000)");
    ITTS_REQUIRE(yl < (int64_t)RCHUNK << NJUMP, "utterance too long for the RNG jump table");
    SynUtt& d = utts[u];
    d.f_off = h_f_off[u]; d.T = (int)T; d.y_off = h_y_off[u]; d.yl = (int)yl;
    d.s_off = s_n; d.b_off = b_n; d.nblk = (int)((yl + CHUNK - 1) / CHUNK);
    s_n += yl + 8; b_n += d.nblk;
    max_nblk = std::max(max_nblk, d.nblk);
    max_yl = std::max(max_yl, d.yl);
  }
  const int64_t y_total = h_y_off[n_utts];
  SynUtt* d_utts = nullptr;
  double *d_wrap = nullptr, *d_R = nullptr, *d_pc = nullptr, *d_ptot = nullptr, *d_y = nullptr;
  uint8_t* d_vuv = nullptr;
  int* d_pidx = nullptr;
  int64_t* d_gpoff = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_utts, n_utts * sizeof(SynUtt), s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_wrap, s_n * 8, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_R, s_n * 8, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_vuv, s_n, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_pidx, s_n * 4, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_pc, b_n * 8, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_ptot, n_utts * 8, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_gpoff, (n_utts + 1) * 8, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_y, y_total * 8, s));
  if (int rc = itts::staged_upload(d_utts, utts.data(), n_utts * sizeof(SynUtt), s)) return rc;

  const dim3 gblk(max_nblk, n_utts);
  hipLaunchKernelGGL(syn_inc_kernel, gblk, dim3(NT), 0, s, d_f0, d_utts, p, d_wrap, d_vuv, d_y);
  ITTS_LAUNCH_CHECK();
  const bool seq_phase = getenv("ITTS_SYNTH_SEQ_PHASE") != nullptr;   // A/B switch for the tests
  if (seq_phase) {
    // the checker: sequential phase, then the pulses in three passes of their own
    hipLaunchKernelGGL(syn_phase_seq_kernel, dim3(n_utts), dim3(64), 0, s, d_utts, d_wrap);
    ITTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(syn_pulse_count_kernel, gblk, dim3(NT), 0, s, d_utts, d_wrap, d_pc);
    ITTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(syn_scan_blocks_kernel, dim3(n_utts), dim3(64), 0, s, d_utts, d_pc, d_ptot);
    ITTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(syn_pulse_emit_kernel, gblk, dim3(NT), 0, s, d_utts, d_wrap, d_pc, d_pidx, p);
  } else {
    // the phase scan leaves the pulse lists and their lengths as it goes
    hipLaunchKernelGGL(syn_phase_scan_kernel, dim3(n_utts), dim3(PST), 0, s, d_utts, d_wrap, d_pidx, d_ptot);
  }
  ITTS_LAUNCH_CHECK();
  const int h = fft_size / 2;
  double* d_dcr = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_dcr, (size_t)(h + 1) * 8, s));
  // the pulses by kind (a pulse kernel per kind: the unvoiced ones without what only a periodic response needs)
  int *d_kq = nullptr, *d_kcnt = nullptr;
  int64_t* d_kgp = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_kq, s_n * 4, s));
  // (+ the two pulse kernels' counters; a multiple of 256 bytes: one fill instead of two)
  const size_t kcnt_bytes = (((size_t)(2 * n_utts + 2) * 4 + 255) / 256) * 256;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_kcnt, kcnt_bytes, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_kgp, (size_t)2 * (n_utts + 1) * 8, s));
  hipLaunchKernelGGL(syn_pulse_offsets_kernel, dim3(2), dim3(NT), 0, s, d_ptot, n_utts, d_gpoff, d_kcnt,
                     (int)(kcnt_bytes / 4), fft_size, d_dcr);
  ITTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(syn_pulse_split_kernel, dim3(8, n_utts), dim3(NT), 0, s, d_utts, d_ptot, d_pidx, d_vuv, n_utts,
                     d_kq, d_kcnt);
  ITTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(syn_kind_offsets_kernel, dim3(1), dim3(128), 0, s, d_kcnt, n_utts, d_kgp);
  ITTS_LAUNCH_CHECK();
  const bool wave_kernels = fft_size == 2 * wf::WF_N;
  int64_t* h_totals[2] = {nullptr, nullptr};
  hipEvent_t ev_total = nullptr;
  if (!wave_kernels) {
    // one workgroup per pulse: the host needs the two counts (copied to page-locked slots and awaited by polling
    // further down; the noise generator queued behind the copies keeps the GPU busy meanwhile)
    ITTS_HIP_CHECK(hipEventCreateWithFlags(&ev_total, hipEventDisableTiming));
    for (int kind = 0; kind < 2; ++kind) {
      h_totals[kind] = pinned_slot(ctx);
      ITTS_HIP_CHECK(hipMemcpyAsync(h_totals[kind], d_kgp + kind * (n_utts + 1) + n_utts, sizeof(int64_t),
                                    hipMemcpyDeviceToHost, s));
    }
    ITTS_HIP_CHECK(hipEventRecord(ev_total, s));
  }
  PulseArgs a{d_f0, d_sp, d_ap, d_utts, d_gpoff, d_ptot, d_pidx, d_wrap, d_vuv, d_R, d_y, p,
              ctx->tw_compact[p.logfft], d_dcr, d_kq, d_kgp, d_kcnt + 2 * n_utts};
  {
    const int nchunks = (max_yl + RCHUNK - 1) / RCHUNK;
    hipLaunchKernelGGL(syn_randn_kernel, dim3((nchunks + NT - 1) / NT, n_utts), dim3(NT), 0, s, d_utts, jt, d_R);
    ITTS_LAUNCH_CHECK();
  }
  if (wave_kernels) {
    // 16 .. 24 kHz: one wave per pulse, persistent waves (three or two workgroups of four waves per CU); the
    // pulse counts stay on the device
    int dev = 0, n_cu = 256;
    ITTS_HIP_CHECK(hipGetDevice(&dev));
    ITTS_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    constexpr size_t lds_v = syn_wave_lds_bytes(false), lds_u = syn_wave_lds_bytes(true);
    static_assert(SYN_WAVE_OCC * lds_v <= 160 * 1024 && 3 * lds_u <= 160 * 1024, "LDS budget exceeded");
    {
      // more than 64 KB of dynamic LDS: the attribute once per device (runtimes that enforce it fail the launch otherwise)
      static std::atomic<uint64_t> attr_done{0};
      if (dev >= 64 || !((attr_done.load() >> dev) & 1)) {
        ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)syn_pulse_wave_kernel<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v));
        ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)syn_pulse_wave_kernel<true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_u));
        if (dev < 64) attr_done.fetch_or(uint64_t(1) << dev);
      }
    }
    // a kernel per kind: the unvoiced pulses at three waves per SIMD, the voiced ones at two
    // (the unvoiced pulses read the envelope only: the aperiodicity may still be on its way while their kernel runs)
    if (spectra_ready) ITTS_HIP_CHECK(hipStreamWaitEvent(s, spectra_ready, 0));
    hipLaunchKernelGGL(syn_pulse_wave_kernel<true>, dim3((unsigned)(3 * n_cu)), dim3(NT), lds_u, s, a);
    ITTS_LAUNCH_CHECK();
    if (ap_ready) ITTS_HIP_CHECK(hipStreamWaitEvent(s, ap_ready, 0));
    a.kgpoff = d_kgp + (n_utts + 1);
    a.next = d_kcnt + 2 * n_utts + 1;
    hipLaunchKernelGGL(syn_pulse_wave_kernel<false>, dim3((unsigned)(SYN_WAVE_OCC * n_cu)), dim3(NT), lds_v, s, a);
    ITTS_LAUNCH_CHECK();
  } else {
    // other transform sizes: one workgroup per pulse
    {
      hipError_t e;
      while ((e = hipEventQuery(ev_total)) == hipErrorNotReady) {
      }
      (void)hipEventDestroy(ev_total);
      ITTS_HIP_CHECK(e);
    }
    const int64_t n_unv = *h_totals[0], n_voi = *h_totals[1];
    ITTS_REQUIRE(n_unv >= 0 && n_voi >= 0 && n_unv + n_voi <= y_total, "corrupt pulse count");
    if (spectra_ready) ITTS_HIP_CHECK(hipStreamWaitEvent(s, spectra_ready, 0));
    const bool sized = 2 * h == 2048 && !getenv("ITTS_SYN_GENERIC");
    for (int kind = 0; kind < 2; ++kind) {
      const int64_t n_pulses = kind == 0 ? n_unv : n_voi;
      if (n_pulses == 0) continue;
      const bool unv = kind == 0;
      const size_t lds = syn_pulse_lds_bytes(h, unv);
      ITTS_REQUIRE(lds <= 160 * 1024, "LDS budget exceeded");
      const void* kern = sized ? (unv ? (const void*)syn_pulse_kernel<11, true> : (const void*)syn_pulse_kernel<11, false>)
                               : (unv ? (const void*)syn_pulse_kernel<0, true> : (const void*)syn_pulse_kernel<0, false>);
      ITTS_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      a.kgpoff = d_kgp + kind * (n_utts + 1);
      if (!unv && ap_ready) ITTS_HIP_CHECK(hipStreamWaitEvent(s, ap_ready, 0));
      if (sized && unv) hipLaunchKernelGGL((syn_pulse_kernel<11, true>), dim3((unsigned)n_pulses), dim3(NT), lds, s, a);
      else if (sized) hipLaunchKernelGGL((syn_pulse_kernel<11, false>), dim3((unsigned)n_pulses), dim3(NT), lds, s, a);
      else if (unv) hipLaunchKernelGGL((syn_pulse_kernel<0, true>), dim3((unsigned)n_pulses), dim3(NT), lds, s, a);
      else hipLaunchKernelGGL((syn_pulse_kernel<0, false>), dim3((unsigned)n_pulses), dim3(NT), lds, s, a);
      ITTS_LAUNCH_CHECK();
    }
  }
  ITTS_HIP_CHECK(itts::scratch_free(d_dcr, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_kq, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_kcnt, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_kgp, s));
  if (preemphasis == 0.0) {
    hipLaunchKernelGGL(syn_cast_kernel, dim3((unsigned)std::min<int64_t>((y_total + 255) / 256, 8192)), dim3(256),
                       0, s, d_y, y_total, d_y_f32, d_y_f64);
  } else {
    // pre^warm <= 2^-64; |pre| >= 1 (not a de-emphasis filter) or a very slow decay: one thread per
    // utterance from its first sample
    const double ap = std::fabs(preemphasis);
    int64_t warm = ap < 1.0 ? (int64_t)std::ceil(-64.0 * std::log(2.0) / std::log(ap)) : max_yl;
    int64_t seg = std::max<int64_t>(2048, warm);
    if (warm >= max_yl) { warm = max_yl; seg = max_yl; }
    const int64_t nseg = (max_yl + seg - 1) / seg;
    hipLaunchKernelGGL(syn_deemph_kernel, dim3((unsigned)((nseg + 63) / 64), n_utts), dim3(64), 0, s, d_y,
                       d_utts, preemphasis, (int)seg, (int)warm, d_y_f32, d_y_f64);
  }
  ITTS_LAUNCH_CHECK();
  ITTS_HIP_CHECK(itts::scratch_free(d_utts, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_wrap, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_R, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_vuv, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_pidx, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_pc, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_ptot, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_gpoff, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_y, s));
  return ITTS_OK;
}

extern "C" int itts_world_synthesize(const double* d_f0, const double* d_sp, const double* d_ap,
                                     const int64_t* h_f_off, const int64_t* h_y_off, int n_utts, int fs,
                                     double frame_period_ms, int fft_size, double preemphasis,
                                     float* d_y_f32, double* d_y_f64, void* stream) {
  return world_synthesize_impl(d_f0, d_sp, d_ap, h_f_off, h_y_off, n_utts, fs, frame_period_ms, fft_size,
                               preemphasis, d_y_f32, d_y_f64, stream, nullptr, nullptr);
}

extern "C" int itts_world_synthesize_after(const double* d_f0, const double* d_sp, const double* d_ap,
                                           const int64_t* h_f_off, const int64_t* h_y_off, int n_utts, int fs,
                                           double frame_period_ms, int fft_size, double preemphasis,
                                           float* d_y_f32, double* d_y_f64, void* stream, void* envelope_ready_event,
                                           void* aperiodicity_ready_event) {
  return world_synthesize_impl(d_f0, d_sp, d_ap, h_f_off, h_y_off, n_utts, fs, frame_period_ms, fft_size,
                               preemphasis, d_y_f32, d_y_f64, stream, reinterpret_cast<hipEvent_t>(envelope_ready_event),
                               reinterpret_cast<hipEvent_t>(aperiodicity_ready_event));
}
