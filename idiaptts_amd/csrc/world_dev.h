// Workgroup-cooperative fp64 building blocks for the WORLD / SPTK frame kernels (gfx950).
// One 256-thread workgroup (4 wavefronts) owns one analysis frame or one synthesis pulse; all
// intermediates of the frame (windowed segment, spectra, cepstra, smoothing prefix sums) live
// in LDS, so HBM only sees the waveform samples read and the feature row written.
//
//   fft_lds      in-place DIT complex FFT (two radix-2 stages per pass) on interleaved (re,im) doubles in LDS,
//                twiddles from an LDS copy of the table (ds_read_b128 per operand)
//   rfft/irfft   real transforms of length n through a complex FFT of length n/2
//   block_scan   inclusive prefix sum (WORLD's cumulative spectra)
//   dc_correction / linear_smoothing / interp1Q: WORLD common.cpp + matlabfunctions.cpp
#pragma once
#include "common.h"
#include "fastmath.h"

namespace itts {
namespace wd {

constexpr int NT = 256;       // threads per frame workgroup
constexpr int TW_N = 16384;   // master twiddle table: tw[k] = exp(+2 pi i k / TW_N), k < TW_N/2
constexpr double kEps = 1e-12;  // WORLD kMySafeGuardMinimum
constexpr double kPi = 3.1415926535897932384626433832795;

// Thread index as an opaque value.  The frame kernels are long sequences of short
// `for (i = thread; i < n; i += NT)` loops over LDS arrays; with the plain built-in the compiler
// computes every thread-dependent address once, up front, and keeps it alive through the whole
// kernel (and hoists it out of a persistent workgroup loop): 80+ VGPRs of loop-invariant integers
// that halve the occupancy.  Recomputing a few integer operations per loop is far cheaper.
__device__ __forceinline__ int tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// log / sincos / cos of the per-bin loops: the short forms of fastmath.h.  log_pos is total (zero,
// negative, subnormal, infinite and NaN arguments give what log() gives) through a few selects; the
// trigonometric ones hand arguments beyond 1e5 radians to the library in ONE out-of-line copy per
// function (inlined at every call site the never-taken slow paths were 45 % of the pulse kernel's
// code, which no longer fitted the instruction cache).
__device__ __forceinline__ double log_pos(double x) {
  const bool sub = x < 2.2250738585072014e-308;                    // subnormal (or <= 0: fixed below)
  const double r0 = fm::flog(sub ? x * 18014398509481984.0 : x);   // 2^54
  double r = sub ? r0 - 37.429947750237048 : r0;                   // 54 ln 2
  r = x == 0.0 ? -__builtin_huge_val() : r;
  r = x < 0.0 ? __builtin_nan("") : r;
  r = x == __builtin_huge_val() ? x : r;
  return r;
}
__device__ __attribute__((noinline)) double2 sincos_far(double x) {
  double s, c;
  sincos(x, &s, &c);
  return make_double2(s, c);
}
__device__ __forceinline__ void sincos_mid(double x, double* sn, double* cs) {
  if (__builtin_expect(!(fabs(x) <= 1e5), 0)) {
    const double2 r = sincos_far(x);
    *sn = r.x;
    *cs = r.y;
    return;
  }
  fm::fsincos(x, sn, cs);
}
__device__ __forceinline__ double cos_mid(double x) {
  if (__builtin_expect(!(fabs(x) <= 1e5), 0)) return sincos_far(x).y;
  return fm::fcos(x);
}
__device__ __forceinline__ double sin_mid(double x) {
  if (__builtin_expect(!(fabs(x) <= 1e5), 0)) return sincos_far(x).x;
  return fm::fsin(x);
}

// Lane shuffles whose addresses are computed where they are used, from the caller's lane number:
// __shfl_xor / __shfl_up derive theirs from the hardware lane id, which the compiler hoists out of a
// persistent kernel's loop -- a dozen registers held (and spilled) for the whole kernel.
__device__ __forceinline__ double lane_shfl(double v, int src_lane) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_at(double v, int lane) {   // same butterfly as wave_sum: same bits
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += lane_shfl(v, lane ^ off);
  return v;
}

// Utterance of global frame g (off[u] <= g < off[u + 1]) for a WHOLE WAVE asking about the same g (all 64 lanes
// active): the offsets are compared 64 at a time -- loads that do not depend on each other -- instead of by a
// binary search, whose eight dependent trips to the cache were 4 us at the start of every frame's life (a third
// of StoneMask's time, an eighth of a CheapTrick / D4C frame).  Beyond 1 024 utterances: the binary search.
__device__ __forceinline__ int find_utt_wave(const int64_t* __restrict__ off, int n_utts, int64_t g) {
  if (n_utts > 1024) {
    int lo = 0, hi = n_utts;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
  }
  const int lane = (int)(threadIdx.x & 63);
  int below = 0;                                        // offsets off[0 .. n_utts) that are <= g (off[0] = 0 always is)
  for (int i0 = 0; i0 < n_utts; i0 += 64) {
    const int i = i0 + lane;
    const bool le = i < n_utts && off[i] <= g;
    below += __popcll(__ballot(le));
  }
  return below - 1;
}

__device__ __forceinline__ int mround(double x) { return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5); }
__device__ __forceinline__ int ilog2(int n) { return 31 - __clz(n); }

// Copies the twiddles of an n-point transform into LDS: tw[k] = exp(+2 pi i k / n), k < n/2.
__device__ __forceinline__ void load_twiddles(double2* tw, const double2* __restrict__ g_tw, int n) {
  const int stride = TW_N / n;
  for (int k = tid(); k < n / 2; k += NT) tw[k] = g_tw[k * stride];
}

// LDS position of logical element i while a transform is in flight.  The butterflies of the first
// two radix-4 passes touch 16-byte elements at lane strides of 64 and 256 bytes and the bit
// reversal scatters at n/16: on the natural layout those are 4- to 16-way bank conflicts
// (ds_read_b128: 16 lanes share 16 slots of 16 bytes; ds_write_b128: 8 lanes share 8).  XOR-ing the
// slot inside each 256-byte block with bits 4-7 and the reversed bits 6-9 of the index makes every
// pass conflict-free on reads and near the store-path limit on writes (simulated with the lane
// groups of MI355X_MICROARCH.md: 1024 points 3387 -> 1792 LDS cycles); only the four low index bits
// change, so a wave's butterflies of the LAST pass read and write the same 16-element blocks and
// can write back in natural order in place.
__device__ __forceinline__ int fft_phys(int i) {
  return i ^ (((i >> 4) ^ (int)(__brev((unsigned)(i >> 6)) >> 28)) & 15);
}

constexpr int FFT_PAIR_MAX = 4 * NT;  // largest transform whose first pass is one butterfly per thread (fft_lds_pair)
constexpr int FFT_SWZ_MAX = 8 * NT;   // .. two per thread: the swizzled layout up to 2 048 complex points (the 4 096-point real transforms of D4C at 48 kHz)

// In-place complex FFT of z[0..n) (n = 2^logn <= tw_n). tw holds exp(+2 pi i k / tw_n).
// sign = -1: forward (e^{-i..}), +1: unnormalised inverse. Ends with a barrier.  Input and output
// are in natural order; in between the elements live at fft_phys(i) (64 <= n <= FFT_SWZ_MAX).
// CLOGN / CTSHIFT > 0: n = 2^CLOGN and log2(tw_n) - 1 = CTSHIFT are compile-time constants (the passes unroll, their
// shifts, masks and swizzle constants fold: about a third of a pass's integer instructions go) -- same butterflies,
// same twiddle entries, same order: same bits.
template <bool SWZ = true, int CLOGN = 0, int CTSHIFT = 0>
__device__ inline void fft_lds(double2* z, int n_rt, int logn_rt, const double2* tw, int tw_n, int sign) {
  const int n = CLOGN ? (1 << CLOGN) : n_rt, logn = CLOGN ? CLOGN : logn_rt;
  const bool swz = SWZ && n >= 64 && n <= FFT_SWZ_MAX;
  if (!swz) {
    for (int i = tid(); i < n; i += NT) {
      const int j = (int)(__brev((unsigned)i) >> (32 - logn));
      if (i < j) {
        const double2 a = z[i], b = z[j];
        z[i] = b;
        z[j] = a;
      }
    }
    __syncthreads();
  }
  const int tshift = CTSHIFT ? CTSHIFT : ilog2(tw_n) - 1;  // twiddle index of w_{2h}^r in the tw_n table: r * tw_n/(2h)
  // complex product with the (possibly conjugated) twiddle, the one expression every stage uses
  auto twmul = [&](const double2 v, const double2 w) {
    const double wi = sign < 0 ? -w.y : w.y;
    return make_double2(v.x * w.x - v.y * wi, v.x * wi + v.y * w.x);
  };
  auto at = [&](int i) { return swz ? fft_phys(i) : i; };
  int s = 1;
  if (swz) {
    // First pass with the bit reversal folded in: butterfly t of the bit-reversed array takes the
    // natural elements u + {0, n/2, n/4, 3n/4}, u = bitrev(t) -- so thread u reads four
    // consecutive-lane (conflict-free) addresses and, after everyone has read, stores the results
    // at the swizzled positions of 4 t + {0, 1, 2, 3}.  Same butterflies and twiddle entries as the
    // generic pass below with h = 1.
    // (one butterfly per thread up to 4 NT points, two from there to 8 NT: everything is read before anything is
    // written)
    constexpr int FIRST = FFT_SWZ_MAX / (4 * NT);
    double2 f0[FIRST], f1[FIRST], f2[FIRST], f3[FIRST];
#pragma unroll
    for (int rep = 0; rep < FIRST; ++rep) {
      const int u = tid() + rep * NT;
      if (u < n / 4) {
        f0[rep] = z[u];
        f1[rep] = z[u + n / 2];
        f2[rep] = z[u + n / 4];
        f3[rep] = z[u + n / 2 + n / 4];
      }
    }
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < FIRST; ++rep) {
      const int u = tid() + rep * NT;
      if (u < n / 4) {
        const double2 z0 = f0[rep], z1 = f1[rep], z2 = f2[rep], z3 = f3[rep];
        const int a0 = 4 * (int)(__brev((unsigned)u) >> (32 - (logn - 2)));
        const double2 w1 = tw[0];
        const double2 w2 = tw[0];
        const double2 w3 = tw[1 << (tshift - 1)];
        const double2 x1 = twmul(z1, w1), x3 = twmul(z3, w1);
        const double2 y0 = make_double2(z0.x + x1.x, z0.y + x1.y), y1 = make_double2(z0.x - x1.x, z0.y - x1.y);
        const double2 y2 = make_double2(z2.x + x3.x, z2.y + x3.y), y3 = make_double2(z2.x - x3.x, z2.y - x3.y);
        const double2 u2 = twmul(y2, w2), u3 = twmul(y3, w3);
        const int p0 = fft_phys(a0);  // the swizzle is linear over GF(2) and leaves 1, 2, 3 alone
        z[p0] = make_double2(y0.x + u2.x, y0.y + u2.y);
        z[p0 ^ 2] = make_double2(y0.x - u2.x, y0.y - u2.y);
        z[p0 ^ 1] = make_double2(y1.x + u3.x, y1.y + u3.y);
        z[p0 ^ 3] = make_double2(y1.x - u3.x, y1.y - u3.y);
      }
    }
    __syncthreads();
    s = 3;
  }
  // Two radix-2 stages per pass: a thread takes the four elements a0 + {0, h, 2h, 3h} through
  // stage s (pairs at distance h) and stage s+1 (pairs at distance 2h) in registers.  Same
  // butterflies, same twiddle-table entries, same operation order as two separate radix-2
  // passes -- the results are bit-identical -- with 11 instead of 20 LDS accesses per four
  // elements and half the barriers (these kernels are bound by LDS traffic).
#pragma unroll
  for (; s + 1 <= logn; s += 2) {
    const int h = 1 << (s - 1);
    const bool last = s + 1 == logn;   // the final pass stores in natural order
    // a0 has bits s-1 and s clear, so a0 + m h = a0 ^ m h and (the swizzle being linear over GF(2))
    // its position is the position of a0 XOR a per-pass constant
    const int c1 = swz ? fft_phys(h) : h, c2 = swz ? fft_phys(2 * h) : 2 * h;
    for (int t = tid(); t < n / 4; t += NT) {
      const int r = t & (h - 1);
      const int a0 = ((t >> (s - 1)) << (s + 1)) + r;
      const int a1 = a0 + h, a2 = a0 + 2 * h, a3 = a0 + 3 * h;
      const double2 w1 = tw[r << (tshift - (s - 1))];
      const double2 w2 = tw[r << (tshift - s)];
      const double2 w3 = tw[(r + h) << (tshift - s)];
      const int p0 = at(a0), p1 = p0 ^ c1, p2 = p0 ^ c2, p3 = p1 ^ c2;
      const double2 z0 = z[p0], z1 = z[p1], z2 = z[p2], z3 = z[p3];
      const double2 x1 = twmul(z1, w1), x3 = twmul(z3, w1);
      const double2 y0 = make_double2(z0.x + x1.x, z0.y + x1.y), y1 = make_double2(z0.x - x1.x, z0.y - x1.y);
      const double2 y2 = make_double2(z2.x + x3.x, z2.y + x3.y), y3 = make_double2(z2.x - x3.x, z2.y - x3.y);
      const double2 u2 = twmul(y2, w2), u3 = twmul(y3, w3);
      const int o0 = last ? a0 : p0, o1 = last ? a1 : p1, o2 = last ? a2 : p2, o3 = last ? a3 : p3;
      z[o0] = make_double2(y0.x + u2.x, y0.y + u2.y);
      z[o2] = make_double2(y0.x - u2.x, y0.y - u2.y);
      z[o1] = make_double2(y1.x + u3.x, y1.y + u3.y);
      z[o3] = make_double2(y1.x - u3.x, y1.y - u3.y);
    }
    __syncthreads();
  }
  for (; s <= logn; ++s) {   // odd log2(n): one plain radix-2 stage is left (always the last pass)
    const int h = 1 << (s - 1);
    for (int t = tid(); t < n / 2; t += NT) {
      const int r = t & (h - 1);
      const int a = ((t >> (s - 1)) << s) + r;
      const int b = a + h;
      const int pa = at(a), pb = swz ? pa ^ fft_phys(h) : b;
      const double2 x = twmul(z[pb], tw[r << (tshift - (s - 1))]);
      const double2 za = z[pa];
      z[b] = make_double2(za.x - x.x, za.y - x.y);
      z[a] = make_double2(za.x + x.x, za.y + x.y);
    }
    __syncthreads();
  }
}

// Real FFT: z viewed as n real samples (z[k] = (x[2k], x[2k+1])), needs n/2+1 complex slots.
// On return z[k] = X[k], k = 0..n/2 (numpy.fft.rfft).
template <bool SWZ = true, int CLOGN = 0, int CTSHIFT = 0>     // (CLOGN: log2 of the REAL transform's size)
__device__ inline void rfft_lds(double2* z, int n_rt, int logn, const double2* tw, int tw_n_rt) {
  const int n = CLOGN ? (1 << CLOGN) : n_rt, tw_n = CTSHIFT ? (2 << CTSHIFT) : tw_n_rt;
  const int h = n / 2;
  fft_lds<SWZ, CLOGN ? CLOGN - 1 : 0, CTSHIFT>(z, h, logn - 1, tw, tw_n, -1);
  const int tstride = tw_n / n;
  for (int k = tid(); k <= h / 2; k += NT) {
    if (k == 0) {
      const double2 z0 = z[0];
      z[0] = make_double2(z0.x + z0.y, 0.0);
      z[h] = make_double2(z0.x - z0.y, 0.0);
    } else {
      const int j = h - k;
      const double2 zk = z[k], zj = z[j];
      const double er = 0.5 * (zk.x + zj.x), ei = 0.5 * (zk.y - zj.y);
      const double dr = 0.5 * (zk.x - zj.x), di = 0.5 * (zk.y + zj.y);
      const double orr = di, oi = -dr;  // O = -i D
      const double2 w = tw[k * tstride];
      const double wr = w.x, wi = -w.y;  // w^k = e^{-2 pi i k / n}
      const double tr = orr * wr - oi * wi, ti = orr * wi + oi * wr;
      z[k] = make_double2(er + tr, ei + ti);
      z[j] = make_double2(er - tr, -(ei - ti));
    }
  }
  __syncthreads();
}

// TWO transforms of the same size in lockstep (arrays z0, z1): one set of barriers, index arithmetic and
// twiddle loads serves both -- the workgroup transforms are bound by exactly those (61 %
// of the instructions).  Per array the butterflies, twiddle entries and order of operations are those of
// fft_lds: the results are bit-identical.  64 <= n <= FFT_PAIR_MAX (the swizzled path only).
template <int CLOGN = 0, int CTSHIFT = 0>
__device__ inline void fft_lds_pair(double2* z0, double2* z1, int n_rt, int logn_rt, const double2* tw, int tw_n, int sign) {
  const int n = CLOGN ? (1 << CLOGN) : n_rt, logn = CLOGN ? CLOGN : logn_rt;
  double2* zz[2] = {z0, z1};
  const int tshift = CTSHIFT ? CTSHIFT : ilog2(tw_n) - 1;
  auto twmul = [&](const double2 v, const double2 w) {
    const double wi = sign < 0 ? -w.y : w.y;
    return make_double2(v.x * w.x - v.y * wi, v.x * wi + v.y * w.x);
  };
  {
    const int u = tid();
    const bool on = u < n / 4;
    double2 q0[2], q1[2], q2[2], q3[2];
    if (on) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        q0[a] = zz[a][u];
        q1[a] = zz[a][u + n / 2];
        q2[a] = zz[a][u + n / 4];
        q3[a] = zz[a][u + n / 2 + n / 4];
      }
    }
    __syncthreads();
    if (on) {
      const int a0 = 4 * (int)(__brev((unsigned)u) >> (32 - (logn - 2)));
      const double2 w1 = tw[0];
      const double2 w2 = tw[0];
      const double2 w3 = tw[1 << (tshift - 1)];
      const int p0 = fft_phys(a0);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const double2 x1 = twmul(q1[a], w1), x3 = twmul(q3[a], w1);
        const double2 y0 = make_double2(q0[a].x + x1.x, q0[a].y + x1.y), y1 = make_double2(q0[a].x - x1.x, q0[a].y - x1.y);
        const double2 y2 = make_double2(q2[a].x + x3.x, q2[a].y + x3.y), y3 = make_double2(q2[a].x - x3.x, q2[a].y - x3.y);
        const double2 u2 = twmul(y2, w2), u3 = twmul(y3, w3);
        zz[a][p0] = make_double2(y0.x + u2.x, y0.y + u2.y);
        zz[a][p0 ^ 2] = make_double2(y0.x - u2.x, y0.y - u2.y);
        zz[a][p0 ^ 1] = make_double2(y1.x + u3.x, y1.y + u3.y);
        zz[a][p0 ^ 3] = make_double2(y1.x - u3.x, y1.y - u3.y);
      }
    }
    __syncthreads();
  }
  int s = 3;
#pragma unroll
  for (; s + 1 <= logn; s += 2) {
    const int h = 1 << (s - 1);
    const bool last = s + 1 == logn;
    const int c1 = fft_phys(h), c2 = fft_phys(2 * h);
    for (int t = tid(); t < n / 4; t += NT) {
      const int r = t & (h - 1);
      const int a0 = ((t >> (s - 1)) << (s + 1)) + r;
      const int a1 = a0 + h, a2 = a0 + 2 * h, a3 = a0 + 3 * h;
      const double2 w1 = tw[r << (tshift - (s - 1))];
      const double2 w2 = tw[r << (tshift - s)];
      const double2 w3 = tw[(r + h) << (tshift - s)];
      const int p0 = fft_phys(a0), p1 = p0 ^ c1, p2 = p0 ^ c2, p3 = p1 ^ c2;
      const int o0 = last ? a0 : p0, o1 = last ? a1 : p1, o2 = last ? a2 : p2, o3 = last ? a3 : p3;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const double2 q0 = zz[a][p0], q1 = zz[a][p1], q2 = zz[a][p2], q3 = zz[a][p3];
        const double2 x1 = twmul(q1, w1), x3 = twmul(q3, w1);
        const double2 y0 = make_double2(q0.x + x1.x, q0.y + x1.y), y1 = make_double2(q0.x - x1.x, q0.y - x1.y);
        const double2 y2 = make_double2(q2.x + x3.x, q2.y + x3.y), y3 = make_double2(q2.x - x3.x, q2.y - x3.y);
        const double2 u2 = twmul(y2, w2), u3 = twmul(y3, w3);
        zz[a][o0] = make_double2(y0.x + u2.x, y0.y + u2.y);
        zz[a][o2] = make_double2(y0.x - u2.x, y0.y - u2.y);
        zz[a][o1] = make_double2(y1.x + u3.x, y1.y + u3.y);
        zz[a][o3] = make_double2(y1.x - u3.x, y1.y - u3.y);
      }
    }
    __syncthreads();
  }
  for (; s <= logn; ++s) {   // odd log2(n): one plain radix-2 stage is left (always the last pass)
    const int h = 1 << (s - 1);
    for (int t = tid(); t < n / 2; t += NT) {
      const int r = t & (h - 1);
      const int ia = ((t >> (s - 1)) << s) + r;
      const int ib = ia + h;
      const int pa = fft_phys(ia), pb = pa ^ fft_phys(h);
      const double2 w = tw[r << (tshift - (s - 1))];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const double2 x = twmul(zz[a][pb], w);
        const double2 za = zz[a][pa];
        zz[a][ib] = make_double2(za.x - x.x, za.y - x.y);
        zz[a][ia] = make_double2(za.x + x.x, za.y + x.y);
      }
    }
    __syncthreads();
  }
}

// Two real transforms in lockstep (see rfft_lds): z0 / z1 hold n real samples each on entry, X[0 .. n/2] on return.
template <int CLOGN = 0, int CTSHIFT = 0>
__device__ inline void rfft_lds_pair(double2* z0, double2* z1, int n_rt, int logn, const double2* tw, int tw_n_rt) {
  const int n = CLOGN ? (1 << CLOGN) : n_rt, tw_n = CTSHIFT ? (2 << CTSHIFT) : tw_n_rt;
  const int h = n / 2;
  fft_lds_pair<CLOGN ? CLOGN - 1 : 0, CTSHIFT>(z0, z1, h, logn - 1, tw, tw_n, -1);
  double2* zz[2] = {z0, z1};
  const int tstride = tw_n / n;
  for (int k = tid(); k <= h / 2; k += NT) {
    if (k == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const double2 q0 = zz[a][0];
        zz[a][0] = make_double2(q0.x + q0.y, 0.0);
        zz[a][h] = make_double2(q0.x - q0.y, 0.0);
      }
    } else {
      const int j = h - k;
      const double2 w = tw[k * tstride];
      const double wr = w.x, wi = -w.y;  // w^k = e^{-2 pi i k / n}
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const double2 zk = zz[a][k], zj = zz[a][j];
        const double er = 0.5 * (zk.x + zj.x), ei = 0.5 * (zk.y - zj.y);
        const double dr = 0.5 * (zk.x - zj.x), di = 0.5 * (zk.y + zj.y);
        const double orr = di, oi = -dr;  // O = -i D
        const double tr = orr * wr - oi * wi, ti = orr * wi + oi * wr;
        zz[a][k] = make_double2(er + tr, ei + ti);
        zz[a][j] = make_double2(er - tr, -(ei - ti));
      }
    }
  }
  __syncthreads();
}

// Inverse real FFT: z[k] = X[k], k = 0..n/2 (imag of X[0], X[n/2] ignored) -> z viewed as n real
// samples, normalised like numpy.fft.irfft.
template <bool SWZ = true, int CLOGN = 0, int CTSHIFT = 0>     // (CLOGN: log2 of the REAL transform's size, as rfft_lds)
__device__ inline void irfft_lds(double2* z, int n_rt, int logn, const double2* tw, int tw_n_rt) {
  const int n = CLOGN ? (1 << CLOGN) : n_rt, tw_n = CTSHIFT ? (2 << CTSHIFT) : tw_n_rt;
  const int h = n / 2;
  const int tstride = tw_n / n;
  for (int k = tid(); k <= h / 2; k += NT) {
    const int j = h - k;
    double2 xk = z[k], xj = z[j];
    if (k == 0) {
      xk.y = 0.0;
      xj.y = 0.0;
    }
    const double er = 0.5 * (xk.x + xj.x), ei = 0.5 * (xk.y - xj.y);
    const double dr = 0.5 * (xk.x - xj.x), di = 0.5 * (xk.y + xj.y);
    const double2 w = tw[k * tstride];  // conj(w^k) = e^{+2 pi i k / n}
    const double orr = dr * w.x - di * w.y, oi = dr * w.y + di * w.x;
    const double2 zk = make_double2(er - oi, ei + orr);
    const double2 zj = make_double2(er + oi, -ei + orr);
    if (k == 0) {
      z[0] = zk;
    } else {
      z[k] = zk;
      if (j != k) z[j] = zj;
    }
  }
  __syncthreads();
  fft_lds<SWZ, CLOGN ? CLOGN - 1 : 0, CTSHIFT>(z, h, logn - 1, tw, tw_n, +1);
  const double s = 1.0 / (double)h;
  for (int k = tid(); k < h; k += NT) {
    double2 v = z[k];
    v.x *= s;
    v.y *= s;
    z[k] = v;
  }
  __syncthreads();
}

// Block-wide sum; result identical in every thread. red: >= 8 doubles of LDS.
__device__ __forceinline__ double bsum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid() & 63) == 0) red[tid() >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// Two block-wide sums behind one pair of barriers (each with bsum's order of additions: same bits).
// red: >= 8 doubles of LDS.
__device__ __forceinline__ void bsum2(double& a, double& b, double* red) {
  a = wave_sum(a);
  b = wave_sum(b);
  __syncthreads();
  if ((tid() & 63) == 0) {
    red[tid() >> 6] = a;
    red[4 + (tid() >> 6)] = b;
  }
  __syncthreads();
  a = (red[0] + red[1]) + (red[2] + red[3]);
  b = (red[4] + red[5]) + (red[6] + red[7]);
}

// In-place inclusive prefix sum of a[0..n) in LDS. red: >= NT+8 doubles. Ends with a barrier.
__device__ inline void block_scan(double* a, int n, double* red) {
  const int chunk = (n + NT - 1) / NT;
  const int lo = tid() * chunk;
  const int hi = min(n, lo + chunk);
  double s = 0.0;
  for (int i = lo; i < hi; ++i) {
    s += a[i];
    a[i] = s;
  }
  red[tid()] = s;
  __syncthreads();
  if (tid() < 64) {  // one wave scans the 256 chunk totals (4 per lane)
    double v0 = red[4 * tid()], v1 = red[4 * tid() + 1], v2 = red[4 * tid() + 2],
           v3 = red[4 * tid() + 3];
    v1 += v0;
    v2 += v1;
    v3 += v2;
    double incl = v3;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double o = __shfl_up(incl, off, 64);
      if ((int)tid() >= off) incl += o;
    }
    const double excl = incl - v3;
    red[4 * tid()] = excl;            // exclusive prefix of chunk totals
    red[4 * tid() + 1] = excl + v0;
    red[4 * tid() + 2] = excl + v1;
    red[4 * tid() + 3] = excl + v2;
  }
  __syncthreads();
  const double base = red[tid()];
  for (int i = lo; i < hi; ++i) a[i] += base;
  __syncthreads();
}

// WORLD interp1Q at one point: equally spaced abscissa x0 + i*shift, ordinate y[0..ylen)
__device__ __forceinline__ double interp1q(double x0, double shift, const double* y, int ylen,
                                           double xi) {
  const double pos = (xi - x0) / shift;
  const int base = (int)pos;
  const double frac = pos - base;
  const double dy = (base + 1 < ylen) ? y[base + 1] - y[base] : 0.0;
  return y[base] + dy * frac;
}

// WORLD DCCorrection (common.cpp): P[0..upper-2] += mirrored replica around f0. P has >= upper+1
// valid entries. Needs upper <= NT (f0 <= ~1 kHz at the FFT sizes in use). Ends with a barrier.
__device__ inline void dc_correction(double* P, double f0, int fs, int fft) {
  const int upper = 2 + (int)(f0 * fft / fs);
  double rep = 0.0;
  const int i = tid();
  for (int base = 0; base < upper - 1; base += NT) {
    const int ii = base + i;
    if (ii < upper - 1) {
      const double lfa = (double)ii * fs / fft;
      rep = interp1q(f0, -(double)fs / fft, P, upper + 1, lfa);
    }
    __syncthreads();
    if (ii < upper - 1) P[ii] += rep;
    __syncthreads();
  }
}

// WORLD LinearSmoothing: out[k] = mean of P over [f_k - width/2, f_k + width/2], k = 0..fft/2.
// mir: scratch of fft/2 + 2*boundary + 1 doubles; out may alias P. red: NT+8 doubles.
__device__ inline void linear_smoothing(const double* P, double width, int fs, int fft, double* out,
                                        double* mir, double* red) {
  const int boundary = (int)(width * fft / fs) + 1;
  const int h = fft / 2;
  const int ml = h + boundary * 2 + 1;
  for (int i = tid(); i < ml; i += NT) {
    double v;
    if (i < boundary)
      v = P[boundary - i];
    else if (i < h + boundary)
      v = P[i - boundary];
    else
      v = P[h - (i - (h + boundary))];
    mir[i] = v * fs / fft;
  }
  __syncthreads();
  block_scan(mir, ml, red);
  const double org = -((double)boundary - 0.5) * fs / fft;
  const double dfi = (double)fs / fft;
  for (int k = tid(); k <= h; k += NT) {
    const double fa = (double)k / fft * fs - width / 2.0;
    const double lo = interp1q(org, dfi, mir, ml, fa);
    const double hi = interp1q(org, dfi, mir, ml, fa + width);
    out[k] = (hi - lo) / width;
  }
  __syncthreads();
}

}  // namespace wd
}  // namespace itts
