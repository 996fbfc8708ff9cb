// Wave-per-transform fp64 FFT for the WORLD frame kernels (gfx950): ONE wavefront owns a 512-point
// complex transform (= a 1024-point real transform), eight points per lane in registers, no
// workgroup barrier anywhere.
//
// Why (DESIGN.md sections 11b / 12): the workgroup-per-frame kernels are bound by VALU issue, 61 % of it
// integer address arithmetic, moves and the idle waves of the short transforms.  Here a transform is
// three register passes of three radix-2 stages each; between the passes the wave transposes through
// its own 8.5 KB of LDS with addresses of the form (per-lane base + immediate); the twiddles come
// as wave-uniform constants (pass 1) or from a per-lane table the workgroup builds once in LDS (pass
// 2: 7 entries that depend on lane & 7, pass 3: 7 entries per lane -- 8 KB per workgroup; in
// registers they cost 56 VGPRs for the lifetime of the wave, which the frame kernels do not have).
//
// Numerics: the butterflies, the twiddle-table entries and the order of operations are those of
// wd::fft_lds / rfft_lds / irfft_lds (radix-2 DIT on the bit-reversed array) -- results are
// bit-identical (scripts/wave_fft_sim.py proves the schedule equal as a computation graph and free
// of LDS bank conflicts; scripts/wave_fft_lab compares the bits on the GPU).
//
// Layout "A" (input and output of every transform here): element m = lane + 64 q in register q.
#pragma once
#include "common.h"

namespace itts {
namespace wf {

constexpr int WF_N = 512;                    // complex points per transform
constexpr int WF_PITCH = 68;                 // 16-byte slots per row of the first exchange (64 + 4: conflict-free reads)
constexpr int WF_LDS_BYTES = 8 * WF_PITCH * 16;   // 8704 B per wave
constexpr int WF_TABLE_BYTES = (7 * (8 + 64) + 264) * 16;   // 12288 B per workgroup: [7][8] pass 2, [7][64] pass 3, tw[0 .. 256]

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int bitrev3(int x) { return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1); }

// compiler-level ordering of a wave's own LDS traffic (the hardware executes a wave's LDS
// instructions in order; nothing is emitted)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Plan512 {
  const double2* t2;    // pass 2, a = lane & 7: t2[8 i], i = 0 .. 6 = tw[64 a]; tw[32 a], tw[32 a + 256]; tw[16 a + {0, 128, 256, 384}]
  const double2* t3;    // pass 3, M = lane:     t3[64 i]            = tw[8 M];  tw[4 M],  tw[4 M + 256];  tw[2 M + {0, 128, 256, 384}]
  double2 c128, c256, c384;   // pass 1 (wave-uniform): tw[128], tw[256], tw[384]  (tw[0] = 1 is not multiplied)
  const double2* tw_lo;       // (table) tw + lane:        post / pre-processing twiddles of registers 0 .. 3 (+ 64 q)
  const double2* tw_hi;       // (table) tw + (64 - lane): ... of registers 4 .. 7 (+ 64 (7 - q))
  double2* x1w;         // exchange 1, store base: slot lane             (+ bitrev3(q) * PITCH)
  double2* x1r;         // exchange 1, load base:  slot (lane & 7) * PITCH + (lane >> 3)   (+ 8 q)
  double2* x2w;         // exchange 2, store base: slot (lane & 7) + 64 bitrev3(lane >> 3) (+ 8 bitrev3(q))
  double2* x2r;         // exchange 2, load base / natural slot of element lane            (+ 64 q)
  double2* xpr;         // partner of (lane, q): slot (64 - lane) + 64 (7 - q)
};

// Builds the workgroup's twiddle table (WF_TABLE_BYTES at `table`); ends with a barrier.  tw: the
// compact table of the 1024-point real transform, tw[k] = exp(+2 pi i k / 1024), k < 512.
__device__ __forceinline__ void table512_init(void* table, const double2* __restrict__ tw) {
  double2* t = reinterpret_cast<double2*>(table);
  for (int i = threadIdx.x; i < 7 * 8; i += blockDim.x) {
    const int e = i >> 3, a = i & 7;
    const int idx = e == 0 ? 64 * a : (e < 3 ? 32 * a + 256 * (e - 1) : 16 * a + 128 * (e - 3));
    t[i] = tw[idx];
  }
  for (int i = threadIdx.x; i < 7 * 64; i += blockDim.x) {
    const int e = i >> 6, m = i & 63;
    const int idx = e == 0 ? 8 * m : (e < 3 ? 4 * m + 256 * (e - 1) : 2 * m + 128 * (e - 3));
    t[56 + i] = tw[idx];
  }
  for (int i = threadIdx.x; i <= 256; i += blockDim.x) t[504 + i] = tw[i];
  __syncthreads();
}

// lds: this wave's WF_LDS_BYTES; table: the workgroup's table (table512_init).
__device__ __forceinline__ void plan512_init(Plan512& p, const double2* __restrict__ tw, void* lds, const void* table) {
  const int l = lane_id(), a = l & 7;
  p.t2 = reinterpret_cast<const double2*>(table) + a;
  p.t3 = reinterpret_cast<const double2*>(table) + 56 + l;
  p.c128 = tw[128];
  p.c256 = tw[256];
  p.c384 = tw[384];
  p.tw_lo = reinterpret_cast<const double2*>(table) + 504 + l;
  p.tw_hi = reinterpret_cast<const double2*>(table) + 504 + (64 - l);
  double2* s = reinterpret_cast<double2*>(lds);
  p.x1w = s + l;
  p.x1r = s + a * WF_PITCH + (l >> 3);
  p.x2w = s + a + 64 * bitrev3(l >> 3);
  p.x2r = s + l;
  p.xpr = s + (64 - l);
}

// the one complex product every stage uses (same expression as wd::fft_lds: same contraction)
// (the sign of the transform is a run-time +-1.0 so that forward and inverse transforms share one copy
// of the code: w.y * -1.0 is exactly -w.y)
__device__ __forceinline__ double2 twmul(const double2 v, const double2 w, const double sgn) {
  const double wi = w.y * sgn;
  return make_double2(v.x * w.x - v.y * wi, v.x * wi + v.y * w.x);
}
__device__ __forceinline__ void bfly(double2& a, double2& b, const double2 w, const double sgn) {
  const double2 x = twmul(b, w, sgn);
  const double2 a0 = a;
  a = make_double2(a0.x + x.x, a0.y + x.y);
  b = make_double2(a0.x - x.x, a0.y - x.y);
}
// butterfly with the twiddle tw[0] = (1, 0): the product is the operand itself
__device__ __forceinline__ void bfly1(double2& a, double2& b) {
  const double2 a0 = a, x = b;
  a = make_double2(a0.x + x.x, a0.y + x.y);
  b = make_double2(a0.x - x.x, a0.y - x.y);
}

// In-place 512-point complex FFT of layout A.  sgn = -1.0: forward, +1.0: unnormalised inverse.
__device__ __forceinline__ void cfft512(double2 (&z)[8], const Plan512& p, const double sgn) {
  // pass 1 (stages 1 .. 3): register q holds local position bitrev3(q) of eight consecutive points
  bfly1(z[0], z[4]); bfly1(z[2], z[6]); bfly1(z[1], z[5]); bfly1(z[3], z[7]);
  bfly1(z[0], z[2]); bfly(z[4], z[6], p.c256, sgn); bfly1(z[1], z[3]); bfly(z[5], z[7], p.c256, sgn);
  bfly1(z[0], z[1]); bfly(z[4], z[5], p.c128, sgn); bfly(z[2], z[3], p.c256, sgn); bfly(z[6], z[7], p.c384, sgn);
  // exchange 1
  p.x1w[0 * WF_PITCH] = z[0]; p.x1w[4 * WF_PITCH] = z[1]; p.x1w[2 * WF_PITCH] = z[2]; p.x1w[6 * WF_PITCH] = z[3];
  p.x1w[1 * WF_PITCH] = z[4]; p.x1w[5 * WF_PITCH] = z[5]; p.x1w[3 * WF_PITCH] = z[6]; p.x1w[7 * WF_PITCH] = z[7];
  wave_sync();
#pragma unroll
  for (int q = 0; q < 8; ++q) z[q] = p.x1r[8 * q];
  // pass 2 (stages 4 .. 6), same register pairing
  {
    const double2 w0 = p.t2[0];
    bfly(z[0], z[4], w0, sgn); bfly(z[2], z[6], w0, sgn); bfly(z[1], z[5], w0, sgn); bfly(z[3], z[7], w0, sgn);
    const double2 w1 = p.t2[8], w2 = p.t2[16];
    bfly(z[0], z[2], w1, sgn); bfly(z[4], z[6], w2, sgn); bfly(z[1], z[3], w1, sgn); bfly(z[5], z[7], w2, sgn);
    const double2 w3 = p.t2[24], w4 = p.t2[32], w5 = p.t2[40], w6 = p.t2[48];
    bfly(z[0], z[1], w3, sgn); bfly(z[4], z[5], w4, sgn); bfly(z[2], z[3], w5, sgn); bfly(z[6], z[7], w6, sgn);
  }
  wave_sync();          // every lane has its exchange-1 values before the buffer is rewritten
  // exchange 2 (natural order: slot = position)
  p.x2w[8 * 0] = z[0]; p.x2w[8 * 4] = z[1]; p.x2w[8 * 2] = z[2]; p.x2w[8 * 6] = z[3];
  p.x2w[8 * 1] = z[4]; p.x2w[8 * 5] = z[5]; p.x2w[8 * 3] = z[6]; p.x2w[8 * 7] = z[7];
  wave_sync();
#pragma unroll
  for (int q = 0; q < 8; ++q) z[q] = p.x2r[64 * q];
  // pass 3 (stages 7 .. 9): register c holds position lane + 64 c
  {
    const double2 w0 = p.t3[0];
    bfly(z[0], z[1], w0, sgn); bfly(z[2], z[3], w0, sgn); bfly(z[4], z[5], w0, sgn); bfly(z[6], z[7], w0, sgn);
    const double2 w1 = p.t3[64], w2 = p.t3[128];
    bfly(z[0], z[2], w1, sgn); bfly(z[1], z[3], w2, sgn); bfly(z[4], z[6], w1, sgn); bfly(z[5], z[7], w2, sgn);
    const double2 w3 = p.t3[192], w4 = p.t3[256], w5 = p.t3[320], w6 = p.t3[384];
    bfly(z[0], z[4], w3, sgn); bfly(z[1], z[5], w4, sgn); bfly(z[2], z[6], w5, sgn); bfly(z[3], z[7], w6, sgn);
  }
  wave_sync();          // the buffer is free for the caller (and for the next transform)
}

// Stores layout A to the natural slots; partner(q) then fetches the mirror element 512 - m of register q
// (lane 0, register 0 reads slot 512: unused by the callers).  The callers fetch the partners four at
// a time (eight more complex registers in flight do not fit beside a frame kernel's own state).
__device__ __forceinline__ void partners_store(const double2 (&z)[8], const Plan512& p) {
#pragma unroll
  for (int q = 0; q < 8; ++q) p.x2r[64 * q] = z[q];
  wave_sync();
}
__device__ __forceinline__ double2 partner(const Plan512& p, int q) { return p.xpr[64 * (7 - q)]; }

// Real transform of 1024 samples x, packed as z[m] = (x[2m], x[2m+1]) in layout A.  On return
// z[q] = X[lane + 64 q] (numpy.fft.rfft), and x512 = X[512] (valid in lane 0, computed by all).
__device__ __forceinline__ void rfft1024(double2 (&z)[8], double2& x512, const Plan512& p) {
  cfft512(z, p, -1.0);
  partners_store(z, p);
  const int l = lane_id();
  const double2 z0 = z[0];
#pragma unroll
  for (int q0 = 0; q0 < 8; q0 += 4) {
    double2 pz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pz[i] = partner(p, q0 + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + i;
      const bool lo = q < 4;                        // m < 256: the "k" output of the pair (k, 512 - k); else its "j" output
      const double2 zk = lo ? z[q] : pz[i], zj = lo ? pz[i] : z[q];
      const double2 w = lo ? p.tw_lo[64 * q] : p.tw_hi[64 * (7 - q)];
      const double er = 0.5 * (zk.x + zj.x), ei = 0.5 * (zk.y - zj.y);
      const double dr = 0.5 * (zk.x - zj.x), di = 0.5 * (zk.y + zj.y);
      const double orr = di, oi = -dr;  // O = -i D
      const double wr = w.x, wi = -w.y;  // w^k = e^{-2 pi i k / n}
      const double tr = orr * wr - oi * wi, ti = orr * wi + oi * wr;
      z[q] = lo ? make_double2(er + tr, ei + ti) : make_double2(er - tr, -(ei - ti));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_sync();
  if (l == 0) z[0] = make_double2(z0.x + z0.y, 0.0);
  x512 = make_double2(z0.x - z0.y, 0.0);
}

// Inverse real transform: z[q] = X[lane + 64 q], x512 = X[512] (read in lane 0) -> 1024 real samples
// packed as z[m] = (x[2m], x[2m+1]) in layout A, normalised like numpy.fft.irfft.
__device__ __forceinline__ void irfft1024(double2 (&z)[8], const double2 x512, const Plan512& p) {
  const int l = lane_id();
  partners_store(z, p);
  if (l == 0) z[0].y = 0.0;          // k = 0: the imaginary parts of X[0] and X[512] are ignored
#pragma unroll
  for (int q0 = 0; q0 < 8; q0 += 4) {
    double2 pz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pz[i] = partner(p, q0 + i);
    if (q0 == 0 && l == 0) pz[0] = make_double2(x512.x, 0.0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + i;
      const bool lo = q < 4;
      const double2 xk = lo ? z[q] : pz[i], xj = lo ? pz[i] : z[q];
      const double2 w = lo ? p.tw_lo[64 * q] : p.tw_hi[64 * (7 - q)];  // conj(w^k) = e^{+2 pi i k / n}
      const double er = 0.5 * (xk.x + xj.x), ei = 0.5 * (xk.y - xj.y);
      const double dr = 0.5 * (xk.x - xj.x), di = 0.5 * (xk.y + xj.y);
      const double orr = dr * w.x - di * w.y, oi = dr * w.y + di * w.x;
      const double2 zk = make_double2(er - oi, ei + orr);
      const double2 zj = make_double2(er + oi, -ei + orr);
      // m = 256 (lane 0, register 4) pairs with itself and takes the "k" form, like every m <= 256
      z[q] = (lo || (q == 4 && l == 0)) ? zk : zj;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_sync();
  cfft512(z, p, +1.0);
  const double s = 1.0 / (double)WF_N;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    z[q].x *= s;
    z[q].y *= s;
  }
}

// A real sequence given per spectral index in layout A (v[q] = x[lane + 64 q], v512 = x[512]) ->
// packed z[m] = (x[2m], x[2m+1]) in layout A.  even: x[1024 - n] = x[n]; otherwise zero beyond 512.
// (`even` is a run-time flag so that the two transforms of a minimum-phase construction share code.)
__device__ __forceinline__ void pack_real(const double (&v)[8], const double v512, double2 (&z)[8], const Plan512& p,
                                          const bool even) {
  const int l = lane_id();
  double* s = reinterpret_cast<double*>(p.x2r) - l * 2;      // the wave's buffer as doubles
#pragma unroll
  for (int q = 0; q < 8; ++q) s[l + 64 * q] = v[q];
  if (l == 0) { s[512] = v512; s[513] = 0.0; }
  wave_sync();
  const double2* s2 = reinterpret_cast<const double2*>(s);
#pragma unroll
  for (int q = 0; q < 4; ++q) z[q] = s2[l + 64 * q];                  // m < 256: (x[2m], x[2m+1])
  // m >= 256, even: (x[1024 - 2m], x[1023 - 2m]); m = 256 (lane 0): (x[512], x[511] or 0)
#pragma unroll
  for (int q = 4; q < 8; ++q) {
    const int n = 1024 - 2 * (l + 64 * q);        // 512 .. 2
    const double a = s[n], b = s[n - 1];
    const bool first = q == 4 && l == 0;          // m = 256
    z[q] = make_double2((even || first) ? a : 0.0, even ? b : 0.0);
  }
  wave_sync();
}

// ---- 1024 complex points (2048-point real transforms: 44.1 / 48 kHz) ----------------------------------
// Sixteen points per lane; passes of 4 + 3 + 3 stages (passes 2 and 3 work on two independent groups of
// eight registers); same rules as above (scripts/wave_fft_sim.py checks this plan as well).
constexpr int WF16_N = 1024;
constexpr int WF16_PITCH = 66;                            // conflict-free reads of the first exchange
constexpr int WF16_LDS_BYTES = 16 * WF16_PITCH * 16;      // 16 896 B per wave
constexpr int WF16_T3 = 7 * 16, WF16_TP = WF16_T3 + 14 * 64;
constexpr int WF16_TABLE_BYTES = (WF16_TP + 520) * 16;    // [7][16] pass 2, [14][64] pass 3, tw[0 .. 512]: 24 448 B

__device__ __forceinline__ int bitrev4(int x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x >> 1) & 2) | ((x >> 3) & 1); }

struct Plan1024 {
  const double2* t2;    // pass 2, a = lane & 15: t2[16 i] = tw[64 a]; tw[32 a + {0, 512}]; tw[16 a + {0, 256, 512, 768}]
  const double2* t3;    // pass 3, M = lane: t3[64 i] = tw[8 (M + 64 {0,1})]; tw[4 (M + 64 {0..3})]; tw[2 (M + 64 {0..7})]
  double2 c[7];         // pass 1 (wave-uniform): tw[128 k], k = 1 .. 7
  const double2* tw_lo; // (table) tw + lane (+ 64 q): registers 0 .. 7
  const double2* tw_hi; // (table) tw + (64 - lane) (+ 64 (15 - q)): registers 8 .. 15
  double2 *x1w, *x1r, *x2w, *x2r, *xpr;
};

// tw: the compact table of the 2048-point real transform, tw[k] = exp(+2 pi i k / 2048), k < 1024
__device__ __forceinline__ void table1024_init(void* table, const double2* __restrict__ tw) {
  double2* t = reinterpret_cast<double2*>(table);
  for (int i = threadIdx.x; i < WF16_T3; i += blockDim.x) {
    const int e = i >> 4, a = i & 15;
    const int idx = e == 0 ? 64 * a : (e < 3 ? 32 * a + 512 * (e - 1) : 16 * a + 256 * (e - 3));
    t[i] = tw[idx];
  }
  for (int i = threadIdx.x; i < 14 * 64; i += blockDim.x) {
    const int e = i >> 6, m = i & 63;
    const int idx = e < 2 ? 8 * (m + 64 * e) : (e < 6 ? 4 * (m + 64 * (e - 2)) : 2 * (m + 64 * (e - 6)));
    t[WF16_T3 + i] = tw[idx];
  }
  for (int i = threadIdx.x; i <= 512; i += blockDim.x) t[WF16_TP + i] = tw[i];
  __syncthreads();
}

__device__ __forceinline__ void plan1024_init(Plan1024& p, const double2* __restrict__ tw, void* lds, const void* table) {
  const int l = lane_id(), a = l & 15, hh = l >> 4;
  const int cc = ((hh & 1) << 1) | (hh >> 1);           // (bit 8, bit 9) of the positions this lane holds in pass 2
  const double2* t = reinterpret_cast<const double2*>(table);
  p.t2 = t + a;
  p.t3 = t + WF16_T3 + l;
#pragma unroll
  for (int k = 0; k < 7; ++k) p.c[k] = tw[128 * (k + 1)];
  p.tw_lo = t + WF16_TP + l;
  p.tw_hi = t + WF16_TP + (64 - l);
  double2* s = reinterpret_cast<double2*>(lds);
  p.x1w = s + l;
  p.x1r = s + a * WF16_PITCH + hh;
  p.x2w = s + a + 256 * (cc & 1) + 512 * (cc >> 1);
  p.x2r = s + l;
  p.xpr = s + (64 - l);
}

__device__ __forceinline__ void cfft1024(double2 (&z)[16], const Plan1024& p, const double sgn) {
  // pass 1 (stages 1 .. 4): register q holds local position bitrev4(q) of sixteen consecutive points
  bfly1(z[0], z[8]); bfly1(z[4], z[12]); bfly1(z[2], z[10]); bfly1(z[6], z[14]);
  bfly1(z[1], z[9]); bfly1(z[5], z[13]); bfly1(z[3], z[11]); bfly1(z[7], z[15]);
  bfly1(z[0], z[4]); bfly(z[8], z[12], p.c[3], sgn); bfly1(z[2], z[6]); bfly(z[10], z[14], p.c[3], sgn);
  bfly1(z[1], z[5]); bfly(z[9], z[13], p.c[3], sgn); bfly1(z[3], z[7]); bfly(z[11], z[15], p.c[3], sgn);
  bfly1(z[0], z[2]); bfly(z[8], z[10], p.c[1], sgn); bfly(z[4], z[6], p.c[3], sgn); bfly(z[12], z[14], p.c[5], sgn);
  bfly1(z[1], z[3]); bfly(z[9], z[11], p.c[1], sgn); bfly(z[5], z[7], p.c[3], sgn); bfly(z[13], z[15], p.c[5], sgn);
  bfly1(z[0], z[1]); bfly(z[8], z[9], p.c[0], sgn); bfly(z[4], z[5], p.c[1], sgn); bfly(z[12], z[13], p.c[2], sgn);
  bfly(z[2], z[3], p.c[3], sgn); bfly(z[10], z[11], p.c[4], sgn); bfly(z[6], z[7], p.c[5], sgn); bfly(z[14], z[15], p.c[6], sgn);
  // exchange 1: position 16 V + j at slot j * PITCH + lane
#pragma unroll
  for (int q = 0; q < 16; ++q) p.x1w[bitrev4(q) * WF16_PITCH] = z[q];
  wave_sync();
#pragma unroll
  for (int q = 0; q < 16; ++q) z[q] = p.x1r[8 * (q & 7) + 4 * (q >> 3)];
  // pass 2 (stages 5 .. 7): two groups of eight registers, the pairing of the 512-point pass 2
  {
    const double2 w0 = p.t2[0], w1 = p.t2[16], w2 = p.t2[32];
    const double2 w3 = p.t2[48], w4 = p.t2[64], w5 = p.t2[80], w6 = p.t2[96];
#pragma unroll
    for (int g = 0; g < 16; g += 8) {
      bfly(z[g + 0], z[g + 4], w0, sgn); bfly(z[g + 2], z[g + 6], w0, sgn); bfly(z[g + 1], z[g + 5], w0, sgn); bfly(z[g + 3], z[g + 7], w0, sgn);
      bfly(z[g + 0], z[g + 2], w1, sgn); bfly(z[g + 4], z[g + 6], w2, sgn); bfly(z[g + 1], z[g + 3], w1, sgn); bfly(z[g + 5], z[g + 7], w2, sgn);
      bfly(z[g + 0], z[g + 1], w3, sgn); bfly(z[g + 4], z[g + 5], w4, sgn); bfly(z[g + 2], z[g + 3], w5, sgn); bfly(z[g + 6], z[g + 7], w6, sgn);
    }
  }
  wave_sync();
  // exchange 2 (natural order: slot = position)
#pragma unroll
  for (int q = 0; q < 16; ++q) p.x2w[16 * bitrev3(q & 7) + 128 * (q >> 3)] = z[q];
  wave_sync();
#pragma unroll
  for (int q = 0; q < 16; ++q) z[q] = p.x2r[64 * q];
  // pass 3 (stages 8 .. 10): register c holds position lane + 64 c
  {
    const double2 w8a = p.t3[0], w8b = p.t3[64];
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (!(c & 2)) bfly(z[c], z[c + 2], (c & 1) ? w8b : w8a, sgn);
    const double2 w9[4] = {p.t3[128], p.t3[192], p.t3[256], p.t3[320]};
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (!(c & 4)) bfly(z[c], z[c + 4], w9[c & 3], sgn);
#pragma unroll
    for (int c = 0; c < 8; ++c) bfly(z[c], z[c + 8], p.t3[384 + 64 * c], sgn);
  }
  wave_sync();
}

// ---- what the real transforms share, for either size (R = 8: Plan512, R = 16: Plan1024) -----------------
template <int R> struct PlanOf;
template <> struct PlanOf<8> { typedef Plan512 type; };
template <> struct PlanOf<16> { typedef Plan1024 type; };
template <int R> constexpr int lds_bytes() { return R == 8 ? WF_LDS_BYTES : WF16_LDS_BYTES; }
template <int R> constexpr int table_bytes() { return R == 8 ? WF_TABLE_BYTES : WF16_TABLE_BYTES; }
template <int R>
__device__ __forceinline__ void table_init(void* table, const double2* __restrict__ tw) {
  if constexpr (R == 8) table512_init(table, tw); else table1024_init(table, tw);
}
__device__ __forceinline__ void plan_init(Plan512& p, const double2* tw, void* lds, const void* table) { plan512_init(p, tw, lds, table); }
__device__ __forceinline__ void plan_init(Plan1024& p, const double2* tw, void* lds, const void* table) { plan1024_init(p, tw, lds, table); }
__device__ __forceinline__ void cfft(double2 (&z)[8], const Plan512& p, double sgn) { cfft512(z, p, sgn); }
__device__ __forceinline__ void cfft(double2 (&z)[16], const Plan1024& p, double sgn) { cfft1024(z, p, sgn); }

// Real transform of 128 R samples packed as z[m] = (x[2m], x[2m+1]) in layout A.  On return
// z[q] = X[lane + 64 q], xh = X[64 R] (valid in lane 0).  Same arithmetic as rfft1024 above.
template <int R, class Plan>
__device__ __forceinline__ void rfft(double2 (&z)[R], double2& xh, const Plan& p) {
  cfft(z, p, -1.0);
#pragma unroll
  for (int q = 0; q < R; ++q) p.x2r[64 * q] = z[q];
  wave_sync();
  const int l = lane_id();
  const double2 z0 = z[0];
#pragma unroll
  for (int q0 = 0; q0 < R; q0 += 4) {
    double2 pz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pz[i] = p.xpr[64 * (R - 1 - (q0 + i))];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + i;
      const bool lo = q < R / 2;
      const double2 zk = lo ? z[q] : pz[i], zj = lo ? pz[i] : z[q];
      const double2 w = lo ? p.tw_lo[64 * q] : p.tw_hi[64 * (R - 1 - q)];
      const double er = 0.5 * (zk.x + zj.x), ei = 0.5 * (zk.y - zj.y);
      const double dr = 0.5 * (zk.x - zj.x), di = 0.5 * (zk.y + zj.y);
      const double orr = di, oi = -dr;  // O = -i D
      const double wr = w.x, wi = -w.y;  // w^k = e^{-2 pi i k / n}
      const double tr = orr * wr - oi * wi, ti = orr * wi + oi * wr;
      z[q] = lo ? make_double2(er + tr, ei + ti) : make_double2(er - tr, -(ei - ti));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_sync();
  if (l == 0) z[0] = make_double2(z0.x + z0.y, 0.0);
  xh = make_double2(z0.x - z0.y, 0.0);
}

// pack_real for either size: a real sequence per spectral index in layout A (v[q] = x[lane + 64 q], vh = x[64 R])
// -> packed z[m] = (x[2m], x[2m+1]) in layout A.  even: x[128 R - n] = x[n]; otherwise zero beyond 64 R.
template <int R, class Plan>
__device__ __forceinline__ void pack_real_r(const double (&v)[R], const double vh, double2 (&z)[R], const Plan& p,
                                            const bool even) {
  const int l = lane_id();
  double* s = reinterpret_cast<double*>(p.x2r) - l * 2;      // the wave's buffer as doubles
#pragma unroll
  for (int q = 0; q < R; ++q) s[l + 64 * q] = v[q];
  if (l == 0) { s[64 * R] = vh; s[64 * R + 1] = 0.0; }
  wave_sync();
  const double2* s2 = reinterpret_cast<const double2*>(s);
#pragma unroll
  for (int q = 0; q < R / 2; ++q) z[q] = s2[l + 64 * q];
#pragma unroll
  for (int q = R / 2; q < R; ++q) {
    const int n = 128 * R - 2 * (l + 64 * q);
    const double a = s[n], b = s[n - 1];
    const bool first = q == R / 2 && l == 0;      // m = 32 R
    z[q] = make_double2((even || first) ? a : 0.0, even ? b : 0.0);
  }
  wave_sync();
}

// Inverse: z[q] = X[lane + 64 q], xh = X[64 R] (read in lane 0) -> 128 R real samples packed in layout A.
template <int R, class Plan>
__device__ __forceinline__ void irfft(double2 (&z)[R], const double2 xh, const Plan& p) {
  const int l = lane_id();
#pragma unroll
  for (int q = 0; q < R; ++q) p.x2r[64 * q] = z[q];
  wave_sync();
  if (l == 0) z[0].y = 0.0;
#pragma unroll
  for (int q0 = 0; q0 < R; q0 += 4) {
    double2 pz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pz[i] = p.xpr[64 * (R - 1 - (q0 + i))];
    if (q0 == 0 && l == 0) pz[0] = make_double2(xh.x, 0.0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + i;
      const bool lo = q < R / 2;
      const double2 xk = lo ? z[q] : pz[i], xj = lo ? pz[i] : z[q];
      const double2 w = lo ? p.tw_lo[64 * q] : p.tw_hi[64 * (R - 1 - q)];
      const double er = 0.5 * (xk.x + xj.x), ei = 0.5 * (xk.y - xj.y);
      const double dr = 0.5 * (xk.x - xj.x), di = 0.5 * (xk.y + xj.y);
      const double orr = dr * w.x - di * w.y, oi = dr * w.y + di * w.x;
      const double2 zk = make_double2(er - oi, ei + orr);
      const double2 zj = make_double2(er + oi, -ei + orr);
      z[q] = (lo || (q == R / 2 && l == 0)) ? zk : zj;      // m = 32 R pairs with itself: the "k" form
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_sync();
  cfft(z, p, +1.0);
  const double s = 1.0 / (double)(64 * R);
#pragma unroll
  for (int q = 0; q < R; ++q) {
    z[q].x *= s;
    z[q].y *= s;
  }
}

}  // namespace wf
}  // namespace itts
