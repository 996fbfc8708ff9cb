// fp32-MFMA GEMM fed by LDS-DMA loader waves: the kernel behind the dense layers of the acoustic
// model (rnn_dyn/FFWrapper.py:63-73 forward, autograd's dX / dW products behind it).
//
// Structure (gfx950, written for it):
//   * persistent workgroups of EIGHT waves, two per CU, walk a static list of output tiles of
//     128 x 64 (or 64 x 128): waves 0-3 compute (64 x 32 each = two v_mfma_f32_32x32x2_f32
//     accumulators), waves 4-7 only move data.  (Workgroups of five or six waves are never placed
//     two to a CU on this chip, whatever the occupancy query says; 256- and 512-thread ones are.)
//   * each loader wave brings a quarter of the operand tiles of 32 reduction elements global -> LDS
//     with `buffer_load_dwordx4 ... lds` (1 KB per instruction, no VGPRs, no ds_write) into a ring
//     of three 24-KB slots, two K-steps ahead of the compute waves and straight across tile
//     boundaries, so neither the first loads of a tile nor its epilogue expose memory latency; its
//     counted `s_waitcnt vmcnt(6)` is the only wait on that traffic.  (The DMA is inline asm: hipcc
//     would otherwise drain it with vmcnt(0) in front of every ds_read of the ring.)  The compute
//     waves issue no memory instruction inside the K loop except ds_read, and their epilogue
//     stores are never waited for;
//   * one workgroup barrier per K-step (everybody has left step s-1, whose slot the loaders refill
//     next; the loaders have seen step s land);
//   * row-form tiles [out][32 k] are stored unpadded (the DMA writes lane-linear) with the 16-byte
//     chunk index XORed with (row >> 1) & 7 -- applied to the SOURCE address of the DMA and to the
//     ds_read_b128 address -- which makes the fragment reads bank-conflict free; col-form tiles
//     [32 k][out] are linear and read with ds_read_b32;
//   * edges without branches in the loop: buffer descriptors end at the last valid row, so
//     out-of-range rows of an operand arrive as zeros; the columns k >= K of a partial last K-step
//     of a row-form tile are zeroed in LDS behind the step's barrier (one more barrier on that step
//     only); stores go through a descriptor as well (rows beyond M are dropped);
//   * epilogue without LDS: 4 x 4 transposes inside lane quads (DPP) turn the accumulator layout
//     (lane = column) into float4 row segments; bias + activation / activation derivative / masked
//     MSE are fused as in the register-staged kernel this one replaces.
//
// Two hardware traps found on the way, both only with two workgroups per CU (memory queue busy):
//   * a 16-byte buffer store with an SGPR offset may fetch its data registers AFTER a following
//     VALU write to them (lost elements at lanes 12 / 28 / 44 / 60): the stores carry their whole
//     offset in the VGPR, and the accumulators are never cleared by VALU writes -- a new tile
//     starts from a zero C operand instead;
//   * the clock follows the previous kernel for milliseconds: time a kernel after it has run for a
//     few hundred launches, never right behind another one.
//
// K order inside the MFMA chain equals the register-staged kernel's (nn.hip), so results are
// bit-identical to it for the same split-K chunking.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef RING_DBG
#define RING_DBG 0   // lab only: 1 no DMA, 2 no step barrier, 4 no MFMA, 8 no epilogue, 16 no fragment reads
#endif

namespace itts {
namespace ring {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* lds_char_p;

constexpr int RBK = 32;                       // reduction elements per K-step
constexpr int SLOT_BYTES = 192 * 128;         // A tile + B tile of one K-step: (128 + 64) x 128 B
#ifndef RING_NSLOT
#define RING_NSLOT 3
#endif
constexpr int NSLOT = RING_NSLOT;
constexpr int RED_BYTES = 2 * 256 * 4;        // bias-gradient partial sums, double buffered
constexpr int LDS_BYTES = NSLOT * SLOT_BYTES + RED_BYTES;
constexpr int PIECES = 6;                     // 1-KB DMA instructions per loader wave and K-step
constexpr int THREADS = 512;         // 4 compute waves + 4 loader waves (workgroups of 5 or 6 waves do not share a CU)

enum { EPI_STORE = 0, EPI_BIAS_ACT = 1, EPI_DACT = 2, EPI_MSE = 3 };
enum { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2 };

struct Args {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* aux;
  const uint8_t* row_valid;   // EPI_MSE
  double* loss_partial;       // EPI_MSE: [gridDim.x]
  float* bias_part;           // col-form A: column sums of A over the slab (bias gradient), tn == 0
  uint64_t* stamps;           // diagnostics (NULL in product calls): per workgroup {shader cycles, 100 MHz ticks}
  int64_t slab_stride;        // floats between slabs of C
  int64_t bias_part_stride;
  int lda, ldb, ldc, ldaux;   // pitches in floats (multiples of 4)
  int M;                      // output rows (A's out dimension)
  int N;                      // output cols (B's out dimension)
  int K;                      // reduction length
  int kchunk;                 // reduction elements per split-K slab (multiple of 32)
  int splitk;
  int tiles_m, tiles_n;
  int gn;                     // column tiles per group of the tile order (decode_tile); >= tiles_n: one group
  int act;
  float gscale;               // EPI_MSE
};

__device__ __forceinline__ float fast_tanhf(float z) {
  const float a = fabsf(z);
  const float z2 = z * z;
  const float poly = z * (1.f + z2 * (-0.33333334f + z2 * (0.13333334f + z2 * (-0.053968254f +
                                                                              z2 * 0.021869488f))));
  const float e = __expf(2.f * a);
  const float big = copysignf(1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f), z);
  return a < 0.25f ? poly : big;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// One 1-KB LDS-DMA piece: lane l fetches 16 bytes at rsrc.base + voff + soff, the wave's 64 chunks
// land at LDS byte address `dst` + 16 l.  Hidden from hipcc's wait counting on purpose.
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, uint32_t dst) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :
               : "v"(voff), "s"(rsrc), "s"(soff), "s"(dst)
               : "memory");
}

// Workgroup barrier that also publishes this wave's LDS writes (the raw s_barrier waits for nothing).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void step_barrier() {
  if (!(RING_DBG & 2)) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Fragment (4 reduction elements per lane: k = 8 g + 4 h + j) of the 32-out block whose per-lane
// base address is `a0`.
template <bool ROW, int BT>
__device__ __forceinline__ float4 read_frag(const char* tile, uint32_t a0, int g) {
  if (ROW) {
    return *reinterpret_cast<const float4*>(tile + (a0 ^ (uint32_t)(g << 5)));
  } else {
    const float* p = reinterpret_cast<const float*>(tile + a0) + g * 8 * BT;
    return make_float4(p[0], p[BT], p[2 * BT], p[3 * BT]);
  }
}
template <bool ROW, int BT>
__device__ __forceinline__ uint32_t frag_base(int o, int lane) {
  const int r = lane & 31, h = lane >> 5;
  if (ROW) {
    const int row = o + r;
    return (uint32_t)(row * 128 + ((((row >> 1) & 7) ^ h) << 4));
  } else {
    return (uint32_t)((4 * h * BT + o + r) * 4);
  }
}

struct Tile {
  int m0, n0, z, tn;
  int kbeg, klen, nk;
};

template <bool GROUPED>
__device__ __forceinline__ Tile decode_tile(const Args& g, int t, int bmt, int bnt) {
  Tile c;
  const int per_z = g.tiles_m * g.tiles_n;
  c.z = t / per_z;
  const int rem = t - c.z * per_z;
  // order inside a slab: groups of gn column tiles; inside a group row by row.  The workgroups of an XCD
  // walk neighbouring tiles at the same time, so a group's B panels (gn x bnt x K) stay in its L2 while
  // the A panels stream past once per group -- with one group (gn >= tiles_n) it is B that streams past
  // every pair of A panels, which costs a wide, deep B (the recurrent layers' 4096 x 1024 weights: 16 MB
  // against 4 MB of L2) a trip beyond the L2 per tile.
  int tm;
  if (!GROUPED) {                   // one group: column tile fastest
    tm = rem / g.tiles_n;
    c.tn = rem - tm * g.tiles_n;
  } else {
    const int per_group = g.tiles_m * g.gn;
    const int grp = rem / per_group;
    const int rem2 = rem - grp * per_group;
    const int left = g.tiles_n - grp * g.gn;
    const int width = left < g.gn ? left : g.gn;
    tm = rem2 / width;
    c.tn = grp * g.gn + (rem2 - tm * width);
  }
  c.m0 = tm * bmt;
  c.n0 = c.tn * bnt;
  c.kbeg = c.z * g.kchunk;
  const int kend = c.kbeg + g.kchunk < g.K ? c.kbeg + g.kchunk : g.K;
  c.klen = kend - c.kbeg;
  c.nk = (c.klen + RBK - 1) / RBK;
  return c;
}

// This workgroup's tiles: XCD x owns a contiguous range, its workgroups interleave in it.
struct Walk {
  int first, end, stride;
};
__device__ __forceinline__ Walk my_tiles(const Args& g) {
  const int ntiles = g.tiles_m * g.tiles_n * g.splitk;
  const int xcd = blockIdx.x & 7, rank = blockIdx.x >> 3;
  const int q = ntiles >> 3, r8 = ntiles & 7;
  const int xstart = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
  Walk w;
  w.first = xstart + rank;
  w.end = xstart + q + (xcd < r8 ? 1 : 0);
  w.stride = ((int)gridDim.x - xcd + 7) >> 3;
  return w;
}

// ------------------------------------------------------------------------------------------------
// loader wave
// ------------------------------------------------------------------------------------------------
// Source stream of one operand for one tile: descriptor and byte step of a K-step.
template <bool ROW, int BT>
struct Stream {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t soff, step;

  __device__ __forceinline__ void open(const float* P, int ld, int out0, int out_dim, const Tile& c) {
    if (ROW) {
      int rows = out_dim - out0;
      rows = rows < BT ? rows : BT;
      step = RBK * 4;
      // row >= rows: beyond the descriptor -> zeros.  The descriptor ends at the LAST VALID ELEMENT (last
      // row, column kbeg + klen - 1), not at the end of that row's pitch: the operand may be a column
      // slice of a wider tensor, whose last pitch reaches past the end of the allocation by the
      // slice's column offset (a fault when the tensor ends where the mapping ends).
      rsrc = make_rsrc(P + (int64_t)out0 * ld + c.kbeg, (uint32_t)(((int64_t)(rows - 1) * ld + ((c.klen + 3) & ~3)) * 4));
    } else {
      step = (uint32_t)(RBK * ld * 4);
      // k >= kend: beyond the descriptor -> zeros; it ends at the last valid element (see above; rounded up
      // to the 16-byte chunk, which stays inside the row: bases and pitches are multiples of 16 bytes)
      rsrc = make_rsrc(P + (int64_t)c.kbeg * ld + out0,
                       (uint32_t)(((int64_t)(c.klen - 1) * ld + ((out_dim - out0 + 3) & ~3)) * 4));
    }
    soff = 0;
  }
  // this loader wave's quarter (pieces part * BT / 32 ...) of the current K-step into the LDS tile
  // at byte address dst
  __device__ __forceinline__ void issue(uint32_t v0, uint32_t v1, int ld, uint32_t dst, int part) const {
    constexpr int NP = BT / 32;   // pieces per loader wave: 4 or 2
    if (ROW) {
      // piece j: rows 8 j .. 8 j + 7; the swizzle of a lane's chunk depends on j & 1 only
      const uint32_t pstep = (uint32_t)(16 * ld * 4);
      uint32_t so = soff + (uint32_t)(part * (NP / 2)) * pstep;
      const uint32_t d = dst + part * NP * 1024;
#pragma unroll
      for (int j = 0; j < NP; j += 2) {
        dma_piece(rsrc, v0, so, d + j * 1024);
        dma_piece(rsrc, v1, so, d + (j + 1) * 1024);
        so += pstep;
      }
    } else {
      // piece j: k rows j * (256 / BT) ..
      const uint32_t pstep = (uint32_t)((256 / BT) * ld * 4);
      uint32_t so = soff + (uint32_t)(part * NP) * pstep;
      const uint32_t d = dst + part * NP * 1024;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        dma_piece(rsrc, v0, so, d + j * 1024);
        so += pstep;
      }
    }
  }
};
template <bool ROW, int BT>
__device__ __forceinline__ void lane_offsets(int ld, int lane, uint32_t& v0, uint32_t& v1) {
  if (ROW) {
    const int r = lane >> 3, c = lane & 7;
    v0 = (uint32_t)(r * ld * 4 + ((c ^ ((r >> 1) & 7)) << 4));              // even pieces: row = 16 i + r
    v1 = (uint32_t)((8 + r) * ld * 4 + ((c ^ (((8 + r) >> 1) & 7)) << 4));  // odd pieces: row = 16 i + 8 + r
  } else {
    constexpr int LPR = BT / 4;   // lanes (16-byte chunks) per k row
    v0 = (uint32_t)((lane / LPR) * ld * 4 + (lane % LPR) * 16);
    v1 = v0;
  }
}

// Zeroes the elements k >= krem of a row-form tile of BT rows (partial last K-step); the 256
// threads of the compute waves, behind the barrier that published the tile (an LDS access of the
// loader itself right after its vmcnt wait can still overtake the DMA's write).
template <int BT>
__device__ __forceinline__ void zero_tail(char* tile, int krem, int tid) {
#pragma unroll
  for (int i = 0; i < BT / 32; ++i) {
    const int idx = tid + 256 * i;
    const int row = idx >> 3, c = idx & 7;
    const int lim = krem - 4 * c;   // components j < lim stay
    if (lim < 4) {
      float4* p = reinterpret_cast<float4*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      float4 v = *p;
      v.x = lim > 0 ? v.x : 0.f;
      v.y = lim > 1 ? v.y : 0.f;
      v.z = lim > 2 ? v.z : 0.f;
      v.w = 0.f;
      *p = v;
    }
  }
}

template <bool A_ROW, bool B_ROW, int BMT, int BNT, bool GROUPED = false>
__device__ __forceinline__ void loader_wave(const Args& g, char* lds, uint32_t lds0, int lane, bool mse, int part) {
  constexpr int A_BYTES = BMT * 128;
  const Walk w = my_tiles(g);
  uint32_t va0, va1, vb0, vb1;
  lane_offsets<A_ROW, BMT>(g.lda, lane, va0, va1);
  lane_offsets<B_ROW, BNT>(g.ldb, lane, vb0, vb1);
  Stream<A_ROW, BMT> sa;
  Stream<B_ROW, BNT> sb;

  // producer cursor (two K-steps ahead) and consumer cursor (the step the compute waves are at)
  int pt = w.first, pk = 0, pnk = 0;
  uint32_t pdst = lds0;
  bool pvalid = pt < w.end;
  auto open = [&]() {
    const Tile c = decode_tile<GROUPED>(g, pt, BMT, BNT);
    sa.open(g.A, g.lda, c.m0, g.M, c);
    sb.open(g.B, g.ldb, c.n0, g.N, c);
    pnk = c.nk;
    pk = 0;
  };
  auto produce = [&]() {
    if (!(RING_DBG & 1)) {
      sa.issue(va0, va1, g.lda, pdst, part);
      sb.issue(vb0, vb1, g.ldb, pdst + A_BYTES, part);
    }
    pdst = pdst == lds0 + (NSLOT - 1) * SLOT_BYTES ? lds0 : pdst + SLOT_BYTES;
    sa.soff += sa.step;
    sb.soff += sb.step;
    if (++pk == pnk) {
      pt += w.stride;
      pvalid = pt < w.end;
      if (pvalid) open();
    }
  };
  int ahead = 0;
  if (pvalid) {
    open();
    for (int i = 0; i < NSLOT - 1 && pvalid; ++i) { produce(); ++ahead; }
  }
  int ct = w.first, kt = 0, cslot = 0;
  bool have = ct < w.end;
  Tile c{};
  if (have) c = decode_tile<GROUPED>(g, ct, BMT, BNT);
  while (have) {
    // the pieces of this step have landed (those of the next one stay in flight)
    if (ahead >= 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PIECES) : "memory");
    else if (ahead == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PIECES) : "memory");
    else if (ahead == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
    else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    step_barrier();
    --ahead;
    if (pvalid) { produce(); ++ahead; }
    // a partial last K-step of a row-form tile: the compute waves zero its tail behind the barrier
    if ((A_ROW || B_ROW) && kt == c.nk - 1 && (c.klen & (RBK - 1)) != 0) step_barrier();
    cslot = cslot == NSLOT - 1 ? 0 : cslot + 1;
    if (++kt == c.nk) {
      kt = 0;
      ct += w.stride;
      have = ct < w.end;
      if (have) c = decode_tile<GROUPED>(g, ct, BMT, BNT);
    }
  }
  // barriers beyond the K-steps: the entry of the last epilogue, EPI_MSE's reduction
  if (w.first < w.end) step_barrier();
  if (mse) step_barrier();
}

// ------------------------------------------------------------------------------------------------
// compute waves
// ------------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ float act1(float z) {
  if (ACT == ACT_TANH) return fast_tanhf(z);
  if (ACT == ACT_RELU) return z > 0.f ? z : 0.f;
  return z;
}
template <int ACT>
__device__ __forceinline__ float dact1(float y) {
  if (ACT == ACT_TANH) return 1.f - y * y;
  if (ACT == ACT_RELU) return y > 0.f ? 1.f : 0.f;
  return 1.f;
}

template <int CTRL>
__device__ __forceinline__ float quad_swap(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// Lane 4 q + i holds a[r] = element (row r, column 4 q + i); afterwards it holds element (row i,
// column 4 q + c) in component c.
__device__ __forceinline__ float4 quad_transpose(float a0, float a1, float a2, float a3, int lane) {
  const bool odd = lane & 1, hi = lane & 2;
  const float s0 = quad_swap<0xB1>(a0), s1 = quad_swap<0xB1>(a1), s2 = quad_swap<0xB1>(a2), s3 = quad_swap<0xB1>(a3);
  const float x0 = odd ? s1 : a0, x1 = odd ? a1 : s0;
  const float x2 = odd ? s3 : a2, x3 = odd ? a3 : s2;
  const float t0 = quad_swap<0x4E>(x0), t1 = quad_swap<0x4E>(x1), t2 = quad_swap<0x4E>(x2), t3 = quad_swap<0x4E>(x3);
  return make_float4(hi ? t2 : x0, hi ? t3 : x1, hi ? x2 : t0, hi ? x3 : t1);
}

// Epilogue of one finished tile: a wave's two 32 x 32 accumulator blocks leave as float4 row
// segments (8 store instructions of whole 128-byte lines).
// bias of the four columns a lane stores (columns >= N: 0)
template <int EPI>
__device__ __forceinline__ float4 load_bias(const Args& g, int n0, int wn, int lane) {
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((EPI == EPI_BIAS_ACT || EPI == EPI_MSE) && g.bias) {
    const int col = n0 + wn * 32 + 4 * ((lane & 31) >> 2);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(g.bias, (uint32_t)(g.N * 4));
    bv.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, col * 4, 0, 0));
    bv.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, col * 4 + 4, 0, 0));
    bv.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, col * 4 + 8, 0, 0));
    bv.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, col * 4 + 12, 0, 0));
  }
  return bv;
}

template <int EPI, int ACT>
__device__ __forceinline__ void epilogue(const Args& g, const Tile& pc, const f32x16& acc0, const f32x16& acc1,
                                         int wm, int wn, int lane, const float4& bv, double& lsum) {
  const int quad = (lane & 31) >> 2, j = lane & 3, h = lane >> 5;
  const int col = pc.n0 + wn * 32 + 4 * quad;
  const int rloc = wm * 64 + 4 * h + j;          // + 32 i + 8 rg
  int rows_valid = g.M - pc.m0;
  rows_valid = rows_valid < 0 ? 0 : rows_valid;
  // C through a descriptor: rows >= M are beyond it and dropped; a float4 beyond the pitch gets an
  // offset beyond it
  float* Cb = g.C + (int64_t)pc.z * g.slab_stride + (int64_t)pc.m0 * g.ldc;
  const __amdgpu_buffer_rsrc_t rc = make_rsrc(Cb, (uint32_t)((int64_t)rows_valid * g.ldc * 4));
  const uint32_t cofs = col < g.ldc ? (uint32_t)((rloc * g.ldc + col) * 4) : 0xfffffff0u;
  __amdgpu_buffer_rsrc_t rx = rc;
  uint32_t xofs = 0;
  if (EPI == EPI_DACT || EPI == EPI_MSE) {
    rx = make_rsrc(g.aux + (int64_t)pc.m0 * g.ldaux, (uint32_t)((int64_t)rows_valid * g.ldaux * 4));
    xofs = col < g.ldaux ? (uint32_t)((rloc * g.ldaux + col) * 4) : 0xfffffff0u;
  }
  const bool k0 = col < g.N, k1 = col + 1 < g.N, k2 = col + 2 < g.N, k3 = col + 3 < g.N;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x16& a = i == 0 ? acc0 : acc1;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int rl = i * 32 + 8 * rg;   // uniform part of the row
      float4 v = quad_transpose(a[4 * rg], a[4 * rg + 1], a[4 * rg + 2], a[4 * rg + 3], lane);
      if (EPI == EPI_BIAS_ACT) {
        v.x = act1<ACT>(v.x + bv.x); v.y = act1<ACT>(v.y + bv.y);
        v.z = act1<ACT>(v.z + bv.z); v.w = act1<ACT>(v.w + bv.w);
      }
      if (EPI == EPI_DACT) {
        const f32x4 y = __builtin_amdgcn_raw_buffer_load_b128(rx, xofs, rl * g.ldaux * 4, 0);
        v.x *= dact1<ACT>(y[0]); v.y *= dact1<ACT>(y[1]);
        v.z *= dact1<ACT>(y[2]); v.w *= dact1<ACT>(y[3]);
      }
      if (EPI == EPI_MSE) {
        // the target may have any pitch (e.g. 187 floats): four dword loads, not one 16-byte load
        f32x4 t;
        {
          const uint32_t so = (uint32_t)(rl * g.ldaux * 4);
          const uint32_t tofs = (uint32_t)((rloc * g.ldaux + col) * 4);
          t[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, tofs, so, 0));
          t[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, tofs + 4, so, 0));
          t[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, tofs + 8, so, 0));
          t[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, tofs + 12, so, 0));
        }
        const int row = pc.m0 + rloc + rl;
        const bool ok = row < g.M && g.row_valid[row < g.M ? row : 0] != 0;
        const float d0 = ok && k0 ? (v.x + bv.x) - t[0] : 0.f;
        const float d1 = ok && k1 ? (v.y + bv.y) - t[1] : 0.f;
        const float d2 = ok && k2 ? (v.z + bv.z) - t[2] : 0.f;
        const float d3 = ok && k3 ? (v.w + bv.w) - t[3] : 0.f;
        lsum += (double)d0 * (double)d0;
        lsum += (double)d1 * (double)d1;
        lsum += (double)d2 * (double)d2;
        lsum += (double)d3 * (double)d3;
        v = make_float4(g.gscale * d0, g.gscale * d1, g.gscale * d2, g.gscale * d3);
      }
      f32x4 o;
      o[0] = k0 ? v.x : 0.f;   // pad columns inside the pitch get zeros
      o[1] = k1 ? v.y : 0.f;
      o[2] = k2 ? v.z : 0.f;
      o[3] = k3 ? v.w : 0.f;
      // the whole offset in the VGPR: a 16-byte store with an SGPR offset may fetch its data registers
      // after a following VALU write to them (seen as lost elements when the CU's memory queue was
      // busy with the loaders' DMA)
      const uint32_t vo = col < g.ldc ? cofs + (uint32_t)(rl * g.ldc * 4) : 0xfffffff0u;
      if (RING_DBG & 64) asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
      else __builtin_amdgcn_raw_buffer_store_b128(o, rc, vo, 0, 0);
    }
  }
}

template <bool A_ROW, bool B_ROW, int EPI, int BMT, int BNT, bool GROUPED = false>
__device__ __forceinline__ void compute_waves(const Args& g, char* lds, int wid, int lane) {
  constexpr int WN = BNT / 32;
  constexpr int A_BYTES = BMT * 128;
  const int wm = wid / WN, wn = wid % WN;
  const int tid = wid * 64 + lane;   // 0 .. 255
  const Walk w = my_tiles(g);
  const uint32_t fa0 = frag_base<A_ROW, BMT>(wm * 64, lane);
  const uint32_t fa1 = frag_base<A_ROW, BMT>(wm * 64 + 32, lane);
  const uint32_t fb0 = frag_base<B_ROW, BNT>(wn * 32, lane);
  float* red = reinterpret_cast<float*>(lds + NSLOT * SLOT_BYTES);
  constexpr bool DO_BIAS = !A_ROW && EPI == EPI_STORE;
  constexpr int BPARTS = 256 / BMT;      // k ranges of the column sums

  double lsum = 0.0;   // EPI_MSE: this workgroup's sum of squared differences over all its tiles
  int cslot = 0, redbuf = 0;
  bool pending = false;      // an accumulator tile waits for its epilogue
  Tile pc{};                 // its coordinates
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float bsum = 0.f;

  int ct = w.first;
  bool have = ct < w.end;
  Tile c{};
  if (have) c = decode_tile<GROUPED>(g, ct, BMT, BNT);
  int kt = 0;
  while (have || pending) {
    // step entry: behind the barrier the slot of this step is complete and everybody has left the
    // previous one
    step_barrier();
    if (pending) {
      if (DO_BIAS && g.bias_part != nullptr && pc.tn == 0) {
        // the k ranges' column sums of A met in LDS before the barrier
        const float* rb = red + (redbuf ^ 1) * 256;
        if (tid < BMT && pc.m0 + tid < g.M) {
          float s = rb[tid];
#pragma unroll
          for (int p = 1; p < BPARTS; ++p) s += rb[tid + p * BMT];
          g.bias_part[(int64_t)pc.z * g.bias_part_stride + pc.m0 + tid] = s;
        }
      }
      if (RING_DBG & 8) {
        float t = 0.f;
        for (int r = 0; r < 16; ++r) t += acc0[r] + acc1[r];
        if (t == 1.2345f) g.C[0] = t;
      } else {
        const float4 bv = load_bias<EPI>(g, pc.n0, wn, lane);
        if (EPI == EPI_STORE || g.act == ACT_NONE) epilogue<EPI, ACT_NONE>(g, pc, acc0, acc1, wm, wn, lane, bv, lsum);
        else if (g.act == ACT_TANH) epilogue<EPI, ACT_TANH>(g, pc, acc0, acc1, wm, wn, lane, bv, lsum);
        else epilogue<EPI, ACT_RELU>(g, pc, acc0, acc1, wm, wn, lane, bv, lsum);
      }
      pending = false;
    }
    if (!have) break;   // only the last epilogue was left

    char* tA = lds + cslot * SLOT_BYTES;
    char* tB = tA + A_BYTES;
    if ((A_ROW || B_ROW) && kt == c.nk - 1 && (c.klen & (RBK - 1)) != 0) {
      if (A_ROW) zero_tail<BMT>(tA, c.klen & (RBK - 1), tid);
      if (B_ROW) zero_tail<BNT>(tB, c.klen & (RBK - 1), tid);
      lds_barrier();
    }
    if (DO_BIAS && g.bias_part != nullptr && c.tn == 0) {
      constexpr int KP = RBK / BPARTS;
      const float* ctile = reinterpret_cast<const float*>(tA) + (tid / BMT) * KP * BMT + tid % BMT;
#pragma unroll
      for (int kk = 0; kk < KP; ++kk) bsum += ctile[kk * BMT];
    }
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      float4 a0, a1, b0;
      if (RING_DBG & 16) {
        a0 = make_float4(1.f + kg, 2.f, 3.f, 4.f); a1 = a0; b0 = a0;
        asm volatile("" : "+v"(a0.x), "+v"(a1.x), "+v"(b0.x));
      } else {
        a0 = read_frag<A_ROW, BMT>(tA, fa0, kg);
        a1 = read_frag<A_ROW, BMT>(tA, fa1, kg);
        b0 = read_frag<B_ROW, BNT>(tB, fb0, kg);
      }
      if (RING_DBG & 4) {
        asm volatile("" ::"v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w), "v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w),
                     "v"(b0.x), "v"(b0.y), "v"(b0.z), "v"(b0.w));
        continue;
      }
      if (kg == 0 && kt == 0) {
        // a new tile starts from a zero C operand: the accumulator registers are never cleared by
        // a VALU write (such a write right behind the epilogue's stores, which may use the same
        // registers as store data, lost elements with two workgroups per CU)
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, zero, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, zero, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc1, 0, 0, 0);
      }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, acc1, 0, 0, 0);
    }
    cslot = cslot == NSLOT - 1 ? 0 : cslot + 1;
    if (++kt == c.nk) {
      if (DO_BIAS && g.bias_part != nullptr && c.tn == 0) {
        red[redbuf * 256 + tid] = bsum;   // read behind the next barrier
        bsum = 0.f;
        redbuf ^= 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      pending = true;
      pc = c;
      kt = 0;
      ct += w.stride;
      have = ct < w.end;
      if (have) c = decode_tile<GROUPED>(g, ct, BMT, BNT);
    }
  }
  if (EPI == EPI_MSE) {
    // one double per wave through LDS, fixed order (the loader joins the barrier)
    double* redd = reinterpret_cast<double*>(red);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off, 64);
    if (lane == 0) redd[wid] = lsum;
    lds_barrier();
    if (tid == 0) g.loss_partial[blockIdx.x] = (redd[0] + redd[1]) + (redd[2] + redd[3]);
  }
}

// WM = compute waves along the A (row) dimension: 2 -> 128 x 64 tile, 1 -> 64 x 128 tile.
// GROUPED: the tile order in groups of g.gn column tiles (decode_tile); a kernel of its own so that the
// plain order's code is untouched (as a run-time branch it cost the FF step 1 %, same box, same day).
template <bool A_ROW, bool B_ROW, int EPI, int WM, bool GROUPED = false>
__global__ __launch_bounds__(THREADS, (2 * THREADS + 255) / 256) void gemm_ring_kernel(Args g) {
  constexpr int BMT = 64 * WM, BNT = 32 * (4 / WM);
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_p)lds;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint64_t t0c = 0, t0r = 0;
  if (g.stamps) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
  if (wid >= 4) {
    loader_wave<A_ROW, B_ROW, BMT, BNT, GROUPED>(g, lds, lds0, lane, EPI == EPI_MSE, wid - 4);
  } else {
    compute_waves<A_ROW, B_ROW, EPI, BMT, BNT, GROUPED>(g, lds, wid, lane);
  }
  if (g.stamps && threadIdx.x == 0) {
    g.stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
    g.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
    g.stamps[4 * blockIdx.x + 2] = t0r;
  }
}

// Two independent GEMMs in one launch (the weight-gradient and the input-gradient product of a
// layer both start from dz): every persistent workgroup walks its tiles of the first problem, then
// its tiles of the second.  One launch boundary less per layer (ramp, drain and the end-of-kernel
// wait cost ~10 us each on this chip) and the workgroups that run out of tiles of the first
// problem start the second right away instead of idling to the end of the launch.
template <bool A1, bool B1, int E1, int W1, bool A2, bool B2, int E2, int W2>
__global__ __launch_bounds__(THREADS, (2 * THREADS + 255) / 256) void gemm_ring_pair_kernel(Args g1, Args g2) {
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_p)lds;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid >= 4) {
    loader_wave<A1, B1, 64 * W1, 32 * (4 / W1)>(g1, lds, lds0, lane, E1 == EPI_MSE, wid - 4);
    loader_wave<A2, B2, 64 * W2, 32 * (4 / W2)>(g2, lds, lds0, lane, E2 == EPI_MSE, wid - 4);
  } else {
    compute_waves<A1, B1, E1, 64 * W1, 32 * (4 / W1)>(g1, lds, wid, lane);
    compute_waves<A2, B2, E2, 64 * W2, 32 * (4 / W2)>(g2, lds, wid, lane);
  }
}

}  // namespace ring
}  // namespace itts
