// Dense acoustic-model kernels for gfx950: fp32 MFMA GEMM with fused epilogues, masked MSE,
// Adam.  Replaces the torch.nn.Linear / Tanh / MSELoss / torch.optim.Adam calls the reference
// makes per mini-batch (rnn_dyn/FFWrapper.py:63-73, loss/NamedLoss.py:70-117,
// ModularModelHandlerPyTorch.py:769-820).
//
// GEMM design (exact fp32: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak):
//   * 128x128 output tile per 256-thread workgroup, 4 waves as 2x2, each wave 64x64 =
//     2x2 MFMA tiles of 32x32 (64 accumulator VGPRs), K step 32, LDS double buffered
//     (73.7 KB -> 2 workgroups per CU); or a 128x64 tile with a single LDS stage (36.9 KB,
//     106 VGPRs -> 4 workgroups per CU) -- see launch_gemm for which shape runs when.
//   * an operand is either "row form" [out][k] (k contiguous, e.g. x[M,K], w[N,K]) or "col
//     form" [k][out] (e.g. dz[M,N] as the reduction-major operand of dW).  Row-form tiles are
//     copied to LDS unchanged with a 4-float pad (144-B rows: conflict-free ds_read_b128);
//     col-form tiles are [k][128+4] and read with ds_read_b32 (lanes = consecutive floats).
//   * K permutation instead of an LDS transpose: the MFMA takes k = lane>>5 from each lane;
//     lane half h feeds k = 8g + 4h + j on step j of k-group g, for A and B alike, so one
//     ds_read_b128 per lane supplies four MFMAs.
//   * roofline: MFMA fp32.  FLOPs per valid frame of the 425-512-512-187 model: fwd 1.15 M,
//     fwd+bwd 3.45 M (first-layer input gradient skipped: 3.01 M) -- SURVEY.md section 8d.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "gemm_ring.h"

namespace itts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;  // BN: widest tile (TN = 2)
constexpr int LD_ROW = BK + 4;    // row-form tile [128][36]
constexpr int TILE_FLOATS = 128 * LD_ROW;  // 4608 >= 32*132 = 4224

enum { EPI_STORE = 0, EPI_BIAS_ACT = 1, EPI_DACT = 2, EPI_MSE = 3 };

struct GemmArgs {
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  float* C;
  int64_t ldc;
  int64_t M;  // output rows
  int N;      // output cols
  int64_t K;  // reduction length
  const float* bias;
  const float* aux;
  int64_t ldaux;
  int act;
  int64_t kchunk;       // reduction elements per blockIdx.z (multiple of BK)
  int64_t slab_stride;  // floats between split-K slabs of C
  int vecA, vecB;       // 16-B vector loads allowed
  int wide_out;         // C (and bias, aux) allow 16-B accesses: float4 epilogue of the row-form kernel
  // EPI_MSE (last layer of a training step): C receives d loss / d output instead of the output
  const uint8_t* row_valid;   // [M]
  float gscale;               // 2 * loss_weight / (n_valid * D)
  double* loss_partial;       // [grid] sums of squared masked differences, one per workgroup
  // col-form A (weight gradients): column sums of A over this launch's K chunk, i.e. the bias
  // gradient, as a by-product of the workgroups of the first column tile
  float* bias_part;           // [slab][M] or NULL
  int64_t bias_part_stride;
};

// tanh in ~12 VALU ops (ocml tanhf costs ~40 and showed up as ~15 % of the fused-epilogue GEMMs):
// |z| < 0.25: odd Taylor polynomial up to z^9 (truncation < 9e-9 relative);
// else 1 - 2/(exp(2|z|)+1) with the hardware exp2/rcp (abs. error <= ~1.5e-7, i.e. ~2 ulp of
// the result in [0.24, 1]).  Max deviation from torch.tanh (fp32) observed: 2.4e-7.
__device__ __forceinline__ float fast_tanhf(float z) {
  const float a = fabsf(z);
  const float z2 = z * z;
  const float poly = z * (1.f + z2 * (-0.33333334f + z2 * (0.13333334f + z2 * (-0.053968254f +
                                                                              z2 * 0.021869488f))));
  const float e = __expf(2.f * a);
  const float big = copysignf(1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f), z);
  return a < 0.25f ? poly : big;
}

__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == ITTS_ACT_TANH) return fast_tanhf(z);
  if (act == ITTS_ACT_RELU) return z > 0.f ? z : 0.f;
  return z;
}
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
  if (act == ITTS_ACT_TANH) return 1.f - y * y;
  if (act == ITTS_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  return 1.f;
}

// Global -> registers for one 128(out) x 32(k) tile of an operand, 16 floats per thread.
// Branch-free: out-of-range elements read a clamped (valid) address and are zeroed by a select,
// so all loads of a tile are issued back to back (hipcc otherwise branches around every guarded
// load and waits for each one).  VEC: 16-byte loads; requires ld % 4 == 0 and a 16-B aligned base.
// When the contiguous extent (K for row form, out_dim for col form) is not a multiple of 4 the
// last float4 of a row also reads the 1-3 pad elements between the extent and the pitch: pad
// output columns are never stored, pad K elements meet a zero of the other operand -- so the pad
// only has to be finite (callers keep it zero; see the pitch rule in include/idiaptts_amd.h).
template <bool ROWFORM, bool VEC, int NROWS>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int64_t ld, int64_t out0,
                                          int64_t out_dim, int64_t k0, int64_t k_end,
                                          float4 (&r)[NROWS / 32]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NROWS / 32; ++i) {
    const int idx = tid + 256 * i;
    int64_t o, k;
    if (ROWFORM) {
      o = out0 + (idx >> 3);
      k = k0 + ((idx & 7) << 2);
    } else {
      k = k0 + idx / (NROWS / 4);
      o = out0 + ((idx % (NROWS / 4)) << 2);
    }
    // contiguous index c (4 consecutive elements), strided index t
    const int64_t c = ROWFORM ? k : o, c_end = ROWFORM ? k_end : out_dim;
    const int64_t t = ROWFORM ? o : k, t_end = ROWFORM ? out_dim : k_end;
    const bool t_ok = t < t_end;
    const int64_t tc = t_ok ? t : 0;
    if (VEC) {
      const bool ok = t_ok && (c < c_end);
      const float4 v = *reinterpret_cast<const float4*>(P + tc * ld + (ok ? c : 0));
      r[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const float* row = P + tc * ld;
      const bool k0ok = t_ok && c < c_end, k1ok = t_ok && c + 1 < c_end, k2ok = t_ok && c + 2 < c_end,
                 k3ok = t_ok && c + 3 < c_end;
      const float a0 = row[k0ok ? c : 0], a1 = row[k1ok ? c + 1 : 0], a2 = row[k2ok ? c + 2 : 0],
                  a3 = row[k3ok ? c + 3 : 0];
      r[i] = make_float4(k0ok ? a0 : 0.f, k1ok ? a1 : 0.f, k2ok ? a2 : 0.f, k3ok ? a3 : 0.f);
    }
  }
}

template <bool ROWFORM, int NROWS>
__device__ __forceinline__ void store_tile(float* __restrict__ S, const float4 (&r)[NROWS / 32]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NROWS / 32; ++i) {
    const int idx = tid + 256 * i;
    int off;
    if (ROWFORM)
      off = (idx >> 3) * LD_ROW + ((idx & 7) << 2);
    else
      off = (idx / (NROWS / 4)) * (NROWS + 4) + ((idx % (NROWS / 4)) << 2);
    *reinterpret_cast<float4*>(S + off) = r[i];
  }
}

// Fragment of k-group g for the 32 rows starting at `o` (tile-local): 4 k values per lane.
template <bool ROWFORM, int NROWS>
__device__ __forceinline__ float4 read_frag(const float* __restrict__ S, int o, int g, int lane) {
  const int r = lane & 31, h = lane >> 5;
  if (ROWFORM) {
    return *reinterpret_cast<const float4*>(S + (o + r) * LD_ROW + g * 8 + 4 * h);
  } else {
    constexpr int LDC = NROWS + 4;
    const float* p = S + (g * 8 + 4 * h) * LDC + o + r;
    return make_float4(p[0], p[LDC], p[2 * LDC], p[3 * LDC]);
  }
}

// TN = MFMA tiles per wave along N: output tile 128 x (64*TN). TN = 1 halves the tile so that
// narrow outputs (N = 187) and awkward tile counts waste fewer workgroup slots.
template <bool A_ROW, bool B_ROW, int EPI, bool VEC_A, bool VEC_B, int TN, int STAGES = 2>
__global__ __launch_bounds__(256, STAGES == 1 ? 4 : 2) void gemm_f32_kernel(GemmArgs g) {
  constexpr int BNT = 64 * TN;
  constexpr int B_FLOATS = STAGES == 1 ? (B_ROW ? BNT * LD_ROW : BK * (BNT + 4)) : TILE_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[STAGES * 2 * TILE_FLOATS - (STAGES == 1 ? TILE_FLOATS - B_FLOATS : 0)];
  // buffer b: A tile at lds + 2b*TILE, B tile at lds + (2b+1)*TILE.  STAGES == 1: one LDS buffer
  // (36.9 KB -> 4 workgroups per CU), the next tile waits in registers and two barriers per K
  // tile separate its store from the reads of the current one.

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each
  // XCD a contiguous run of tiles that share the same B panel (weights) where possible.
  const int tiles_n = (g.N + BNT - 1) / BNT;
  const int64_t tiles_m = (g.M + BM - 1) / BM;
  const int64_t ntiles = tiles_m * tiles_n;
  int64_t bid = blockIdx.x;
  {
    const int64_t q = ntiles / 8, r = ntiles % 8;
    const int64_t xcd = bid % 8, pos = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  const int64_t tm = bid / tiles_n;
  const int tn = (int)(bid % tiles_n);
  const int64_t m0 = tm * BM;
  const int n0 = tn * BNT;

  const int64_t kbeg = (int64_t)blockIdx.z * g.kchunk;
  const int64_t kend = std::min<int64_t>(g.K, kbeg + g.kchunk);
  const int64_t nkt = (kend - kbeg + BK - 1) / BK;

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool do_bias = !A_ROW && g.bias_part != nullptr && tn == 0;
  float bsum = 0.f;
  float4 ra[4], rb[BNT / 32];
  if (nkt > 0) {
    load_tile<A_ROW, VEC_A, BM>(g.A, g.lda, m0, g.M, kbeg, kend, ra);
    load_tile<B_ROW, VEC_B, BNT>(g.B, g.ldb, n0, g.N, kbeg, kend, rb);
    store_tile<A_ROW, BM>(lds, ra);
    store_tile<B_ROW, BNT>(lds + TILE_FLOATS, rb);
  }
  __syncthreads();

  for (int64_t kt = 0; kt < nkt; ++kt) {
    const int cur = STAGES == 1 ? 0 : (int)(kt & 1);
    const bool more = kt + 1 < nkt;
    if (more) {
      load_tile<A_ROW, VEC_A, BM>(g.A, g.lda, m0, g.M, kbeg + (kt + 1) * BK, kend, ra);
      load_tile<B_ROW, VEC_B, BNT>(g.B, g.ldb, n0, g.N, kbeg + (kt + 1) * BK, kend, rb);
    }
    const float* cA = lds + (2 * cur) * TILE_FLOATS;
    const float* cB = lds + (2 * cur + 1) * TILE_FLOATS;
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float4 fa[2], fb[TN];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = read_frag<A_ROW, BM>(cA, wm * 64 + i * 32, kg, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = read_frag<B_ROW, BNT>(cB, wn * 32 * TN + j * 32, kg, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (!A_ROW && do_bias) {   // column sums of the A tile ([k][out], pitch BM + 4) while it is resident
      const int o = threadIdx.x & 127, kh = threadIdx.x >> 7;
      const float* ct = cA + (kh * (BK / 2)) * (BM + 4) + o;
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) bsum += ct[kk * (BM + 4)];
    }
    if (STAGES == 1) __syncthreads();   // everyone has read the current tile
    if (more) {
      constexpr int nb = STAGES == 1 ? 0 : 1;
      store_tile<A_ROW, BM>(lds + (2 * (cur ^ nb)) * TILE_FLOATS, ra);
      store_tile<B_ROW, BNT>(lds + (2 * (cur ^ nb) + 1) * TILE_FLOATS, rb);
    }
    __syncthreads();
  }

  if (!A_ROW && do_bias) {   // the two k halves meet in LDS (all tile reads are behind the loop's last barrier)
    const int o = threadIdx.x & 127, kh = threadIdx.x >> 7;
    if (kh == 1) lds[o] = bsum;
    __syncthreads();
    if (kh == 0 && m0 + o < g.M)
      g.bias_part[(int64_t)blockIdx.z * g.bias_part_stride + m0 + o] = bsum + lds[o];
  }
  // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* C = g.C + (int64_t)blockIdx.z * g.slab_stride;
  const int cl = lane & 31, rh = lane >> 5;
  if (EPI == EPI_MSE) {
    // Last layer of a training step: y = acc + bias never goes to memory; the masked difference to
    // the target (aux) is squared into the workgroup's partial sum and scaled into C as
    // d loss / d y (masked_mse_kernel's arithmetic: double accumulation of float differences).
    float* stage = lds + wid * (32 * 36);
    const int c4 = (lane & 7) << 2, rq = lane >> 3;
    const int colb = n0 + wn * 32;
    double lsum = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i > 0) {   // the reads of the previous block are done before it is overwritten
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * rh) * 36 + cl] = acc[i][0][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the lanes of this wave exchange the block through LDS
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int col = colb + c4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rl = rq + 8 * q;
        const int64_t row = m0 + wm * 64 + i * 32 + rl;
        const float4 v = *reinterpret_cast<const float4*>(stage + rl * 36 + c4);
        if (row >= g.M || col >= g.N) continue;
        const bool ok = g.row_valid[row] != 0;
        const float vv[4] = {v.x, v.y, v.z, v.w};
        float dz[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float diff = 0.f;
          if (col + e < g.N && ok)
            diff = (vv[e] + (g.bias ? g.bias[col + e] : 0.f)) - g.aux[row * g.ldaux + col + e];
          lsum += (double)diff * (double)diff;
          dz[e] = g.gscale * diff;
        }
        if (col + 3 < g.N) {
          *reinterpret_cast<float4*>(C + row * g.ldc + col) = make_float4(dz[0], dz[1], dz[2], dz[3]);
        } else {
          for (int e = 0; e < 4 && col + e < g.N; ++e) C[row * g.ldc + col + e] = dz[e];
        }
      }
    }
    __shared__ double mse_red[16];
    lsum = block_sum(lsum, mse_red);
    if (threadIdx.x == 0) g.loss_partial[blockIdx.x] = lsum;
    return;
  }
  if (STAGES == 1 && TN == 1 && g.wide_out) {
    // Wide stores: a wave passes each of its 32 x 32 result blocks through its own 4.6 KB of the
    // (now idle) LDS tile and writes rows back as float4 -- 8 store instructions of whole 128-byte
    // row segments per thread instead of 32 dword stores; bias / activation / derivative are applied
    // four columns at a time on the way out.  (The loop's last barrier has retired all tile reads;
    // the staging area is private to the wave, LDS operations of a wave execute in order.)
    float* stage = lds + wid * (32 * 36);
    const int c4 = (lane & 7) << 2, rq = lane >> 3;
    const int colb = n0 + wn * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i > 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * rh) * 36 + cl] = acc[i][0][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the lanes of this wave exchange the block through LDS
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int col = colb + c4;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool full = col + 3 < g.N;
      if (EPI == EPI_BIAS_ACT && g.bias && full) bv = *reinterpret_cast<const float4*>(g.bias + col);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rl = rq + 8 * q;
        const int64_t row = m0 + wm * 64 + i * 32 + rl;
        float4 v = *reinterpret_cast<const float4*>(stage + rl * 36 + c4);
        if (row >= g.M || col >= g.N) continue;
        if (full) {
          if (EPI == EPI_BIAS_ACT) {
            v.x = act_fwd(v.x + bv.x, g.act); v.y = act_fwd(v.y + bv.y, g.act);
            v.z = act_fwd(v.z + bv.z, g.act); v.w = act_fwd(v.w + bv.w, g.act);
          }
          if (EPI == EPI_DACT) {
            const float4 a = *reinterpret_cast<const float4*>(g.aux + row * g.ldaux + col);
            v.x *= act_grad_from_out(a.x, g.act); v.y *= act_grad_from_out(a.y, g.act);
            v.z *= act_grad_from_out(a.z, g.act); v.w *= act_grad_from_out(a.w, g.act);
          }
          *reinterpret_cast<float4*>(C + row * g.ldc + col) = v;
        } else {   // the float4 that straddles N: column by column
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int e = 0; e < 4 && col + e < g.N; ++e) {
            float o = vv[e];
            if (EPI == EPI_BIAS_ACT) o = act_fwd(o + (g.bias ? g.bias[col + e] : 0.f), g.act);
            if (EPI == EPI_DACT) o *= act_grad_from_out(g.aux[row * g.ldaux + col + e], g.act);
            C[row * g.ldc + col + e] = o;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + j * 32 + cl;
      if (col >= g.N) continue;
      float bv = 0.f;
      if (EPI == EPI_BIAS_ACT && g.bias) bv = g.bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rh;
        if (row >= g.M) continue;
        float v = acc[i][j][r];
        if (EPI == EPI_BIAS_ACT) v = act_fwd(v + bv, g.act);
        if (EPI == EPI_DACT) v *= act_grad_from_out(g.aux[row * g.ldaux + col], g.act);
        C[row * g.ldc + col] = v;
      }
    }
}

template <bool A_ROW, bool B_ROW, int EPI, int TN, int STAGES>
static int launch_gemm_tn(const GemmArgs& g, int splitk, hipStream_t s) {
  const int64_t tiles = ((g.M + BM - 1) / BM) * ((g.N + 64 * TN - 1) / (64 * TN));
  dim3 grid((unsigned)tiles, 1, (unsigned)splitk);
  const bool va = g.vecA, vb = g.vecB;   // pitch and alignment allow 16-byte loads (see load_tile)
  if (va && vb)
    hipLaunchKernelGGL((gemm_f32_kernel<A_ROW, B_ROW, EPI, true, true, TN, STAGES>), grid, dim3(256), 0, s, g);
  else if (va)
    hipLaunchKernelGGL((gemm_f32_kernel<A_ROW, B_ROW, EPI, true, false, TN, STAGES>), grid, dim3(256), 0, s, g);
  else if (vb)
    hipLaunchKernelGGL((gemm_f32_kernel<A_ROW, B_ROW, EPI, false, true, TN, STAGES>), grid, dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<A_ROW, B_ROW, EPI, false, false, TN, STAGES>), grid, dim3(256), 0, s, g);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

// Kernel shape choice.  Measured on the acoustic-model GEMMs (MI355X, MfmaUtil from rocprofv3):
//   * row-form A (forward and input-gradient GEMMs: M = frames, short K): 128x64 tiles with ONE LDS
//     stage -- 37 KB and 106 VGPRs, so 4 workgroups per CU instead of 2.  Twice the waves per SIMD
//     cover the barrier / LDS-refill gaps of a K tile: +21 % per-tile rate over the double-buffered
//     128x128 kernel although it needs two barriers per K tile.
//   * col-form A (weight-gradient GEMMs, split-K, long K loops): double-buffered tiles; 128x128
//     reuses operands best, 128x64 quantises better onto the 512 resident slots (256 CUs x 2) and
//     wastes less on narrow outputs -- pick by the slot-quantisation estimate below.  (The
//     single-stage variant spills in col form and is slower.)
static double tile_efficiency(int64_t M, int N, int splitk, int tn, double loop_eff) {
  const int bn = 64 * tn;
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + bn - 1) / bn) * splitk;
  const int64_t rounds = (tiles + 511) / 512;
  const double useful = (double)N / (double)(((N + bn - 1) / bn) * bn);
  return loop_eff * useful * (double)tiles / (double)(rounds * 512);
}

static inline int aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- the LDS-DMA ring kernel (gemm_ring.h) -----------------------------------------------------
// Takes every call whose operands allow 16-byte row accesses (the trainer's buffers all do); the
// register-staged kernel above stays the general path (odd pitches, unaligned bases).
constexpr int kRingGrid = 512;   // persistent workgroups: 2 per CU

template <bool A_ROW, bool B_ROW>
static bool ring_ok(const GemmArgs& g, int splitk, int epi) {
  if (!g.vecA || !g.vecB) return false;
  if (g.lda % 4 || g.ldb % 4 || g.ldc % 4 || g.slab_stride % 4 || !aligned16(g.C)) return false;
  if (epi == EPI_DACT && (g.ldaux % 4 || !aligned16(g.aux))) return false;   // the MSE target is read dword-wise
  const int64_t lim = (int64_t)1 << 30;   // 32-bit byte offsets inside the buffer descriptors
  if (g.M >= lim || g.K >= lim || g.lda >= lim / 256 || g.ldb >= lim / 256 || g.ldc >= lim / 256 ||
      g.ldaux >= lim / 256)
    return false;
  if (!A_ROW && (g.kchunk + 32) * g.lda >= lim) return false;
  if (!B_ROW && (g.kchunk + 32) * g.ldb >= lim) return false;
  if (splitk > 1 && g.kchunk % 32) return false;
  return true;
}

static int ring_group(int tiles_n, int bnt, int64_t K, bool row_row);

template <bool A_ROW, bool B_ROW, int EPI, int WM>
static int launch_ring_wm(const GemmArgs& g, int splitk, hipStream_t s) {
  constexpr int BMT = 64 * WM, BNT = 32 * (4 / WM);
  ring::Args r{};
  r.A = g.A; r.B = g.B; r.C = g.C; r.bias = g.bias; r.aux = g.aux;
  r.row_valid = g.row_valid; r.loss_partial = g.loss_partial; r.bias_part = g.bias_part;
  r.slab_stride = g.slab_stride; r.bias_part_stride = g.bias_part_stride;
  r.lda = (int)g.lda; r.ldb = (int)g.ldb; r.ldc = (int)g.ldc; r.ldaux = (int)g.ldaux;
  r.M = (int)g.M; r.N = g.N; r.K = (int)g.K; r.kchunk = (int)g.kchunk; r.splitk = splitk;
  r.tiles_m = (int)((g.M + BMT - 1) / BMT);
  r.tiles_n = (g.N + BNT - 1) / BNT;
  r.gn = ring_group(r.tiles_n, BNT, g.K, A_ROW && B_ROW);
  r.act = g.act; r.gscale = g.gscale;
  const int64_t ntiles = (int64_t)r.tiles_m * r.tiles_n * splitk;
  ITTS_REQUIRE(ntiles < ((int64_t)1 << 31), "too many tiles");
  const int grid = (int)std::min<int64_t>(kRingGrid, (ntiles + 7) / 8 * 8);
  if (A_ROW && B_ROW && EPI == ring::EPI_BIAS_ACT && r.gn < r.tiles_n)
    hipLaunchKernelGGL((ring::gemm_ring_kernel<A_ROW, B_ROW, EPI, WM, A_ROW && B_ROW && EPI == ring::EPI_BIAS_ACT>),
                       dim3(grid), dim3(ring::THREADS), 0, s, r);
  else
    hipLaunchKernelGGL((ring::gemm_ring_kernel<A_ROW, B_ROW, EPI, WM>), dim3(grid), dim3(ring::THREADS), 0, s, r);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

// Column tiles per group of the tile order (ring::decode_tile).  Forward products (row x row) whose weight
// matrix is larger than an XCD's L2 -- the recurrent layers' [4096 x 1024] input projections, 16 MB -- walk
// groups of ~4 MB of weight panels (4.78 -> 4.54 ms for 73 138 rows, scripts/ring_order_probe.py); every
// other product keeps one group (the weight-gradient products measured 2-3 % slower in groups).
static int ring_group(int tiles_n, int bnt, int64_t K, bool row_row) {
  if (!row_row) return tiles_n;
  const int64_t panel = (int64_t)bnt * K * 4;
  const int64_t gn = std::max<int64_t>(1, ((int64_t)4 << 20) / std::max<int64_t>(panel, 1));
  return (int)std::min<int64_t>(gn, tiles_n);
}

// tile shape of the ring kernel for an [M x N] output: 128 x 64 or 64 x 128, whichever wastes less
static int ring_wm(int64_t M, int N) {
  const double w2 = (double)(((M + 127) / 128) * 128) * (double)(((N + 63) / 64) * 64);
  const double w1 = (double)(((M + 63) / 64) * 64) * (double)(((N + 127) / 128) * 128);
  return w1 < w2 ? 1 : 2;
}

template <bool A_ROW, bool B_ROW, int EPI>
static int launch_ring(const GemmArgs& g, int splitk, hipStream_t s) {
  if (ring_wm(g.M, g.N) == 1) return launch_ring_wm<A_ROW, B_ROW, EPI, 1>(g, splitk, s);
  return launch_ring_wm<A_ROW, B_ROW, EPI, 2>(g, splitk, s);
}

template <int WM>
static ring::Args ring_args(const GemmArgs& g, int splitk) {
  constexpr int BMT = 64 * WM, BNT = 32 * (4 / WM);
  ring::Args r{};
  r.A = g.A; r.B = g.B; r.C = g.C; r.bias = g.bias; r.aux = g.aux;
  r.row_valid = g.row_valid; r.loss_partial = g.loss_partial; r.bias_part = g.bias_part;
  r.slab_stride = g.slab_stride; r.bias_part_stride = g.bias_part_stride;
  r.lda = (int)g.lda; r.ldb = (int)g.ldb; r.ldc = (int)g.ldc; r.ldaux = (int)g.ldaux;
  r.M = (int)g.M; r.N = g.N; r.K = (int)g.K; r.kchunk = (int)g.kchunk; r.splitk = splitk;
  r.tiles_m = (int)((g.M + BMT - 1) / BMT);
  r.tiles_n = (g.N + BNT - 1) / BNT;
  r.gn = r.tiles_n;
  r.act = g.act; r.gscale = g.gscale;
  return r;
}

// weight-gradient GEMM (col x col, split-K slabs) and input-gradient GEMM (row x col, activation
// derivative in the epilogue) of one layer in ONE launch
template <int WM_W, int EPI_X>
static int launch_ring_bwd_pair(const GemmArgs& gw, int splitk, const GemmArgs& gx, hipStream_t s) {
  const ring::Args rw = ring_args<WM_W>(gw, splitk), rx = ring_args<2>(gx, 1);
  hipLaunchKernelGGL((ring::gemm_ring_pair_kernel<false, false, EPI_STORE, WM_W, true, false, EPI_X, 2>),
                     dim3(kRingGrid), dim3(ring::THREADS), 0, s, rw, rx);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

template <bool A_ROW, bool B_ROW, int EPI>
static int launch_gemm(GemmArgs g, int splitk, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0) return ITTS_OK;
  if (ring_ok<A_ROW, B_ROW>(g, splitk, EPI)) return launch_ring<A_ROW, B_ROW, EPI>(g, splitk, s);
  if (A_ROW) return launch_gemm_tn<A_ROW, B_ROW, EPI, 1, 1>(g, splitk, s);
  const double e2 = tile_efficiency(g.M, g.N, splitk, 2, 1.0);
  const double e1 = tile_efficiency(g.M, g.N, splitk, 1, 0.90);
  if (e1 > e2) return launch_gemm_tn<A_ROW, B_ROW, EPI, 1, 2>(g, splitk, s);
  return launch_gemm_tn<A_ROW, B_ROW, EPI, 2, 2>(g, splitk, s);
}


// ---- split-K slab reduction (deterministic order) -------------------------------------------
// out[i] = sum_z slabs[z][i].  Memory bound (S x n floats in, n out); a workgroup owns 64 float4
// columns, its 4 waves take every 4th slab each (all loads of a thread independent -> in flight
// together) and the four partial sums meet in LDS in a fixed order.  n % 4 == 0 and 16-byte
// aligned buffers take this path, anything else the scalar tail kernel.
__device__ __forceinline__ void reduce_slabs_vec_body(const float* __restrict__ slabs, int S, int64_t n,
                                                      float* __restrict__ out, int accumulate, int64_t block,
                                                      float4 (*part)[64]) {
  const int tx = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t n4 = n >> 2;
  const int64_t c = block * 64 + tx;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < n4) {
    const float4* p = reinterpret_cast<const float4*>(slabs) + c;
#pragma unroll 8
    for (int z = grp; z < S; z += 4) {
      const float4 v = p[(int64_t)z * n4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  if (grp > 0) part[grp - 1][tx] = s;
  __syncthreads();
  if (grp == 0 && c < n4) {
#pragma unroll
    for (int g2 = 0; g2 < 3; ++g2) {
      const float4 v = part[g2][tx];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4* o = reinterpret_cast<float4*>(out) + c;
    if (accumulate) {
      const float4 v = *o;
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *o = s;
  }
}

// same split of the work for buffers that cannot be read as float4 (n % 4 != 0, e.g. 187 biases)
__device__ __forceinline__ void reduce_slabs_scalar_body(const float* __restrict__ slabs, int S, int64_t n,
                                                         float* __restrict__ out, int accumulate,
                                                         int64_t block, float (*part)[64]) {
  const int tx = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t c = block * 64 + tx;
  float s = 0.f;
  if (c < n) {
#pragma unroll 8
    for (int z = grp; z < S; z += 4) s += slabs[(int64_t)z * n + c];
  }
  if (grp > 0) part[grp - 1][tx] = s;
  __syncthreads();
  if (grp == 0 && c < n) {
    s += part[0][tx];
    s += part[1][tx];
    s += part[2][tx];
    out[c] = accumulate ? out[c] + s : s;
  }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int S,
                                                           int64_t n, float* __restrict__ out,
                                                           int accumulate) {
  __shared__ float4 part[3][64];
  reduce_slabs_vec_body(slabs, S, n, out, accumulate, blockIdx.x, part);
}

__global__ __launch_bounds__(256) void reduce_slabs_scalar_kernel(const float* __restrict__ slabs, int S,
                                                                  int64_t n, float* __restrict__ out,
                                                                  int accumulate) {
  __shared__ float part[3][64];
  reduce_slabs_scalar_body(slabs, S, n, out, accumulate, blockIdx.x, part);
}

// Several reductions in ONE launch (itts_defer_reductions / itts_reduce_deferred): a training step
// owes one slab reduction per layer (plus odd-sized bias vectors and the loss partial sums), each a
// 5 us launch of its own that the next GEMM then waits for; queued, they run as segments of one
// grid.  Same arithmetic per element as the single kernels above (bit-identical results).
constexpr int kMaxDeferred = 16;
struct ReduceSeg {
  const void* src;      // slabs (float) or partial sums (double, kind 2)
  void* out;
  int64_t n;            // floats per slab / number of partial sums
  double scale;         // kind 2
  int S;
  int accumulate;
  int kind;             // 0: float4 columns, 1: scalar columns, 2: sum of doubles -> one float
  int first_block;
};
struct ReduceMulti {
  ReduceSeg seg[kMaxDeferred];
  int nseg;
};

__global__ __launch_bounds__(256) void reduce_multi_kernel(ReduceMulti d) {
  __shared__ float4 part4[3][64];
  __shared__ double red[16];
  int i = 0;
  while (i + 1 < d.nseg && (int)blockIdx.x >= d.seg[i + 1].first_block) ++i;
  const ReduceSeg& g = d.seg[i];
  const int64_t block = (int64_t)blockIdx.x - g.first_block;
  if (g.kind == 0) {
    reduce_slabs_vec_body(static_cast<const float*>(g.src), g.S, g.n, static_cast<float*>(g.out), g.accumulate,
                          block, part4);
  } else if (g.kind == 1) {
    reduce_slabs_scalar_body(static_cast<const float*>(g.src), g.S, g.n, static_cast<float*>(g.out),
                             g.accumulate, block, reinterpret_cast<float (*)[64]>(part4));
  } else {
    const double* partial = static_cast<const double*>(g.src);
    double s = 0.0;
    for (int k = threadIdx.x; k < (int)g.n; k += 256) s += partial[k];   // fixed order: deterministic
    s = block_sum(s, red);
    if (threadIdx.x == 0) *static_cast<float*>(g.out) = (float)(s * g.scale);
  }
}

static thread_local bool t_defer = false;
static thread_local ReduceMulti t_pending{};

static int flush_deferred(hipStream_t s) {
  if (t_pending.nseg == 0) return ITTS_OK;
  const ReduceSeg& last = t_pending.seg[t_pending.nseg - 1];
  const int64_t per = last.kind == 0 ? (last.n / 4 + 63) / 64 : (last.kind == 1 ? (last.n + 63) / 64 : 1);
  const int64_t blocks = last.first_block + per;
  hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, s, t_pending);
  t_pending.nseg = 0;
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

static int queue_deferred(ReduceSeg g, hipStream_t s) {
  if (t_pending.nseg == kMaxDeferred) {
    const int rc = flush_deferred(s);
    if (rc) return rc;
  }
  int64_t first = 0;
  if (t_pending.nseg > 0) {
    const ReduceSeg& p = t_pending.seg[t_pending.nseg - 1];
    first = p.first_block + (p.kind == 0 ? (p.n / 4 + 63) / 64 : (p.kind == 1 ? (p.n + 63) / 64 : 1));
  }
  ITTS_REQUIRE(first < ((int64_t)1 << 30), "too many deferred reduction blocks");
  g.first_block = (int)first;
  t_pending.seg[t_pending.nseg++] = g;
  return ITTS_OK;
}

static int launch_reduce_slabs(const float* slabs, int S, int64_t n, float* out, int accumulate,
                               hipStream_t s) {
  const bool vec = n % 4 == 0 && aligned16(slabs) && aligned16(out);
  if (t_defer) {
    ReduceSeg g{};
    g.src = slabs; g.out = out; g.n = n; g.S = S; g.accumulate = accumulate; g.kind = vec ? 0 : 1;
    return queue_deferred(g, s);
  }
  if (vec)
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, s, slabs, S,
                       n, out, accumulate);
  else
    hipLaunchKernelGGL(reduce_slabs_scalar_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, slabs,
                       S, n, out, accumulate);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_defer_reductions(int on) {
  ITTS_REQUIRE(on || t_pending.nseg == 0,
               "queued reductions must be run (itts_reduce_deferred) before deferral is switched off");
  t_defer = on != 0;
  return ITTS_OK;
}

extern "C" int itts_reduce_deferred(void* stream) {
  t_defer = false;
  return flush_deferred(as_stream(stream));
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                               float* __restrict__ dz, int64_t n, int act) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dz[i] = dy[i] * act_grad_from_out(y[i], act);
}

// ---- masked MSE -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void masked_mse_kernel(
    const float* __restrict__ pred, int64_t ldp, const float* __restrict__ target, int64_t ldt,
    const uint8_t* __restrict__ valid, int64_t M, int D, float gscale, float* __restrict__ grad,
    int64_t ldg, double* __restrict__ partial) {
  __shared__ double red[16];
  const int64_t n = M * D;
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    const int c = (int)(i - r * D);
    float diff = 0.f;
    if (valid[r]) diff = pred[r * ldp + c] - target[r * ldt + c];
    s += (double)diff * (double)diff;
    if (grad) grad[r * ldg + c] = gscale * diff;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void masked_mse_final_kernel(const double* __restrict__ partial,
                                                               int nb, double scale,
                                                               float* __restrict__ loss) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];  // fixed order: deterministic
  s = block_sum(s, red);
  if (threadIdx.x == 0) *loss = (float)(s * scale);
}

static int launch_mse_final(const double* partial, int nb, double scale, float* loss, hipStream_t s) {
  if (t_defer) {
    ReduceSeg g{};
    g.src = partial; g.out = loss; g.n = nb; g.scale = scale; g.S = 1; g.kind = 2;
    return queue_deferred(g, s);
  }
  hipLaunchKernelGGL(masked_mse_final_kernel, dim3(1), dim3(256), 0, s, partial, nb, scale, loss);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}


// ---- row-weighted elementwise losses (the other NamedLoss types / reductions) ------------------
// loss = sum_r w[r] sum_c e(pred - target), e = squared (kind 0, MSELoss) or absolute (kind 1,
// L1Loss) error; grad = w[r] e'(.) ; elem (optional) = w[r] e(.) per element (reduction 'none').
// The sequence mask and every reduction of NamedLoss._reduce (loss/NamedLoss.py:113-131) are a
// per-row weight: mean_per_frame mask / (frames D), mean_per_sample mask / (len_b B D),
// mean mask / (T B D), sum / none mask.
__global__ __launch_bounds__(256) void weighted_loss_kernel(
    const float* __restrict__ pred, int64_t ldp, const float* __restrict__ target, int64_t ldt,
    const float* __restrict__ w, int64_t M, int D, int kind, float* __restrict__ grad, int64_t ldg,
    float* __restrict__ elem, int64_t lde, double* __restrict__ partial) {
  __shared__ double red[16];
  const int64_t n = M * D;
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    const int c = (int)(i - r * D);
    const float wr = w[r];
    const float diff = pred[r * ldp + c] - target[r * ldt + c];
    const float e = kind == 0 ? diff * diff : fabsf(diff);
    const float g = kind == 0 ? 2.f * diff : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
    const float v = wr == 0.f ? 0.f : wr * e;          // padding may hold anything (also NaN)
    s += (double)v;
    if (grad) grad[r * ldg + c] = wr == 0.f ? 0.f : wr * g;
    if (elem) elem[r * lde + c] = v;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// ---- Adam ---------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, int64_t n, float beta1,
                            float beta2, float eps, float wd, float step_size, float bc2_sqrt,
                            float gscale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.f - beta1);         // exp_avg.lerp_(grad, 1-beta1)
    vi = vi * beta2 + (1.f - beta2) * gi * gi;   // mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// Adam with the gradient clipping in front and the parameter EMA behind it, one pass.
__global__ void adam_fused_kernel(float* __restrict__ p, const float* __restrict__ g,
                                  float* __restrict__ m, float* __restrict__ v, int64_t n,
                                  float beta1, float beta2, float eps, float wd, float step_size,
                                  float bc2_sqrt, float gscale, const float* __restrict__ norm_accum,
                                  int norm_kind, float max_norm, float clip_value,
                                  float* __restrict__ shadow, float one_minus_decay) {
  float coef = gscale;
  if (norm_accum) {
    const float a = norm_accum[0];
    const float total = (norm_kind == 2 ? sqrtf(a) : a) * fabsf(gscale);
    coef *= fminf(1.f, max_norm / (total + 1e-6f));
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i] * coef;
    if (clip_value > 0.f) gi = fminf(fmaxf(gi, -clip_value), clip_value);
    float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.f - beta1);
    vi = vi * beta2 + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (shadow) {
      const float si = shadow[i];
      shadow[i] = si - one_minus_decay * (si - pi);
    }
  }
}

// sum x^2 / max |x| of a flat buffer: fixed grid of partials, then one block (deterministic)
constexpr int kNormBlocks = 1024;
__global__ __launch_bounds__(256) void grad_norm_partial_kernel(const float* __restrict__ x, int64_t n,
                                                               int kind, float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float xi = x[i];
    acc = kind == 2 ? acc + xi * xi : fmaxf(acc, fabsf(xi));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(acc, off, 64);
    acc = kind == 2 ? acc + o : fmaxf(acc, o);
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    part[blockIdx.x] = kind == 2 ? (red[0] + red[1]) + (red[2] + red[3])
                                 : fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void grad_norm_final_kernel(const float* __restrict__ part, int nb,
                                                             int kind, float* __restrict__ out,
                                                             int accumulate) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc = kind == 2 ? acc + part[i] : fmaxf(acc, part[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(acc, off, 64);
    acc = kind == 2 ? acc + o : fmaxf(acc, o);
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = kind == 2 ? (red[0] + red[1]) + (red[2] + red[3])
                        : fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (accumulate) r = kind == 2 ? out[0] + r : fmaxf(out[0], r);
    out[0] = r;
  }
}

// ---- SGD (torch.optim.SGD: weight decay, momentum, dampening, Nesterov) -------------------------
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                           float* __restrict__ buf, int64_t n, float lr, float momentum,
                           float dampening, float wd, int nesterov, int first, float gscale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    if (momentum != 0.f) {
      float bi = first ? gi : buf[i] * momentum + (1.f - dampening) * gi;
      buf[i] = bi;
      gi = nesterov ? gi + momentum * bi : bi;
    }
    p[i] = pi - lr * gi;
  }
}

// ---- exponential moving average of the parameters: shadow -= (1-decay) * (shadow - x) ---------
__global__ void ema_kernel(float* __restrict__ shadow, const float* __restrict__ x, int64_t n,
                           float one_minus_decay) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float sh = shadow[i];
    shadow[i] = sh - one_minus_decay * (sh - x[i]);
  }
}

constexpr int kColsumSlices = 256;  // row slices of the bias-gradient column sums

// split-K of a weight-gradient GEMM on the ring kernel: as many slabs as keep the 512 persistent
// workgroups busy, chunks of whole K-steps
static int choose_splitk_ring(int64_t M, int N, int K) {
  const int wm = ring_wm(N, K);
  const int bmt = 64 * wm, bnt = 32 * (4 / wm);
  const int64_t tiles = (int64_t)((N + bmt - 1) / bmt) * ((K + bnt - 1) / bnt);
  int64_t s = std::max<int64_t>(1, kRingGrid / std::max<int64_t>(tiles, 1));
  s = std::min<int64_t>(s, std::max<int64_t>(1, (M + 255) / 256));
  return (int)std::min<int64_t>(s, 128);
}

static int choose_splitk(int64_t M, int N, int K) {
  const int64_t tiles = (int64_t)((N + BM - 1) / BM) * ((K + BN - 1) / BN);
  int64_t s = std::max<int64_t>(1, 512 / std::max<int64_t>(tiles, 1));
  const int64_t max_by_rows = std::max<int64_t>(1, (M + 511) / 512);
  s = std::min(s, max_by_rows);
  return (int)std::min<int64_t>(s, 128);
}

}  // namespace itts

using namespace itts;

extern "C" int itts_linear_fwd(const float* d_x, int64_t ldx, const float* d_w, const float* d_b,
                               float* d_y, int64_t ldy, int64_t M, int N, int K, int act,
                               void* stream) {
  ITTS_REQUIRE(d_w && (M == 0 || (d_x && d_y)), "null pointer");
  ITTS_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K && ldy >= N, "bad sizes");
  ITTS_REQUIRE(act >= 0 && act <= 2, "unknown activation");
  if (M == 0) return ITTS_OK;
  GemmArgs g{};
  g.A = d_x; g.lda = ldx; g.B = d_w; g.ldb = K; g.C = d_y; g.ldc = ldy;
  g.M = M; g.N = N; g.K = K; g.bias = d_b; g.act = act;
  g.kchunk = ((K + BK - 1) / BK) * BK; g.slab_stride = 0;
  g.vecA = (ldx % 4 == 0) && aligned16(d_x);
  g.vecB = (K % 4 == 0) && aligned16(d_w);
  g.wide_out = (ldy % 4 == 0) && aligned16(d_y) && (!d_b || aligned16(d_b));
  return launch_gemm<true, true, EPI_BIAS_ACT>(g, 1, as_stream(stream));
}

// ---- row gather: pack_padded_sequence / pad_packed_sequence / the h_{t-1} shift of the recurrent layers
// dst row r = src row idx[r] (its first `width` columns), the fill row (or zeros) where idx[r] is outside
// [0, n_src); columns width .. dst_width - 1 of every dst row are zeroed (row pitches of 16 bytes for the
// GEMMs).  One kernel for what the reference leaves to torch.nn.utils.rnn (rnn_dyn/RNNWrapper.py:89-102)
// and autograd: index_select + pad on the way in, index_copy into zeros on the way out, cat + index_select
// for the shifted state.
__global__ __launch_bounds__(256) void rows_gather_kernel(const float* __restrict__ src, int64_t ld_src, int64_t n_src,
                                                          const int64_t* __restrict__ idx, int64_t n_out, int width,
                                                          const float* __restrict__ fill, float* __restrict__ dst,
                                                          int64_t ld_dst, int dst_width, int vec) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), wn = (int64_t)gridDim.x * 4;
  for (int64_t r = w0; r < n_out; r += wn) {          // one wave per row
    const int64_t i = idx[r];
    const float* s = (i >= 0 && i < n_src) ? src + i * ld_src : fill;
    float* d = dst + r * ld_dst;
    if (vec) {
      const int w4 = width >> 2, d4 = dst_width >> 2;
      for (int c = lane; c < d4; c += 64) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < w4 && s) v = reinterpret_cast<const float4*>(s)[c];
        reinterpret_cast<float4*>(d)[c] = v;
      }
    } else {
      for (int c = lane; c < dst_width; c += 64) d[c] = (c < width && s) ? s[c] : 0.f;
    }
  }
}

extern "C" int itts_rows_gather_f32(const float* d_src, int64_t ld_src, int64_t n_src, const int64_t* d_idx,
                                    int64_t n_out, int width, const float* d_fill_row, float* d_dst, int64_t ld_dst,
                                    int dst_width, void* stream) {
  ITTS_REQUIRE(n_out == 0 || (d_idx && d_dst && (d_src || n_src == 0)), "null pointer");
  ITTS_REQUIRE(n_out >= 0 && n_src >= 0 && width >= 0 && dst_width >= width && ld_src >= width && ld_dst >= dst_width,
               "bad sizes");
  if (n_out == 0 || dst_width == 0) return ITTS_OK;
  const int vec = width % 4 == 0 && dst_width % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && aligned16(d_src) &&
                  aligned16(d_dst) && (!d_fill_row || aligned16(d_fill_row));
  const unsigned grid = (unsigned)std::min<int64_t>((n_out + 3) / 4, 16384);
  hipLaunchKernelGGL(rows_gather_kernel, dim3(grid), dim3(256), 0, as_stream(stream), d_src, ld_src, n_src, d_idx, n_out,
                     width, d_fill_row, d_dst, ld_dst, dst_width, vec);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int64_t itts_linear_fwd_mse_workspace_bytes(int64_t M, int N) {
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + 63) / 64);
  return std::max<int64_t>(tiles, kRingGrid) * 8;   // one partial sum per tile, or per persistent workgroup
}

// Forward of the output layer fused with NamedLoss(MSELoss, 'mean_per_frame'): d_loss [1] and
// d_dz [M, N] = d loss / d y; y itself is not stored (FFWrapper.py:63-73 followed by
// NamedLoss.py:70-117 in a training step).  Falls back to nothing: the buffers must allow 16-byte
// rows for x, w and dz (the trainer's buffers do); other callers use itts_linear_fwd +
// itts_masked_mse.
extern "C" int itts_linear_fwd_mse(const float* d_x, int64_t ldx, const float* d_w, const float* d_b,
                                   const float* d_target, int64_t ldt, const uint8_t* d_row_valid,
                                   double n_valid, float loss_weight, int64_t M, int N, int K,
                                   float* d_loss, float* d_dz, int64_t lddz, void* d_workspace,
                                   void* stream) {
  ITTS_REQUIRE(d_w && d_loss && d_workspace && (M == 0 || (d_x && d_target && d_row_valid && d_dz)),
               "null pointer");
  ITTS_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K && ldt >= N && lddz >= N && n_valid > 0, "bad sizes");
  hipStream_t s = as_stream(stream);
  if (M == 0) {
    ITTS_HIP_CHECK(hipMemsetAsync(d_loss, 0, 4, s));
    return ITTS_OK;
  }
  ITTS_REQUIRE((ldx % 4 == 0) && aligned16(d_x) && (K % 4 == 0) && aligned16(d_w) && (lddz % 4 == 0) &&
                   aligned16(d_dz),
               "x, w and dz need 16-byte aligned rows (use itts_linear_fwd + itts_masked_mse otherwise)");
  GemmArgs g{};
  g.A = d_x; g.lda = ldx; g.B = d_w; g.ldb = K; g.C = d_dz; g.ldc = lddz;
  g.M = M; g.N = N; g.K = K; g.bias = d_b; g.act = 0;
  g.kchunk = ((K + BK - 1) / BK) * BK; g.slab_stride = 0;
  g.vecA = 1; g.vecB = 1; g.wide_out = 1;
  g.aux = d_target; g.ldaux = ldt; g.row_valid = d_row_valid;
  const double scale = (double)loss_weight / (n_valid * (double)N);
  g.gscale = (float)(2.0 * scale);
  g.loss_partial = reinterpret_cast<double*>(d_workspace);
  int64_t tiles = ((M + BM - 1) / BM) * ((N + 63) / 64);
  int rc;
  if (ring_ok<true, true>(g, 1, EPI_MSE)) {
    tiles = std::min<int64_t>(kRingGrid, (tiles + 7) / 8 * 8);   // partial sums per persistent workgroup
    rc = launch_ring_wm<true, true, EPI_MSE, 2>(g, 1, s);
  } else {
    rc = launch_gemm_tn<true, true, EPI_MSE, 1, 1>(g, 1, s);
  }
  if (rc) return rc;
  return launch_mse_final(reinterpret_cast<const double*>(d_workspace), (int)tiles, scale, d_loss, s);
}

extern "C" int itts_act_bwd(const float* d_dy, const float* d_y, float* d_dz, int64_t n_elem,
                            int act, void* stream) {
  ITTS_REQUIRE(d_dy && d_y && d_dz && n_elem >= 0, "bad arguments");
  ITTS_REQUIRE(act >= 0 && act <= 2, "unknown activation");
  if (n_elem == 0) return ITTS_OK;
  const int blocks = (int)std::min<int64_t>((n_elem + 255) / 256, 4096);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), d_dy, d_y,
                     d_dz, n_elem, act);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_linear_bwd_input(const float* d_dz, int64_t lddz, const float* d_w,
                                     float* d_dx, int64_t lddx, const float* d_yprev,
                                     int64_t ldyp, int act_prev, int64_t M, int N, int K,
                                     void* stream) {
  ITTS_REQUIRE(d_w && (M == 0 || (d_dz && d_dx)), "null pointer");
  ITTS_REQUIRE(M >= 0 && N > 0 && K > 0 && lddz >= N && lddx >= K, "bad sizes");
  if (M == 0) return ITTS_OK;
  // dx[M,K] = dz[M,N] (row form, reduction N) x w[N,K] (col form: [k=N][out=K])
  GemmArgs g{};
  g.A = d_dz; g.lda = lddz; g.B = d_w; g.ldb = K; g.C = d_dx; g.ldc = lddx;
  g.M = M; g.N = K; g.K = N; g.aux = d_yprev; g.ldaux = ldyp; g.act = act_prev;
  g.kchunk = ((N + BK - 1) / BK) * BK; g.slab_stride = 0;
  g.vecA = (lddz % 4 == 0) && aligned16(d_dz);
  g.vecB = (K % 4 == 0) && aligned16(d_w);
  g.wide_out = (lddx % 4 == 0) && aligned16(d_dx) && (!d_yprev || ((ldyp % 4 == 0) && aligned16(d_yprev)));
  if (d_yprev) {
    ITTS_REQUIRE(ldyp >= K, "ldyp too small");
    return launch_gemm<true, false, EPI_DACT>(g, 1, as_stream(stream));
  }
  return launch_gemm<true, false, EPI_STORE>(g, 1, as_stream(stream));
}

extern "C" int64_t itts_linear_bwd_weight_workspace_bytes(int64_t M, int N, int K) {
  if (M < 0 || N <= 0 || K <= 0) return 0;
  const int s = std::max(choose_splitk(M, N, K), choose_splitk_ring(M, N, K));
  return (int64_t)s * ((int64_t)N * K + N) * 4 + (int64_t)kColsumSlices * N * 4 + 8192;
}

extern "C" int itts_linear_bwd_weight(const float* d_dz, int64_t lddz, const float* d_x,
                                      int64_t ldx, float* d_dw, float* d_db, int64_t M, int N,
                                      int K, void* d_workspace, int accumulate, void* stream) {
  ITTS_REQUIRE(d_dw && d_workspace && (M == 0 || (d_dz && d_x)), "null pointer");
  ITTS_REQUIRE(M >= 0 && N > 0 && K > 0 && lddz >= N && ldx >= K, "bad sizes");
  hipStream_t s = as_stream(stream);
  if (M == 0) {
    if (!accumulate) {
      ITTS_HIP_CHECK(hipMemsetAsync(d_dw, 0, (size_t)N * K * 4, s));
      if (d_db) ITTS_HIP_CHECK(hipMemsetAsync(d_db, 0, (size_t)N * 4, s));
    }
    return ITTS_OK;
  }
  const bool vec_ok = (lddz % 4 == 0) && aligned16(d_dz) && (ldx % 4 == 0) && aligned16(d_x) && (K % 4 == 0);
  const int S = vec_ok ? choose_splitk_ring(M, N, K) : choose_splitk(M, N, K);
  int64_t kchunk = (M + S - 1) / S;
  kchunk = ((kchunk + BK - 1) / BK) * BK;
  const int S_eff = (int)((M + kchunk - 1) / kchunk);
  float* slabs = reinterpret_cast<float*>(d_workspace);
  // dw[N,K] = dz^T x: A = dz as col form [k=M][out=N]; B = x as col form [k=M][out=K]
  GemmArgs g{};
  g.A = d_dz; g.lda = lddz; g.B = d_x; g.ldb = ldx; g.C = slabs; g.ldc = K;
  g.M = N; g.N = K; g.K = M; g.kchunk = kchunk; g.slab_stride = (int64_t)N * K;
  g.vecA = (lddz % 4 == 0) && aligned16(d_dz);
  g.vecB = (ldx % 4 == 0) && aligned16(d_x);
  const int64_t n = (int64_t)N * K;
  if (d_db) {
    // The bias gradient (column sums of dz) comes out of the same launch: the workgroups of the
    // first column tile add up their resident dz tiles.  When db follows dw in memory (the flat
    // gradient arenas) and both lengths are multiples of 4, one reduction serves both.
    const bool merged = d_db == d_dw + n && N % 4 == 0 && n % 4 == 0;
    const int64_t stride = merged ? n + N : n;
    g.slab_stride = stride;
    g.bias_part = merged ? slabs + n : slabs + (int64_t)S_eff * n;
    g.bias_part_stride = merged ? stride : N;
    int rc = launch_gemm<false, false, EPI_STORE>(g, S_eff, s);
    if (rc) return rc;
    if (merged) return launch_reduce_slabs(slabs, S_eff, n + N, d_dw, accumulate, s);
    rc = launch_reduce_slabs(slabs, S_eff, n, d_dw, accumulate, s);
    if (rc) return rc;
    return launch_reduce_slabs(g.bias_part, S_eff, (int64_t)N, d_db, accumulate, s);
  }
  int rc = launch_gemm<false, false, EPI_STORE>(g, S_eff, s);
  if (rc) return rc;
  return launch_reduce_slabs(slabs, S_eff, n, d_dw, accumulate, s);
}

// Weight gradient (+ bias gradient) and input gradient of one layer from one call: with operands
// that allow the LDS-DMA kernel the two GEMMs share a launch (gemm_ring_pair_kernel), otherwise
// this is itts_linear_bwd_weight followed by itts_linear_bwd_input.  Same results either way
// (bit for bit: the tiles, the split-K chunks and their order are those of the separate calls).
extern "C" int itts_linear_bwd(const float* d_dz, int64_t lddz, const float* d_x, int64_t ldx,
                               const float* d_w, float* d_dw, float* d_db, float* d_dx, int64_t lddx,
                               const float* d_yprev, int64_t ldyp, int act_prev, int64_t M, int N, int K,
                               void* d_workspace, int accumulate, void* stream) {
  ITTS_REQUIRE(d_w && d_dw && d_workspace && (M == 0 || (d_dz && d_x && d_dx)), "null pointer");
  ITTS_REQUIRE(M >= 0 && N > 0 && K > 0 && lddz >= N && ldx >= K && lddx >= K, "bad sizes");
  ITTS_REQUIRE(!d_yprev || ldyp >= K, "ldyp too small");
  hipStream_t s = as_stream(stream);
  const bool vec_ok = (lddz % 4 == 0) && aligned16(d_dz) && (ldx % 4 == 0) && aligned16(d_x) && (K % 4 == 0) &&
                      aligned16(d_w);
  const int64_t n = (int64_t)N * K;
  const bool merged = d_db && d_db == d_dw + n && N % 4 == 0 && n % 4 == 0;
  bool fused = vec_ok && M > 0 && (lddx % 4 == 0) && aligned16(d_dx) &&
               (!d_yprev || ((ldyp % 4 == 0) && aligned16(d_yprev)));
  GemmArgs gw{}, gx{};
  int S_eff = 1;
  if (fused) {
    const int S = choose_splitk_ring(M, N, K);
    int64_t kchunk = (M + S - 1) / S;
    kchunk = ((kchunk + BK - 1) / BK) * BK;
    S_eff = (int)((M + kchunk - 1) / kchunk);
    float* slabs = reinterpret_cast<float*>(d_workspace);
    gw.A = d_dz; gw.lda = lddz; gw.B = d_x; gw.ldb = ldx; gw.C = slabs; gw.ldc = K;
    gw.M = N; gw.N = K; gw.K = M; gw.kchunk = kchunk; gw.vecA = gw.vecB = 1;
    const int64_t stride = merged ? n + N : n;
    gw.slab_stride = stride;
    if (merged) { gw.bias_part = slabs + n; gw.bias_part_stride = stride; }
    else if (d_db) { gw.bias_part = slabs + (int64_t)S_eff * n; gw.bias_part_stride = N; }
    gx.A = d_dz; gx.lda = lddz; gx.B = d_w; gx.ldb = K; gx.C = d_dx; gx.ldc = lddx;
    gx.M = M; gx.N = K; gx.K = N; gx.aux = d_yprev; gx.ldaux = ldyp; gx.act = act_prev;
    gx.kchunk = ((N + BK - 1) / BK) * BK; gx.vecA = gx.vecB = 1;
    fused = ring_ok<false, false>(gw, S_eff, EPI_STORE) && ring_ok<true, false>(gx, 1, d_yprev ? EPI_DACT : EPI_STORE);
  }
  if (!fused) {
    int rc = itts_linear_bwd_weight(d_dz, lddz, d_x, ldx, d_dw, d_db, M, N, K, d_workspace, accumulate, stream);
    if (rc) return rc;
    return itts_linear_bwd_input(d_dz, lddz, d_w, d_dx, lddx, d_yprev, ldyp, act_prev, M, N, K, stream);
  }
  int rc;
  const int wm = ring_wm(gw.M, gw.N);
  if (d_yprev) rc = wm == 1 ? launch_ring_bwd_pair<1, EPI_DACT>(gw, S_eff, gx, s) : launch_ring_bwd_pair<2, EPI_DACT>(gw, S_eff, gx, s);
  else rc = wm == 1 ? launch_ring_bwd_pair<1, EPI_STORE>(gw, S_eff, gx, s) : launch_ring_bwd_pair<2, EPI_STORE>(gw, S_eff, gx, s);
  if (rc) return rc;
  float* slabs = reinterpret_cast<float*>(d_workspace);
  rc = launch_reduce_slabs(slabs, S_eff, merged ? n + N : n, d_dw, accumulate, s);
  if (rc || merged || !d_db) return rc;
  return launch_reduce_slabs(slabs + (int64_t)S_eff * n, S_eff, (int64_t)N, d_db, accumulate, s);
}

static int mse_blocks(int64_t M, int D) {
  return (int)std::max<int64_t>(1, std::min<int64_t>((M * D + 255) / 256, 2048));
}

extern "C" int64_t itts_masked_mse_workspace_bytes(int64_t M, int D) {
  if (M < 0 || D <= 0) return 0;
  return (int64_t)mse_blocks(M, D) * 8;
}

extern "C" int itts_masked_mse(const float* d_pred, int64_t ldp, const float* d_target,
                               int64_t ldt, const uint8_t* d_row_valid, int64_t M, int D,
                               double n_valid, float loss_weight, float* d_loss, float* d_grad,
                               int64_t ldg, void* d_workspace, void* stream) {
  ITTS_REQUIRE(d_pred && d_target && d_row_valid && d_loss && d_workspace, "null pointer");
  ITTS_REQUIRE(M >= 0 && D > 0 && ldp >= D && ldt >= D && n_valid > 0, "bad sizes");
  ITTS_REQUIRE(!d_grad || ldg >= D, "ldg too small");
  hipStream_t s = as_stream(stream);
  if (M == 0) {
    ITTS_HIP_CHECK(hipMemsetAsync(d_loss, 0, 4, s));
    return ITTS_OK;
  }
  const int nb = mse_blocks(M, D);
  const double scale = (double)loss_weight / (n_valid * (double)D);
  hipLaunchKernelGGL(masked_mse_kernel, dim3(nb), dim3(256), 0, s, d_pred, ldp, d_target, ldt,
                     d_row_valid, M, D, (float)(2.0 * scale), d_grad, ldg,
                     reinterpret_cast<double*>(d_workspace));
  ITTS_LAUNCH_CHECK();
  return launch_mse_final(reinterpret_cast<const double*>(d_workspace), nb, scale, d_loss, s);
}

extern "C" int itts_weighted_loss(const float* d_pred, int64_t ldp, const float* d_target, int64_t ldt,
                                  const float* d_row_weight, int64_t M, int D, int kind,
                                  float* d_loss, float* d_grad, int64_t ldg, float* d_elem,
                                  int64_t lde, void* d_workspace, void* stream) {
  ITTS_REQUIRE(d_pred && d_target && d_row_weight && d_loss && d_workspace, "null pointer");
  ITTS_REQUIRE(M >= 0 && D > 0 && ldp >= D && ldt >= D && (kind == 0 || kind == 1), "bad sizes");
  ITTS_REQUIRE((!d_grad || ldg >= D) && (!d_elem || lde >= D), "leading dimension too small");
  hipStream_t s = as_stream(stream);
  if (M == 0) {
    ITTS_HIP_CHECK(hipMemsetAsync(d_loss, 0, 4, s));
    return ITTS_OK;
  }
  const int nb = mse_blocks(M, D);
  hipLaunchKernelGGL(weighted_loss_kernel, dim3(nb), dim3(256), 0, s, d_pred, ldp, d_target, ldt,
                     d_row_weight, M, D, kind, d_grad, ldg, d_elem, lde,
                     reinterpret_cast<double*>(d_workspace));
  ITTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(masked_mse_final_kernel, dim3(1), dim3(256), 0, s,
                     reinterpret_cast<const double*>(d_workspace), nb, 1.0, d_loss);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_adam_step(float* d_param, const float* d_grad, float* d_exp_avg,
                              float* d_exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int64_t step, float grad_scale,
                              void* stream) {
  ITTS_REQUIRE(d_param && d_grad && d_exp_avg && d_exp_avg_sq, "null pointer");
  ITTS_REQUIRE(n >= 0 && step >= 1, "bad sizes (step counts from 1)");
  if (n == 0) return ITTS_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), d_param, d_grad,
                     d_exp_avg, d_exp_avg_sq, n, beta1, beta2, eps, weight_decay, step_size,
                     bc2_sqrt, grad_scale);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_grad_norm_accum(const float* d_x, int64_t n, int norm_kind, float* d_accum,
                                    int accumulate, void* d_workspace, void* stream) {
  ITTS_REQUIRE(d_accum && d_workspace && (n == 0 || d_x), "null pointer");
  ITTS_REQUIRE(n >= 0 && (norm_kind == 2 || norm_kind == 0), "norm_kind must be 2 or 0 (infinity)");
  const int nb = (int)std::min<int64_t>((n + 255) / 256, kNormBlocks);
  hipStream_t s = as_stream(stream);
  float* part = static_cast<float*>(d_workspace);
  if (nb > 0) {
    hipLaunchKernelGGL(grad_norm_partial_kernel, dim3(nb), dim3(256), 0, s, d_x, n, norm_kind, part);
    ITTS_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(grad_norm_final_kernel, dim3(1), dim3(256), 0, s, part, nb, norm_kind, d_accum,
                     accumulate);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_adam_step_fused(float* d_param, const float* d_grad, float* d_exp_avg,
                                    float* d_exp_avg_sq, int64_t n, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int64_t step,
                                    float grad_scale, const float* d_norm_accum, int norm_kind,
                                    float clip_max_norm, float clip_value, float* d_ema_shadow,
                                    float ema_decay, void* stream) {
  ITTS_REQUIRE(d_param && d_grad && d_exp_avg && d_exp_avg_sq, "null pointer");
  ITTS_REQUIRE(n >= 0 && step >= 1, "bad sizes (step counts from 1)");
  ITTS_REQUIRE(!d_norm_accum || ((norm_kind == 2 || norm_kind == 0) && clip_max_norm > 0.f),
               "norm clipping needs norm_kind 2 or 0 and a positive max norm");
  if (n == 0) return ITTS_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(adam_fused_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), d_param,
                     d_grad, d_exp_avg, d_exp_avg_sq, n, beta1, beta2, eps, weight_decay,
                     (float)(lr / bc1), (float)sqrt(bc2), grad_scale, d_norm_accum, norm_kind,
                     clip_max_norm, clip_value, d_ema_shadow, 1.f - ema_decay);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_sgd_step(float* d_param, const float* d_grad, float* d_momentum_buf, int64_t n,
                             float lr, float momentum, float dampening, float weight_decay,
                             int nesterov, int first_step, float grad_scale, void* stream) {
  ITTS_REQUIRE(d_param && d_grad, "null pointer");
  ITTS_REQUIRE(momentum == 0.f || d_momentum_buf, "momentum needs a buffer");
  ITTS_REQUIRE(n >= 0, "bad size");
  if (n == 0) return ITTS_OK;
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), d_param, d_grad,
                     d_momentum_buf, n, lr, momentum, dampening, weight_decay, nesterov,
                     first_step, grad_scale);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_ema_update(float* d_shadow, const float* d_param, int64_t n, float decay,
                               void* stream) {
  ITTS_REQUIRE(d_shadow && d_param, "null pointer");
  ITTS_REQUIRE(n >= 0, "bad size");
  if (n == 0) return ITTS_OK;
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(ema_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), d_shadow, d_param,
                     n, 1.f - decay);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}
