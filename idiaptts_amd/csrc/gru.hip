// (Bi)GRU recurrence for padded batches of frame sequences: the GRU flavour of the recurrent
// acoustic-model groups (rnn_dyn/RNNWrapper.py:45-107 builds torch.nn.GRU for `..._BiGRU_...`
// groups; cuDNN / MIOpen RNN in the reference).  Same split of the work as lstm.hip:
//   * gin = X W_ih^T + b_ih for all time steps and both directions is one fp32-MFMA GEMM (nn.hip),
//     and so are dX, dW_ih, dW_hh and the bias gradients of the backward pass;
//   * the recurrence h_{t-1} W_hh^T runs one launch per step (the kernel boundary is the grid-wide
//     dependency), both directions in it, operands streamed from L2 into
//     v_mfma_f32_16x16x4_f32 with the K-permutation trick (one 16-byte load feeds 4 MFMAs);
//   * packed-sequence semantics and the packed row layout of lstm.hip (rows sorted by decreasing
//     length, frame t of row b at packed row row_off[t] + b, step s launches only the batch tiles
//     that still hold active rows, the reverse direction starts at each row's own last frame).
// torch.nn.GRU cell, gate order r, z, n:
//   r = sigmoid(gin_r + W_hr h + b_hr)      z = sigmoid(gin_z + W_hz h + b_hz)
//   n = tanh(gin_n + r * (W_hn h + b_hn))   h' = (1 - z) * n + z * h
#include <algorithm>

#include "common.h"

namespace itts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GRU_FW_UNITS = 4;    // hidden units per workgroup, forward (12 of 16 tile rows used)
constexpr int GRU_BW_UNITS = 16;   // hidden units per workgroup, backward

struct GruArgs {
  int T, B, H, ndir;
  const int* lengths;     // [B] sorted by decreasing length
  const int* row_off;     // [T] packed row of (t, b) = row_off[t] + b
  const float* gin;       // [N, ndir*3H] input projections incl. b_ih (N = sum of lengths)
  const float* whh;       // [ndir][3H][H]
  const float* whh_t;     // [ndir][H][3H]   (backward)
  const float* bhh;       // [ndir][3H]
  const float* h0;        // [ndir][H] or NULL
  float* hs;              // [2 parity][ndir][B][H] running hidden state (fwd) / carried dh*z (bwd)
  float* y;               // [N, ndir*H]
  float* gates;           // [N, ndir*3H] r, z, n after activation (saved for backward)
  float* hnpre;           // [N, ndir*H]  W_hn h + b_hn
  float* hprev;           // [N, ndir*H]  h_{t-1} that entered step t
  const float* dy;        // [N, ndir*H]
  float* dgi;             // [N, ndir*3H] gradient wrt gin  (da_r, da_z, da_n)
  float* dgh;             // [N, ndir*3H] gradient wrt the hidden projections (da_r, da_z, da_n*r)
  int step;
  int ksplit, kiter;
};

__device__ __forceinline__ float gru_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ int gru_time_of(int dir, int s, int len) {
  if (s >= len) return -1;
  return dir == 0 ? s : len - 1 - s;
}

// ---- forward step ---------------------------------------------------------------------------------
// Workgroup = 4 hidden units x 3 gates (tile rows 0..11, row = gate*4 + unit) x 16 batch rows; the
// waves split K = H, partial tiles meet in LDS, 64 threads apply the cell update.
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(GruArgs a) {
  __shared__ float P[4][16][17];
  const int H = a.H, B = a.B, G3 = 3 * H;
  const int dir = blockIdx.y;
  const int ngroups = H / GRU_FW_UNITS;
  const int j0 = (blockIdx.x % ngroups) * GRU_FW_UNITS;
  const int b0 = (blockIdx.x / ngroups) * 16;
  const int par = a.step & 1;
  const float* whh = a.whh + (size_t)dir * G3 * H;
  const float* hprev = a.hs + ((size_t)par * a.ndir + dir) * B * H;
  float* hnext = a.hs + ((size_t)(par ^ 1) * a.ndir + dir) * B * H;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;

  const int bl = threadIdx.x >> 2, u = threadIdx.x & 3;
  const int b = b0 + bl, j = j0 + u;
  const bool ew = threadIdx.x < 64 && b < B;
  int t = -1;
  float hp_v = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, bh0 = 0.f, bh1 = 0.f, bh2 = 0.f;
  size_t r = 0;
  if (ew) {
    t = gru_time_of(dir, a.step, a.lengths[b]);
    hp_v = hprev[(size_t)b * H + j];
    if (t >= 0) {
      r = (size_t)a.row_off[t] + b;
      const float* gi = a.gin + r * (size_t)(a.ndir * G3) + (size_t)dir * G3 + j;
      g0 = gi[0]; g1 = gi[H]; g2 = gi[2 * H];
      const float* bh = a.bhh + (size_t)dir * G3 + j;
      bh0 = bh[0]; bh1 = bh[H]; bh2 = bh[2 * H];
    }
  }

  const int row = b0 + lr;
  const bool rok = row < B;
  const int kiter = wv < a.ksplit ? a.kiter : 0;
  const int kbase = wv * (16 * a.kiter) + 4 * kg;
  const int gate = lr >> 2;                    // tile row -> gate; rows 12..15 are padding
  const bool gok = gate < 3;
  const float* hp = hprev + (size_t)(rok ? row : 0) * H + (kiter ? kbase : 0);
  const float* wp = whh + (size_t)((gok ? gate : 0) * H + j0 + (lr & 3)) * H + (kiter ? kbase : 0);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int c = 0; c * 8 < kiter; ++c) {
    float4 av[8], bv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int i = c * 8 + s;
      const int o = i < kiter ? 16 * i : 0;
      av[s] = *reinterpret_cast<const float4*>(hp + o);
      bv[s] = *reinterpret_cast<const float4*>(wp + o);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float4 x = av[s];
      if (!rok || c * 8 + s >= kiter) x = make_float4(0.f, 0.f, 0.f, 0.f);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, bv[s].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, bv[s].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, bv[s].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, bv[s].w, acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) P[wv][kg * 4 + q][lr] = acc0[q] + acc1[q];
  __syncthreads();
  if (ew) {
    float hn = hp_v;
    if (t >= 0) {
      auto proj = [&](int n) { return (P[0][bl][n] + P[1][bl][n]) + (P[2][bl][n] + P[3][bl][n]); };
      const float rg = gru_sigmoid(g0 + proj(u) + bh0);
      const float zg = gru_sigmoid(g1 + proj(4 + u) + bh1);
      const float hnp = proj(8 + u) + bh2;
      const float ng = tanhf(g2 + rg * hnp);
      hn = (1.f - zg) * ng + zg * hp_v;
      const size_t oh = r * (size_t)(a.ndir * H) + (size_t)dir * H + j;
      a.y[oh] = hn;
      if (a.gates) {
        float* gs = a.gates + r * (size_t)(a.ndir * G3) + (size_t)dir * G3 + j;
        gs[0] = rg; gs[H] = zg; gs[2 * H] = ng;
        a.hnpre[oh] = hnp;
        a.hprev[oh] = hp_v;
      }
    }
    hnext[(size_t)b * H + j] = hn;
  }
}

__global__ void gru_init_state_kernel(const float* __restrict__ h0, float* __restrict__ hs, int ndir,
                                      int B, int H) {
  const int64_t n = (int64_t)ndir * B * H;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % H);
    const int d = (int)(i / ((int64_t)B * H));
    hs[i] = h0 ? h0[d * H + j] : 0.f;
  }
}

// ---- backward step --------------------------------------------------------------------------------
// Recurrence step s = a.step (called with s = T-1 ... 0).  For row b active at s (time t):
//   dh   = dy[t] + dGh[t_{s+1}] W_hh + carry            carry = dh_{s+1} * z_{s+1} (0 if inactive)
//   dn   = dh (1 - z)      dz = dh (h_prev - n)         new carry = dh * z
//   da_n = dn (1 - n^2)    da_r = da_n * hn_pre * r (1 - r)      da_z = dz * z (1 - z)
//   dGi[t] = (da_r, da_z, da_n)      dGh[t] = (da_r, da_z, da_n * r)
// Workgroup = 16 hidden units x 16 batch rows; three waves take one gate block (K = H) each of the
// 3H rows of W_hh^T, the fourth idles.
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(GruArgs a) {
  __shared__ float P[4][16][17];
  const int H = a.H, B = a.B, G3 = 3 * H;
  const int dir = blockIdx.y;
  const int ngroups = H / GRU_BW_UNITS;
  const int j0 = (blockIdx.x % ngroups) * GRU_BW_UNITS;
  const int b0 = (blockIdx.x / ngroups) * 16;
  const int par = a.step & 1;
  const float* wt = a.whh_t + (size_t)dir * H * G3;
  const float* carry_in = a.hs + ((size_t)(par ^ 1) * a.ndir + dir) * B * H;
  float* carry_out = a.hs + ((size_t)par * a.ndir + dir) * B * H;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  const size_t ldg = (size_t)a.ndir * G3, ldh = (size_t)a.ndir * H;

  const int bl = threadIdx.x >> 4, n = threadIdx.x & 15;
  const int b = b0 + bl, j = j0 + n;
  int t = -1;
  float rg = 0.f, zg = 0.f, ng = 0.f, hnp = 0.f, hpv = 0.f, dyv = 0.f, cin = 0.f;
  size_t r = 0;
  if (b < B) {
    t = gru_time_of(dir, a.step, a.lengths[b]);
    if (t >= 0) {
      r = (size_t)a.row_off[t] + b;
      const float* gs = a.gates + r * ldg + (size_t)dir * G3 + j;
      rg = gs[0]; zg = gs[H]; ng = gs[2 * H];
      const size_t oh = r * ldh + (size_t)dir * H + j;
      hnp = a.hnpre[oh];
      hpv = a.hprev[oh];
      dyv = a.dy[oh];
      cin = carry_in[(size_t)b * H + j];
    }
  }

  const int row = b0 + lr;
  bool has_next = false;
  const float* dgp = a.dgh;
  if (row < B) {
    const int tn = gru_time_of(dir, a.step + 1, a.lengths[row]);
    if (tn >= 0) {
      has_next = true;
      dgp = a.dgh + ((size_t)a.row_off[tn] + row) * ldg + (size_t)dir * G3;
    }
  }
  const int kiter = wv < 3 ? H / 16 : 0;
  const int kbase = (wv < 3 ? wv * H : 0) + 4 * kg;
  dgp += kbase;
  const float* wp = wt + (size_t)(j0 + lr) * G3 + kbase;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int c = 0; c * 8 < kiter; ++c) {
    float4 av[8], bv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int i = c * 8 + s;
      const int o = i < kiter ? 16 * i : 0;
      av[s] = *reinterpret_cast<const float4*>(dgp + o);
      bv[s] = *reinterpret_cast<const float4*>(wp + o);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float4 x = av[s];
      if (!has_next || c * 8 + s >= kiter) x = make_float4(0.f, 0.f, 0.f, 0.f);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, bv[s].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, bv[s].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, bv[s].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, bv[s].w, acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) P[wv][kg * 4 + q][lr] = acc0[q] + acc1[q];
  __syncthreads();
  if (b < B) {
    float carry = 0.f;
    if (t >= 0) {
      const float dhr = (P[0][bl][n] + P[1][bl][n]) + (P[2][bl][n] + P[3][bl][n]);
      const float dh = dyv + dhr + cin;
      const float dn = dh * (1.f - zg);
      const float dz = dh * (hpv - ng);
      const float dan = dn * (1.f - ng * ng);
      const float dar = dan * hnp * rg * (1.f - rg);
      const float daz = dz * zg * (1.f - zg);
      float* gi = a.dgi + r * ldg + (size_t)dir * G3 + j;
      float* gh = a.dgh + r * ldg + (size_t)dir * G3 + j;
      gi[0] = dar; gi[H] = daz; gi[2 * H] = dan;
      gh[0] = dar; gh[H] = daz; gh[2 * H] = dan * rg;
      carry = dh * zg;
    }
    carry_out[(size_t)b * H + j] = carry;
  }
}

__global__ void gru_final_state_kernel(const float* __restrict__ st, const int* __restrict__ lengths,
                                       float* __restrict__ out, int ndir, int B, int H) {
  const int64_t n = (int64_t)ndir * B * H;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)((i / H) % B);
    out[i] = st[(int64_t)(lengths[b] & 1) * n + i];   // parity written by the row's last active step
  }
}

}  // namespace itts

using namespace itts;

static int gru_check(const int* h_lengths, int T, int B, int H, int ndir) {
  ITTS_REQUIRE(T >= 1 && B >= 1 && (ndir == 1 || ndir == 2), "bad sizes");
  ITTS_REQUIRE(H >= 16 && H % 16 == 0 && H <= 4096, "hidden size must be a multiple of 16");
  ITTS_REQUIRE(h_lengths != nullptr, "host copy of the lengths is required");
  ITTS_REQUIRE(h_lengths[0] == T && h_lengths[B - 1] >= 1, "T must be the longest length, all lengths >= 1");
  for (int b = 1; b < B; ++b) ITTS_REQUIRE(h_lengths[b] <= h_lengths[b - 1], "rows must be sorted by decreasing length");
  return ITTS_OK;
}

static inline int gru_active_rows(const int* h_lengths, int B, int s, int* p) {
  while (*p > 0 && h_lengths[*p - 1] <= s) --*p;
  while (*p < B && h_lengths[*p] > s) ++*p;
  return *p;
}

extern "C" int64_t itts_gru_state_bytes(int B, int H, int ndir) {
  if (B <= 0 || H <= 0 || ndir <= 0) return 0;
  return (int64_t)2 * ndir * B * H * 4;  // two parities of the running state
}

extern "C" int itts_gru_layer_fwd(const float* d_gin, const float* d_whh, const float* d_bhh,
                                  const float* d_h0, const int* d_lengths, const int* h_lengths,
                                  const int* d_row_off, int T, int B, int H, int ndir, float* d_y,
                                  float* d_gates, float* d_hnpre, float* d_hprev, float* d_hn,
                                  void* d_state, void* stream) {
  ITTS_REQUIRE(d_gin && d_whh && d_bhh && d_lengths && d_row_off && d_y && d_state, "null pointer");
  ITTS_REQUIRE((d_gates == nullptr) == (d_hnpre == nullptr) && (d_gates == nullptr) == (d_hprev == nullptr),
               "gates / hnpre / hprev must be given together (training) or all NULL (inference)");
  int rc = gru_check(h_lengths, T, B, H, ndir);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  GruArgs a{};
  a.T = T; a.B = B; a.H = H; a.ndir = ndir; a.lengths = d_lengths; a.row_off = d_row_off; a.gin = d_gin;
  a.whh = d_whh; a.bhh = d_bhh; a.h0 = d_h0; a.y = d_y; a.gates = d_gates; a.hnpre = d_hnpre;
  a.hprev = d_hprev;
  a.hs = reinterpret_cast<float*>(d_state);
  const int64_t n = (int64_t)ndir * B * H;
  const dim3 eg((unsigned)std::min<int64_t>((n + 255) / 256, 1024));
  hipLaunchKernelGGL(gru_init_state_kernel, eg, dim3(256), 0, s, d_h0, a.hs, ndir, B, H);
  ITTS_LAUNCH_CHECK();
  a.ksplit = (H % 64 == 0) ? 4 : ((H % 32 == 0) ? 2 : 1);
  a.kiter = H / (16 * a.ksplit);
  int p = B;
  for (int step = 0; step < T; ++step) {
    a.step = step;
    const int nact = gru_active_rows(h_lengths, B, step, &p);
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3((H / GRU_FW_UNITS) * ((nact + 15) / 16), ndir), dim3(256), 0,
                       s, a);
  }
  ITTS_LAUNCH_CHECK();
  if (d_hn) {
    hipLaunchKernelGGL(gru_final_state_kernel, eg, dim3(256), 0, s, a.hs, d_lengths, d_hn, ndir, B, H);
    ITTS_LAUNCH_CHECK();
  }
  return ITTS_OK;
}

extern "C" int itts_gru_layer_bwd(const float* d_dy, const float* d_whh_t, const float* d_gates,
                                  const float* d_hnpre, const float* d_hprev, const int* d_lengths,
                                  const int* h_lengths, const int* d_row_off, int T, int B, int H,
                                  int ndir, float* d_dgi, float* d_dgh, void* d_state, void* stream) {
  ITTS_REQUIRE(d_dy && d_whh_t && d_gates && d_hnpre && d_hprev && d_lengths && d_row_off && d_dgi && d_dgh &&
                   d_state, "null pointer");
  int rc = gru_check(h_lengths, T, B, H, ndir);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  GruArgs a{};
  a.T = T; a.B = B; a.H = H; a.ndir = ndir; a.lengths = d_lengths; a.row_off = d_row_off; a.whh_t = d_whh_t;
  a.gates = const_cast<float*>(d_gates); a.hnpre = const_cast<float*>(d_hnpre);
  a.hprev = const_cast<float*>(d_hprev); a.dy = d_dy; a.dgi = d_dgi; a.dgh = d_dgh;
  a.hs = reinterpret_cast<float*>(d_state);
  ITTS_HIP_CHECK(hipMemsetAsync(a.hs, 0, (size_t)2 * ndir * B * H * 4, s));   // carry of not-yet-active rows
  int p = 0;
  for (int step = T - 1; step >= 0; --step) {
    a.step = step;
    const int nact = gru_active_rows(h_lengths, B, step, &p);
    hipLaunchKernelGGL(gru_step_bwd_kernel, dim3((H / GRU_BW_UNITS) * ((nact + 15) / 16), ndir), dim3(256), 0,
                       s, a);
  }
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}
