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
#include <vector>

#include "rnn_common.h"
#include "rnn_persist.h"

namespace itts {

constexpr int GRU_FW_UNITS = 4;    // hidden units per workgroup, forward (12 of 16 tile rows used)
constexpr int GRU_BW_UNITS = 16;   // hidden units per workgroup, backward

struct GruArgs {
  int T, B, H, ndir;
  const int* row_off;     // [T] packed row of (t, b) = row_off[t] + b
  const int* rev_row;     // [T*B] packed row of the reverse direction at step s, row b
  const float* gin;       // [N, ndir*3H] input projections incl. b_ih (N = sum of lengths)
  const float* wp;        // re-tiled W_hh (rnn_common.h)
  const float* bhh;       // [ndir][3H]
  float* hs;              // [2 parity][ndir] K-blocked running hidden state (fwd) / dh*z carry (bwd)
  float* dgb;             // [2 parity][ndir] K-blocked dGh of the step just processed (backward)
  float* y;               // [N, ndir*H]
  float* gates;           // [N, ndir, H, 4] (r, z, n, W_hn h + b_hn) of every unit, saved for backward:
                          //       one 16-byte store / load per (frame, unit)
  const float* hprev;     // [N, ndir*H]  h_{t-1} that entered step t (backward input)
  const float* dy;        // [N, ndir*H]
  float* dgi;             // [N, ndir*3H] gradient wrt gin  (da_r, da_z, da_n)
  float* dgh;             // [N, ndir*3H] gradient wrt the hidden projections (da_r, da_z, da_n*r)
  int step;
  int ksplit, kiter;
  int nact, nact_next;    // rows active at this step / at step + 1 (a prefix: rows are sorted)
  int row_base;           // row_off[step] from the host's copy of the lengths (no table read)
};

__device__ __forceinline__ size_t gru_row_at(const GruArgs& a, int dir, int s, int b) {
  return dir == 0 ? (size_t)(a.row_off[s] + b) : (size_t)a.rev_row[(size_t)s * a.B + b];
}
// packed row of the step being processed: only the reverse direction reads its table (see lstm.hip)
__device__ __forceinline__ int gru_row_now(const GruArgs& a, int dir, int b) {
  return dir == 0 ? a.row_base + b : a.rev_row[(size_t)a.step * a.B + b];
}

// ---- forward step ---------------------------------------------------------------------------------
// Workgroup = 4 hidden units x 3 gates (tile rows 0..11, row = gate*4 + unit; rows 12..15 of the
// re-tiled W_hh are zero) x every active batch tile (NT tiles of 16 rows per pass); the waves
// split K = H, a wave's W_hh fragments are loaded once and all operand loads are in flight before
// the first MFMA; partial tiles meet in LDS and thread (tile, row, unit) applies the cell update.
// Grid (H/4, ndir).
template <int NT>
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(GruArgs a) {
  __shared__ float P[NT][4][16][17];
  const int H = a.H, B = a.B, G3 = 3 * H;
  const int dir = blockIdx.y;
  const int j0 = blockIdx.x * GRU_FW_UNITS;
  const int par = a.step & 1;
  const size_t dsz = (size_t)B * H;
  const float* hprev = a.hs + ((size_t)par * a.ndir + dir) * dsz;
  float* hnext = a.hs + ((size_t)(par ^ 1) * a.ndir + dir) * dsz;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  const int ntiles = (a.nact + 15) >> 4;

  const int kiter = wv < a.ksplit ? a.kiter : 0;
  const int kb0 = wv * 4 * a.kiter + kg;
  const float4* wp = reinterpret_cast<const float4*>(a.wp) +
                     (((size_t)dir * (H / 4) + blockIdx.x) * (H / 4) + (kiter ? kb0 : 0)) * 16 + lr;
  const float4* hp4 = reinterpret_cast<const float4*>(hprev);

  for (int tb = 0; tb < ntiles; tb += NT) {
    f32x4 acc[NT][2];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[tt][0] = acc[tt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q = threadIdx.x >> 6, bl = (threadIdx.x >> 2) & 15, u = threadIdx.x & 3;
    const int b = (tb + q) * 16 + bl, j = j0 + u;
    const bool ew = q < NT && b < B && tb + q < ntiles;
    const bool act = ew && b < a.nact;
    const size_t sidx = ((size_t)blockIdx.x * B + (ew ? b : 0)) * 4 + u;   // blocked(b, j)
    float hp_v = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, bh0 = 0.f, bh1 = 0.f, bh2 = 0.f;
    // index first: it heads the only dependent load chain of the step (see lstm.hip)
    const int ridx = act ? gru_row_now(a, dir, b) : 0;
    __builtin_amdgcn_sched_barrier(0);
    size_t r = 0;
#pragma unroll 1
    for (int c = 0; c < kiter || c == 0; c += 8) {
      float4 bv[8], av[NT][8];
#pragma unroll
      for (int s = 0; s < 8; ++s) bv[s] = wp[(size_t)(c + s < kiter ? 4 * (c + s) : 0) * 16];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int row = (tb + tt) * 16 + lr;
        const float4* hp = hp4 + (size_t)(kiter ? kb0 : 0) * B + (row < B ? row : 0);
#pragma unroll
        for (int s = 0; s < 8; ++s) av[tt][s] = hp[(size_t)(c + s < kiter ? 4 * (c + s) : 0) * B];
      }
      if (c == 0 && ew) {
        hp_v = hprev[sidx];
        if (act) {
          r = (size_t)ridx;
          const float* gi = a.gin + r * (size_t)(a.ndir * G3) + (size_t)dir * G3 + j;
          g0 = gi[0]; g1 = gi[H]; g2 = gi[2 * H];
          const float* bh = a.bhh + (size_t)dir * G3 + j;
          bh0 = bh[0]; bh1 = bh[H]; bh2 = bh[2 * H];
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // all loads above are in flight before the first MFMA
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const bool rok = (tb + tt) * 16 + lr < B;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          float4 x = av[tt][s];
          if (!rok || c + s >= kiter) x = make_float4(0.f, 0.f, 0.f, 0.f);
          acc[tt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, bv[s].x, acc[tt][0], 0, 0, 0);
          acc[tt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, bv[s].y, acc[tt][1], 0, 0, 0);
          acc[tt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, bv[s].z, acc[tt][0], 0, 0, 0);
          acc[tt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, bv[s].w, acc[tt][1], 0, 0, 0);
        }
      }
    }
    if (tb > 0) __syncthreads();
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
      for (int e = 0; e < 4; ++e) P[tt][wv][kg * 4 + e][lr] = acc[tt][0][e] + acc[tt][1][e];
    }
    __syncthreads();
    if (ew) {
      float hn = hp_v;
      if (act) {
        const int qq = q < NT ? q : 0;
        auto proj = [&](int n) {
          return (P[qq][0][bl][n] + P[qq][1][bl][n]) + (P[qq][2][bl][n] + P[qq][3][bl][n]);
        };
        const float rg = sigmoid_acc(g0 + proj(u) + bh0);
        const float zg = sigmoid_acc(g1 + proj(4 + u) + bh1);
        const float hnp = proj(8 + u) + bh2;
        const float ng = tanh_cell(g2 + rg * hnp);
        hn = (1.f - zg) * ng + zg * hp_v;
        const size_t oh = r * (size_t)(a.ndir * H) + (size_t)dir * H + j;
        a.y[oh] = hn;
        if (a.gates) reinterpret_cast<float4*>(a.gates)[(r * a.ndir + dir) * H + j] = make_float4(rg, zg, ng, hnp);
      }
      hnext[sidx] = hn;
    }
  }
}

// ---- backward step --------------------------------------------------------------------------------
// Recurrence step s = a.step (called with s = T-1 ... 0).  For row b active at s (time t):
//   dh   = dy[t] + dGh[t_{s+1}] W_hh + carry            carry = dh_{s+1} * z_{s+1} (0 if inactive)
//   dn   = dh (1 - z)      dz = dh (h_prev - n)         new carry = dh * z
//   da_n = dn (1 - n^2)    da_r = da_n * hn_pre * r (1 - r)      da_z = dz * z (1 - z)
//   dGi[t] = (da_r, da_z, da_n)      dGh[t] = (da_r, da_z, da_n * r)
// dGh is also written K-blocked into dgb, the MFMA operand of the next launch.
// Workgroup = 16 hidden units x 16 batch rows; up to 12 waves split the K = 3H rows of W_hh^T so
// that a wave has at most 8 k-steps (16 operand loads) per chunk, all in flight at once.
__global__ __launch_bounds__(768) void gru_step_bwd_kernel(GruArgs a) {
  __shared__ float P[12][16][17];
  const int H = a.H, B = a.B, G3 = 3 * H;
  const int dir = blockIdx.y;
  const int ngroups = H / GRU_BW_UNITS;
  const int jg = blockIdx.x % ngroups;
  const int j0 = jg * GRU_BW_UNITS;
  const int b0 = (blockIdx.x / ngroups) * 16;
  const int par = a.step & 1;
  const size_t dsz = (size_t)B * H;
  const float* carry_in = a.hs + ((size_t)(par ^ 1) * a.ndir + dir) * dsz;
  float* carry_out = a.hs + ((size_t)par * a.ndir + dir) * dsz;
  const float* dgb_in = a.dgb + ((size_t)(par ^ 1) * a.ndir + dir) * 3 * dsz;
  float* dgb_out = a.dgb + ((size_t)par * a.ndir + dir) * 3 * dsz;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  const size_t ldg = (size_t)a.ndir * G3, ldh = (size_t)a.ndir * H;

  const int row = b0 + lr;
  const bool has_next = row < a.nact_next;
  const int kiter = a.kiter;                     // 3H rows / waves / 16 per step
  const int kb0 = wv * 4 * kiter + kg;
  const int KB = 3 * H / 4;
  const float4* ap = reinterpret_cast<const float4*>(dgb_in) + (size_t)kb0 * B + (row < B ? row : 0);
  const float4* wp = reinterpret_cast<const float4*>(a.wp) +
                     (((size_t)dir * ngroups + jg) * (size_t)KB + kb0) * 16 + lr;

  const int bl = (threadIdx.x >> 4) & 15, n = threadIdx.x & 15;
  const int b = b0 + bl, j = j0 + n;
  const bool ew = threadIdx.x < 256 && b < B;
  const bool act = ew && b < a.nact;
  float rg = 0.f, zg = 0.f, ng = 0.f, hnp = 0.f, hpv = 0.f, dyv = 0.f, cin = 0.f;
  const int ridx = act ? gru_row_now(a, dir, b) : 0;     // index first (see lstm.hip)
  __builtin_amdgcn_sched_barrier(0);
  size_t r = 0;

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};   // four independent chains
#pragma unroll 1
  for (int c = 0; c < kiter; c += 8) {
    float4 av[8], bv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const size_t o = c + s < kiter ? 4 * (c + s) : 0;
      av[s] = ap[o * B];
      bv[s] = wp[o * 16];
    }
    if (c == 0 && act) {
      r = (size_t)ridx;
      const float4 gs = reinterpret_cast<const float4*>(a.gates)[(r * a.ndir + dir) * H + j];
      rg = gs.x; zg = gs.y; ng = gs.z; hnp = gs.w;
      const size_t oh = r * ldh + (size_t)dir * H + j;
      hpv = a.hprev[oh];
      dyv = a.dy[oh];
      cin = carry_in[(size_t)b * H + j];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float4 x = av[s];
      if (!has_next || c + s >= kiter) x = make_float4(0.f, 0.f, 0.f, 0.f);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, bv[s].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, bv[s].y, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, bv[s].z, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, bv[s].w, acc3, 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) P[wv][kg * 4 + q][lr] = (acc0[q] + acc1[q]) + (acc2[q] + acc3[q]);
  __syncthreads();
  if (ew) {
    float carry = 0.f;
    if (act) {
      float dhr = 0.f;
      for (int w = 0; w < a.ksplit; ++w) dhr += P[w][bl][n];
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
      dgb_out[blocked(b, j, B)] = dar;
      dgb_out[blocked(b, H + j, B)] = daz;
      dgb_out[blocked(b, 2 * H + j, B)] = dan * rg;
      carry = dh * zg;
    }
    carry_out[(size_t)b * H + j] = carry;
  }
}

}  // namespace itts

using namespace itts;

// d_state: [hs / carry 2*ndir*B*H | dgb 2*ndir*B*3H | re-tiled W_hh ndir*4H*H (fwd pads the 4th gate)]
extern "C" int64_t itts_gru_state_bytes(int B, int H, int ndir) {
  if (B <= 0 || H <= 0 || ndir <= 0) return 0;
  return ((int64_t)8 * ndir * B * H + (int64_t)ndir * 4 * H * H) * 4;
}

extern "C" int itts_gru_layer_fwd(const float* d_gin, const float* d_whh, const float* d_bhh,
                                  const float* d_h0, const int* d_lengths, const int* h_lengths,
                                  const int* d_row_off, const int* d_rev_row, int T, int B, int H,
                                  int ndir, float* d_y, float* d_gates, float* d_hn, void* d_state,
                                  void* stream) {
  ITTS_REQUIRE(d_gin && d_whh && d_bhh && d_lengths && d_row_off && d_y && d_state, "null pointer");
  ITTS_REQUIRE(ndir == 1 || d_rev_row, "the reverse direction needs its row table");
  int rc = rnn_check(h_lengths, T, B, H, ndir);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  {
    RnnPersistArgs p{};
    p.gin = d_gin; p.whh = d_whh; p.bhh = d_bhh; p.h0 = d_h0; p.lengths = d_lengths; p.row_off = d_row_off;
    p.rev_row = d_rev_row; p.y = d_y; p.gates = d_gates; p.hn = d_hn;
    p.T = T; p.B = B; p.ndir = ndir;
    const int done = rnn_persist_forward<3>(p, H, s);      // rnn_persist.h
    if (done < 0) return ITTS_E_HIP;
    if (done) return ITTS_OK;
  }
  GruArgs a{};
  a.T = T; a.B = B; a.H = H; a.ndir = ndir; a.row_off = d_row_off; a.rev_row = d_rev_row; a.gin = d_gin;
  a.bhh = d_bhh; a.y = d_y; a.gates = d_gates;
  const size_t st = (size_t)2 * ndir * B * H;
  a.hs = reinterpret_cast<float*>(d_state);
  float* wp = a.hs + 4 * st;
  a.wp = wp;
  const int64_t n = (int64_t)ndir * B * H;
  hipLaunchKernelGGL(rnn_pack_w_fwd_kernel, rnn_ew_grid((int64_t)ndir * H * H), dim3(256), 0, s, d_whh, wp, ndir, 3, H);
  hipLaunchKernelGGL(rnn_init_state_kernel, rnn_ew_grid(n), dim3(256), 0, s, d_h0, a.hs, ndir, B, H);
  ITTS_LAUNCH_CHECK();
  a.ksplit = (H % 64 == 0) ? 4 : ((H % 32 == 0) ? 2 : 1);
  a.kiter = H / (16 * a.ksplit);
  const dim3 grid(H / GRU_FW_UNITS, ndir);
  int p = B;
  int row_base = 0;
  for (int step = 0; step < T; ++step) {
    a.step = step;
    a.nact = rnn_active_rows(h_lengths, B, step, &p);
    a.row_base = row_base;          // row_off[step] = rows active in all earlier steps
    row_base += a.nact;
    switch (std::min((a.nact + 15) / 16, 4)) {
      case 1: hipLaunchKernelGGL(gru_step_fwd_kernel<1>, grid, dim3(256), 0, s, a); break;
      case 2: hipLaunchKernelGGL(gru_step_fwd_kernel<2>, grid, dim3(256), 0, s, a); break;
      case 3: hipLaunchKernelGGL(gru_step_fwd_kernel<3>, grid, dim3(256), 0, s, a); break;
      default: hipLaunchKernelGGL(gru_step_fwd_kernel<4>, grid, dim3(256), 0, s, a); break;
    }
  }
  ITTS_LAUNCH_CHECK();
  if (d_hn) {
    hipLaunchKernelGGL(rnn_final_state_kernel, rnn_ew_grid(n), dim3(256), 0, s, a.hs, d_lengths, d_hn, ndir, B, H);
    ITTS_LAUNCH_CHECK();
  }
  return ITTS_OK;
}

extern "C" int itts_gru_layer_bwd(const float* d_dy, const float* d_whh, const float* d_gates,
                                  const float* d_hprev, const int* h_lengths,
                                  const int* d_row_off, const int* d_rev_row, int T, int B, int H,
                                  int ndir, float* d_dgi, float* d_dgh, float* d_dh0, void* d_state,
                                  void* stream) {
  ITTS_REQUIRE(d_dy && d_whh && d_gates && d_hprev && d_row_off && d_dgi && d_dgh && d_state, "null pointer");
  ITTS_REQUIRE(ndir == 1 || d_rev_row, "the reverse direction needs its row table");
  int rc = rnn_check(h_lengths, T, B, H, ndir);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  {
    RnnPersistBwdArgs p{};
    p.dy = d_dy; p.whh = d_whh; p.gates = d_gates; p.aux = d_hprev; p.row_off = d_row_off;
    p.rev_row = d_rev_row; p.dg = d_dgi; p.dg2 = d_dgh; p.d0 = d_dh0; p.T = T; p.B = B; p.ndir = ndir;
    const int done = rnn_persist_backward<3>(p, h_lengths, H, s);     // rnn_persist.h
    if (done < 0) return ITTS_E_HIP;
    if (done) return ITTS_OK;
  }
  GruArgs a{};
  a.T = T; a.B = B; a.H = H; a.ndir = ndir; a.row_off = d_row_off; a.rev_row = d_rev_row;
  a.gates = const_cast<float*>(d_gates); a.hprev = d_hprev; a.dy = d_dy; a.dgi = d_dgi; a.dgh = d_dgh;
  const size_t st = (size_t)2 * ndir * B * H;
  a.hs = reinterpret_cast<float*>(d_state);
  a.dgb = a.hs + st;
  float* wp = a.hs + 4 * st;
  a.wp = wp;
  hipLaunchKernelGGL(rnn_pack_w_bwd_kernel, rnn_ew_grid((int64_t)ndir * H * H), dim3(256), 0, s, d_whh, wp, ndir, 3, H);
  ITTS_LAUNCH_CHECK();
  ITTS_HIP_CHECK(hipMemsetAsync(a.hs, 0, st * 4, s));   // carry of not-yet-active rows
  a.ksplit = (H % 64 == 0) ? 12 : ((H % 32 == 0) ? 6 : 3);   // waves per workgroup; 3H/16 k-steps in all
  a.kiter = (3 * H / 16) / a.ksplit;
  int p = 0, nact_next = 0;
  std::vector<int> row_off(T + 1, 0);      // host copy of the packed-row offsets
  {
    int q = B;
    for (int t = 0; t < T; ++t) row_off[t + 1] = row_off[t] + rnn_active_rows(h_lengths, B, t, &q);
  }
  for (int step = T - 1; step >= 0; --step) {
    a.step = step;
    a.nact = rnn_active_rows(h_lengths, B, step, &p);
    a.nact_next = nact_next;
    nact_next = a.nact;
    a.row_base = row_off[step];
    hipLaunchKernelGGL(gru_step_bwd_kernel, dim3((H / GRU_BW_UNITS) * ((a.nact + 15) / 16), ndir),
                       dim3(64 * a.ksplit), 0, s, a);
  }
  ITTS_LAUNCH_CHECK();
  // step 0 (all rows active) leaves dh * z in the parity-0 carry buffer [ndir][B][H]
  if (d_dh0)
    ITTS_HIP_CHECK(hipMemcpyAsync(d_dh0, a.hs, (size_t)ndir * B * H * 4, hipMemcpyDeviceToDevice, s));
  return ITTS_OK;
}
