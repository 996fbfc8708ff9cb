// (Bi)LSTM recurrence for padded batches of frame sequences -- the recurrent half of the
// acoustic model of BASELINE config 3 (3 x 512 BiLSTM).  Replaces what torch.nn.LSTM does between
// pack_padded_sequence(enforce_sorted=False) and pad_packed_sequence in
// rnn_dyn/RNNWrapper.py:45-107 (cuDNN / MIOpen RNN in the reference).
//
// Split of the work (per layer):
//   * gin = X W_ih^T + (b_ih + b_hh) for ALL time steps and both directions is ONE fp32-MFMA GEMM
//     (nn.hip), so are dX, dW_ih, dW_hh and the bias gradients in the backward pass;
//   * only the true recurrence h_{t-1} W_hh^T runs per time step.  Forward and backward: one
//     persistent launch per layer each where it applies (rnn_persist.h: H = 512, every recurrence
//     inside one XCD).  Otherwise: one launch per step, both directions in it: the kernel
//     boundary is the grid-wide dependency (an in-kernel grid barrier ACROSS XCDs costs 2.3-2.5 us
//     on this chip, scripts/handoff_lab).  A workgroup owns a slice of hidden units and
//     streams its W_hh rows and h_{t-1} [B, H] from L2 through v_mfma_f32_16x16x4_f32
//     (exact fp32, same K-permutation trick as the GEMM: one 16-byte load feeds 4 MFMAs).
//   * packed-sequence semantics: row b is active for step s < len_b; the forward direction
//     visits t = s, the reverse direction t = len_b - 1 - s (it starts at each sequence's own
//     last frame); the state of an inactive row is frozen.
//   * packed row layout (what pack_padded_sequence produces): the batch rows are sorted by
//     decreasing length, frame t of row b lives at packed row row_off[t] + b, where
//     row_off[t] = sum_{t' < t} nact(t') and nact(t) = #{b : len_b > t}.  Only valid frames exist,
//     so the GEMMs around the recurrence touch N = sum(len) rows instead of T*B, and step s
//     works on the ceil(nact(s) / 16) batch tiles that still have active rows.  The reverse
//     direction's row at step s, row_off[len_b - 1 - s] + b, comes from a table built once per
//     batch (rev_row[s][b]) so that no step chases lengths -> offsets -> data through memory.
//   * a step is latency bound (one dependent pass over ~10 MB that the previous launch left cold
//     in L2): every operand load of a wave is issued before the first MFMA
//     (__builtin_amdgcn_sched_barrier keeps the scheduler from pairing loads with their MFMAs,
//     which costs ~0.2 us per load when it happens), and every operand load is a contiguous 1 KB
//     wave access thanks to the K-blocked state / re-tiled W_hh layouts of rnn_common.h.
// Gate order i, f, g, o and the two bias vectors follow torch.nn.LSTM.
#include <atomic>
#include <cstdio>
#include <vector>

#include "rnn_common.h"
#include "rnn_persist.h"

namespace itts {

constexpr int FW_UNITS = 4;    // hidden units per workgroup in the forward step (16 gate rows)
constexpr int BW_UNITS = 16;   // hidden units per workgroup in the backward step

struct LstmArgs {
  // geometry
  int T, B, H, ndir;
  const int* row_off;     // [T] device: packed row of (t, b) = row_off[t] + b
  const int* rev_row;     // [T*B] device: packed row the reverse direction visits at step s for
                          //       row b, row_off[len_b - 1 - s] + b (unused where s >= len_b)
  const float* gin;       // [N, ndir*4H] input projections incl. both biases (N = sum of lengths)
  const float* wp;        // re-tiled W_hh (rnn_pack_w_fwd_kernel / rnn_pack_w_bwd_kernel)
  const float* c0;        // [ndir][H] or NULL
  float* hs;              // [2 parity][ndir] K-blocked running hidden state
  float* cs;              // [2 parity][ndir] K-blocked running cell state / running dc (backward)
  float* dgb;             // [2 parity][ndir] K-blocked dG of the step just processed (backward)
  float* y;               // [N, ndir*H] layer output
  float* gates;           // [N, ndir, H, 4] post-activation gates (i, f, g, o) of every unit, saved
                          //       for backward: one 16-byte store / load per (frame, unit)
  float* csave;           // [N, ndir*H] c_t
  // backward
  const float* dy;        // [N, ndir*H]
  float* dg;              // [N, ndir*4H] gradient wrt pre-activation gates
  int step;
  int ksplit, kiter;      // K is split over `ksplit` waves, `kiter` steps of 16 k each
  int nact, nact_next;    // rows active at this step / at step + 1 (a prefix: rows are sorted)
  int row_base;           // row_off[step], from the host's copy of the lengths (no table read)
  int row_base_prev;      // row_off[step - 1] (backward: c_{t-1} of the forward direction)
};

// packed row that row b visits at recurrence step s (the caller knows that it is active)
__device__ __forceinline__ size_t row_at(const LstmArgs& a, int dir, int s, int b) {
  return dir == 0 ? (size_t)(a.row_off[s] + b) : (size_t)a.rev_row[(size_t)s * a.B + b];
}
// the same for the step being processed / the one before it: the forward direction's rows follow
// from the host-side offsets, only the reverse direction reads its table
__device__ __forceinline__ int row_now(const LstmArgs& a, int dir, int b) {
  return dir == 0 ? a.row_base + b : a.rev_row[(size_t)a.step * a.B + b];
}
__device__ __forceinline__ int row_before(const LstmArgs& a, int dir, int b) {
  return dir == 0 ? a.row_base_prev + b : a.rev_row[(size_t)(a.step - 1) * a.B + b];
}

// ---- forward step -----------------------------------------------------------------------------------
// Workgroup = 4 hidden units (16 gate rows of W_hh) x every active batch tile.  Its 4 waves split
// K = H four ways; a wave loads its W_hh fragments once, then the h_{t-1} fragments of NT batch
// tiles of 16 rows, straight from L2 into registers (every element feeds exactly one MFMA, so
// there is no LDS staging).  The partial 16x16 tiles are reduced through LDS and thread
// (tile, row, unit) applies the cell update.
// Grid: (H/4, ndir) -> 256 workgroups for H = 512; W_hh traffic does not grow with the batch.
template <int NT>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(LstmArgs a) {
  __shared__ float P[NT][4][16][17];
  const int H = a.H, B = a.B, G4 = 4 * H;
  const int dir = blockIdx.y;
  const int j0 = blockIdx.x * FW_UNITS;
  const int par = a.step & 1;
  const size_t dsz = (size_t)B * H;
  const float* hprev = a.hs + ((size_t)par * a.ndir + dir) * dsz;
  const float* cprev = a.cs + ((size_t)par * a.ndir + dir) * dsz;
  float* hnext = a.hs + ((size_t)(par ^ 1) * a.ndir + dir) * dsz;
  float* cnext = a.cs + ((size_t)(par ^ 1) * a.ndir + dir) * dsz;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  const int ntiles = (a.nact + 15) >> 4;

  // K split over a.ksplit waves, a.kiter steps of 16 k (4 k-blocks) each, in chunks of 8 steps
  const int kiter = wv < a.ksplit ? a.kiter : 0;
  const int kb0 = wv * 4 * a.kiter + kg;                       // this lane's first k-block
  const float4* wp = reinterpret_cast<const float4*>(a.wp) +
                     (((size_t)dir * (H / 4) + blockIdx.x) * (H / 4) + (kiter ? kb0 : 0)) * 16 + lr;
  const float4* hp4 = reinterpret_cast<const float4*>(hprev);

  for (int tb = 0; tb < ntiles; tb += NT) {
    f32x4 acc[NT][4];      // four accumulator chains per tile: an MFMA never waits for its predecessor
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
      acc[tt][0] = acc[tt][1] = acc[tt][2] = acc[tt][3] = f32x4{0.f, 0.f, 0.f, 0.f};
    // elementwise operands of thread (tile q, batch row bl, unit u); their loads are issued behind
    // the first chunk's operand loads (loads return in order)
    const int q = threadIdx.x >> 6, bl = (threadIdx.x >> 2) & 15, u = threadIdx.x & 3;
    const int b = (tb + q) * 16 + bl, j = j0 + u;
    const bool ew = q < NT && b < B && tb + q < ntiles;   // rows of a launched tile
    const bool act = ew && b < a.nact;
    const size_t sidx = ((size_t)blockIdx.x * B + (ew ? b : 0)) * 4 + u;   // blocked(b, j)
    float hp_v = 0.f, cp_v = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    // The packed-row index heads the only dependent load chain of a step (index -> gin row): it
    // is requested before the 40 operand loads, which then cover its latency, and the gin loads
    // that need it are covered by the MFMAs (measured before: the first MFMA waited ~7 000 clocks
    // for index + gin issued behind each other in front of it).
    const int ridx = act ? row_now(a, dir, b) : 0;
    // ... and the table row of the NEXT step is pulled into this XCD's L2 now (256 B that every
    // workgroup of the reverse direction needs: whichever XCD it lands on next time finds it there;
    // cold, that one load was ~6 000 clocks at the head of every step)
    int pf = 0;
    if (tb == 0 && dir == 1 && a.step + 1 < a.T && (int)threadIdx.x < B)
      pf = a.rev_row[(size_t)(a.step + 1) * B + threadIdx.x];
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
        cp_v = cprev[sidx];
      }
      __builtin_amdgcn_sched_barrier(0);   // all loads above are in flight before the first MFMA
      // tile-outermost order: the MFMAs of tile 0 start as soon as ITS operands are there, while
      // the loads of the later tiles are still in flight (k-step-outermost, with 2 * NT independent
      // accumulator chains, measured 9 % slower)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const bool rok = (tb + tt) * 16 + lr < B;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          float4 x = av[tt][s];
          if (!rok || c + s >= kiter) x = make_float4(0.f, 0.f, 0.f, 0.f);
          acc[tt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, bv[s].x, acc[tt][0], 0, 0, 0);
          acc[tt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, bv[s].y, acc[tt][1], 0, 0, 0);
          acc[tt][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, bv[s].z, acc[tt][2], 0, 0, 0);
          acc[tt][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, bv[s].w, acc[tt][3], 0, 0, 0);
        }
        if (tt == 0 && c == 0) {
          // the input projections of this element are requested here: the index (the oldest
          // outstanding load) is back once tile 0's operands are, and the remaining tiles' MFMAs
          // cover the latency of these loads
          __builtin_amdgcn_sched_barrier(0);
          if (act) {
            r = (size_t)ridx;
            const float* gi = a.gin + r * (size_t)(a.ndir * G4) + (size_t)dir * G4 + j;
            g0 = gi[0]; g1 = gi[H]; g2 = gi[2 * H]; g3 = gi[3 * H];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (tb > 0) __syncthreads();   // P of the previous group has been consumed
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        P[tt][wv][kg * 4 + e][lr] = (acc[tt][0][e] + acc[tt][1][e]) + (acc[tt][2][e] + acc[tt][3][e]);
    }
    __syncthreads();
    if (ew) {
      float hn = hp_v, cn = cp_v;
      if (act) {
        const int qq = q < NT ? q : 0;
        auto gate = [&](int n) {
          return (P[qq][0][bl][n] + P[qq][1][bl][n]) + (P[qq][2][bl][n] + P[qq][3][bl][n]);
        };
        const float ig = sigmoid_acc(gate(u) + g0), fg = sigmoid_acc(gate(4 + u) + g1),
                    gg = tanh_cell(gate(8 + u) + g2), og = sigmoid_acc(gate(12 + u) + g3);
        cn = fg * cp_v + ig * gg;
        hn = og * tanh_cell(cn);
        a.y[r * (size_t)(a.ndir * H) + (size_t)dir * H + j] = hn;
        if (a.gates) {
          reinterpret_cast<float4*>(a.gates)[(r * a.ndir + dir) * H + j] = make_float4(ig, fg, gg, og);
          a.csave[r * (size_t)(a.ndir * H) + (size_t)dir * H + j] = cn;
        }
      }
      hnext[sidx] = hn;
      cnext[sidx] = cn;
    }
    if (pf == 0x7fffffff) hnext[0] = 0.f;    // never true: keeps the prefetch load alive
  }
}

// ---- backward step ----------------------------------------------------------------------------------
// Processes recurrence step s = a.step (called with s = T-1 ... 0). For row b active at s:
//   dh = dy[t] + dG[t_{s+1}] W_hh   (second term only if the row is active at s+1)
//   standard LSTM cell gradients -> dG[t], running dc (cs buffers, parity by step)
// dG of a step is written twice: row-major into dg (for the dW / dX GEMMs) and K-blocked into dgb,
// which is what the next launch reads as its MFMA operand.
// Workgroup = 16 hidden units x 16 batch rows; up to 16 waves split the K = 4H gate rows so that
// a wave has at most 8 k-steps (16 operand loads) per chunk, all in flight at once.
// Grid (H/16 * ceil(nact/16), ndir).
__global__ __launch_bounds__(1024) void lstm_step_bwd_kernel(LstmArgs a) {
  __shared__ float P[16][16][17];
  const int H = a.H, B = a.B, G4 = 4 * H;
  const int dir = blockIdx.y;
  const int ngroups = H / BW_UNITS;
  const int jg = blockIdx.x % ngroups;
  const int j0 = jg * BW_UNITS;
  const int b0 = (blockIdx.x / ngroups) * 16;
  const int par = a.step & 1;
  const size_t dsz = (size_t)B * H;
  const float* dc_in = a.cs + ((size_t)(par ^ 1) * a.ndir + dir) * dsz;
  float* dc_out = a.cs + ((size_t)par * a.ndir + dir) * dsz;
  const float* dgb_in = a.dgb + ((size_t)(par ^ 1) * a.ndir + dir) * 4 * dsz;
  float* dgb_out = a.dgb + ((size_t)par * a.ndir + dir) * 4 * dsz;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  const size_t ldg = (size_t)a.ndir * G4, ldh = (size_t)a.ndir * H;

  // dh_rec tile: A = dG of the next recurrence step (rows b0..b0+15, K-blocked), B = W_hh^T rows
  // of our 16 units (re-tiled)
  const int row = b0 + lr;
  const bool has_next = row < a.nact_next;
  const int kiter = a.kiter;                     // 4H gate rows / waves / 16 per step
  const int kb0 = wv * 4 * kiter + kg;
  const float4* ap = reinterpret_cast<const float4*>(dgb_in) + (size_t)kb0 * B + (row < B ? row : 0);
  const float4* wp = reinterpret_cast<const float4*>(a.wp) +
                     (((size_t)dir * ngroups + jg) * (size_t)H + kb0) * 16 + lr;   // K/4 = H blocks

  // elementwise operands (thread -> batch row bl, unit n)
  const int bl = (threadIdx.x >> 4) & 15, n = threadIdx.x & 15;
  const int b = b0 + bl, j = j0 + n;
  const bool ew = threadIdx.x < 256 && b < B;
  const bool act = ew && b < a.nact;
  float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, ct = 0.f, cp = 0.f, dyv = 0.f, dcin = 0.f;
  // packed-row indices first (see the forward kernel): the saved tensors they address are then
  // requested behind the operand loads, under the MFMAs
  const int ridx = act ? row_now(a, dir, b) : 0;
  const int rpidx = (act && a.step > 0) ? row_before(a, dir, b) : 0;
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
      ig = gs.x; fg = gs.y; gg = gs.z; og = gs.w;
      ct = a.csave[r * ldh + (size_t)dir * H + j];
      cp = a.step > 0 ? a.csave[(size_t)rpidx * ldh + (size_t)dir * H + j]
                      : (a.c0 ? a.c0[dir * H + j] : 0.f);
      dyv = a.dy[r * ldh + (size_t)dir * H + j];
      dcin = dc_in[(size_t)b * H + j];
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
    float dc_keep = 0.f;
    if (act) {
      float dhr = 0.f;
      for (int w = 0; w < a.ksplit; ++w) dhr += P[w][bl][n];
      const float tc = tanh_cell(ct);
      const float dh = dyv + dhr;
      const float dcv = dh * og * (1.f - tc * tc) + dcin;
      const float d0 = dcv * gg * ig * (1.f - ig), d1 = dcv * cp * fg * (1.f - fg),
                  d2 = dcv * ig * (1.f - gg * gg), d3 = dh * tc * og * (1.f - og);
      float* dgo = a.dg + r * ldg + (size_t)dir * G4 + j;
      dgo[0] = d0; dgo[H] = d1; dgo[2 * H] = d2; dgo[3 * H] = d3;
      dgb_out[blocked(b, j, B)] = d0;
      dgb_out[blocked(b, H + j, B)] = d1;
      dgb_out[blocked(b, 2 * H + j, B)] = d2;
      dgb_out[blocked(b, 3 * H + j, B)] = d3;
      dc_keep = dcv * fg;
    }
    dc_out[(size_t)b * H + j] = dc_keep;
  }
}


}  // namespace itts

using namespace itts;

// d_state: [hs 2*ndir*B*H | cs 2*ndir*B*H | dgb 2*ndir*B*4H | re-tiled W_hh ndir*4H*H] floats
extern "C" int64_t itts_lstm_state_bytes(int B, int H, int ndir) {
  if (B <= 0 || H <= 0 || ndir <= 0) return 0;
  return ((int64_t)12 * ndir * B * H + (int64_t)ndir * 4 * H * H) * 4;
}

// Runs the recurrence of one (bi)directional LSTM layer over T steps (packed rows, see the top).
extern "C" int itts_lstm_layer_fwd(const float* d_gin, const float* d_whh, const float* d_h0,
                                   const float* d_c0, const int* d_lengths, const int* h_lengths,
                                   const int* d_row_off, const int* d_rev_row, int T, int B, int H,
                                   int ndir, float* d_y, float* d_gates, float* d_csave,
                                   float* d_hn, float* d_cn, void* d_state, void* stream) {
  ITTS_REQUIRE(d_gin && d_whh && d_lengths && d_row_off && d_y && d_state, "null pointer");
  ITTS_REQUIRE(ndir == 1 || d_rev_row, "the reverse direction needs its row table");
  ITTS_REQUIRE((d_gates == nullptr) == (d_csave == nullptr),
               "gates / csave must be given together (training) or both NULL (inference)");
  int rc = rnn_check(h_lengths, T, B, H, ndir);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  {
    RnnPersistArgs p{};
    p.gin = d_gin; p.whh = d_whh; p.h0 = d_h0; p.c0 = d_c0; p.lengths = d_lengths; p.row_off = d_row_off;
    p.rev_row = d_rev_row; p.y = d_y; p.gates = d_gates; p.csave = d_csave; p.hn = d_hn; p.cn = d_cn;
    p.T = T; p.B = B; p.ndir = ndir;
    const int done = rnn_persist_forward<4>(p, H, s);      // rnn_common.h
    if (done < 0) return ITTS_E_HIP;
    if (done) return ITTS_OK;
  }
  LstmArgs a{};
  a.T = T; a.B = B; a.H = H; a.ndir = ndir; a.row_off = d_row_off; a.rev_row = d_rev_row; a.gin = d_gin;
  a.y = d_y; a.gates = d_gates; a.csave = d_csave;
  const size_t st = (size_t)2 * ndir * B * H;
  a.hs = reinterpret_cast<float*>(d_state);
  a.cs = a.hs + st;
  float* wp = a.hs + 6 * st;
  a.wp = wp;
  const int64_t n = (int64_t)ndir * B * H;
  hipLaunchKernelGGL(rnn_pack_w_fwd_kernel, rnn_ew_grid((int64_t)ndir * H * H), dim3(256), 0, s, d_whh, wp, ndir, 4, H);
  hipLaunchKernelGGL(rnn_init_state_kernel, rnn_ew_grid(n), dim3(256), 0, s, d_h0, a.hs, ndir, B, H);
  hipLaunchKernelGGL(rnn_init_state_kernel, rnn_ew_grid(n), dim3(256), 0, s, d_c0, a.cs, ndir, B, H);
  ITTS_LAUNCH_CHECK();
  a.ksplit = (H % 64 == 0) ? 4 : ((H % 32 == 0) ? 2 : 1);
  a.kiter = H / (16 * a.ksplit);
  const dim3 grid(H / FW_UNITS, ndir);
  int p = B;
  int row_base = 0;
  for (int step = 0; step < T; ++step) {
    a.step = step;
    a.nact = rnn_active_rows(h_lengths, B, step, &p);
    a.row_base = row_base;          // row_off[step] = rows active in all earlier steps
    row_base += a.nact;
    switch (std::min((a.nact + 15) / 16, 4)) {
      case 1: hipLaunchKernelGGL(lstm_step_fwd_kernel<1>, grid, dim3(256), 0, s, a); break;
      case 2: hipLaunchKernelGGL(lstm_step_fwd_kernel<2>, grid, dim3(256), 0, s, a); break;
      case 3: hipLaunchKernelGGL(lstm_step_fwd_kernel<3>, grid, dim3(256), 0, s, a); break;
      default: hipLaunchKernelGGL(lstm_step_fwd_kernel<4>, grid, dim3(256), 0, s, a); break;
    }
  }
  ITTS_LAUNCH_CHECK();
  if (d_hn) hipLaunchKernelGGL(rnn_final_state_kernel, rnn_ew_grid(n), dim3(256), 0, s, a.hs, d_lengths, d_hn, ndir, B, H);
  if (d_cn) hipLaunchKernelGGL(rnn_final_state_kernel, rnn_ew_grid(n), dim3(256), 0, s, a.cs, d_lengths, d_cn, ndir, B, H);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

// Backward recurrence: fills d_dg [N, ndir*4H] from d_dy and the saved forward tensors.
extern "C" int itts_lstm_layer_bwd(const float* d_dy, const float* d_whh, const float* d_c0,
                                   const float* d_gates, const float* d_csave, const int* h_lengths,
                                   const int* d_row_off, const int* d_rev_row, int T, int B, int H,
                                   int ndir, float* d_dg, float* d_dc0, void* d_state,
                                   void* stream) {
  ITTS_REQUIRE(d_dy && d_whh && d_gates && d_csave && d_row_off && d_dg && d_state, "null pointer");
  ITTS_REQUIRE(ndir == 1 || d_rev_row, "the reverse direction needs its row table");
  int rc = rnn_check(h_lengths, T, B, H, ndir);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  {
    RnnPersistBwdArgs p{};
    p.dy = d_dy; p.whh = d_whh; p.c0 = d_c0; p.gates = d_gates; p.aux = d_csave; p.row_off = d_row_off;
    p.rev_row = d_rev_row; p.dg = d_dg; p.d0 = d_dc0; p.T = T; p.B = B; p.ndir = ndir;
    const int done = rnn_persist_backward<4>(p, h_lengths, H, s);     // rnn_persist.h
    if (done < 0) return ITTS_E_HIP;
    if (done) return ITTS_OK;
  }
  LstmArgs a{};
  a.T = T; a.B = B; a.H = H; a.ndir = ndir; a.row_off = d_row_off; a.rev_row = d_rev_row;
  a.c0 = d_c0; a.gates = const_cast<float*>(d_gates); a.csave = const_cast<float*>(d_csave); a.dy = d_dy;
  a.dg = d_dg;
  const size_t st = (size_t)2 * ndir * B * H;
  a.hs = reinterpret_cast<float*>(d_state);
  a.cs = a.hs + st;
  a.dgb = a.hs + 2 * st;
  float* wp = a.hs + 6 * st;
  a.wp = wp;
  hipLaunchKernelGGL(rnn_pack_w_bwd_kernel, rnn_ew_grid((int64_t)ndir * H * H), dim3(256), 0, s, d_whh, wp, ndir, 4, H);
  ITTS_LAUNCH_CHECK();
  ITTS_HIP_CHECK(hipMemsetAsync(a.cs, 0, st * 4, s));   // running dc of rows that are not active yet
  a.ksplit = (H % 64 == 0) ? 16 : ((H % 32 == 0) ? 8 : 4);   // waves per workgroup
  a.kiter = H / (4 * a.ksplit);
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
    a.row_base_prev = step > 0 ? row_off[step - 1] : 0;
    hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3((H / BW_UNITS) * ((a.nact + 15) / 16), ndir),
                       dim3(64 * a.ksplit), 0, s, a);
  }
  ITTS_LAUNCH_CHECK();
  // step 0 has every row active and leaves dc * f, the gradient of the initial cell state, in the
  // parity-0 carry buffer [ndir][B][H]
  if (d_dc0)
    ITTS_HIP_CHECK(hipMemcpyAsync(d_dc0, a.cs, (size_t)ndir * B * H * 4, hipMemcpyDeviceToDevice, s));
  return ITTS_OK;
}
