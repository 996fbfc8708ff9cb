// (Bi)LSTM recurrence for padded batches of frame sequences -- the recurrent half of the
// acoustic model of BASELINE config 3 (3 x 512 BiLSTM).  Replaces what torch.nn.LSTM does between
// pack_padded_sequence(enforce_sorted=False) and pad_packed_sequence in
// rnn_dyn/RNNWrapper.py:45-107 (cuDNN / MIOpen RNN in the reference).
//
// Split of the work (per layer):
//   * gin = X W_ih^T + (b_ih + b_hh) for ALL time steps and both directions is ONE fp32-MFMA GEMM
//     (nn.hip), so are dX, dW_ih, dW_hh and the bias gradients in the backward pass;
//   * only the true recurrence h_{t-1} W_hh^T runs per time step.  One launch per step, both
//     directions in it: the kernel boundary is the grid-wide dependency (~1.5 us; an in-kernel
//     grid barrier costs 4-7 us on this chip).  A workgroup owns a slice of hidden units, keeps
//     its W_hh rows in LDS and streams h_{t-1} [B, H] from L2 through v_mfma_f32_16x16x4_f32
//     (exact fp32, same K-permutation trick as the GEMM: one 16-byte load feeds 4 MFMAs).
//   * packed-sequence semantics: row b is active for step s < len_b; the forward direction
//     visits t = s, the reverse direction t = len_b - 1 - s (it starts at each sequence's own
//     last frame); the state of an inactive row is frozen.
//   * packed row layout (what pack_padded_sequence produces): the batch rows are sorted by
//     decreasing length, frame t of row b lives at packed row row_off[t] + b, where
//     row_off[t] = sum_{t' < t} nact(t') and nact(t) = #{b : len_b > t}.  Only valid frames exist,
//     so the GEMMs around the recurrence touch N = sum(len) rows instead of T*B, and step s
//     launches workgroups for the ceil(nact(s) / 16) batch tiles that still have active rows.
// Gate order i, f, g, o and the two bias vectors follow torch.nn.LSTM.
#include <algorithm>

#include "common.h"

namespace itts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FW_UNITS = 4;    // hidden units per workgroup in the forward step (16 gate rows)
constexpr int BW_UNITS = 16;   // hidden units per workgroup in the backward step

struct LstmArgs {
  // geometry
  int T, B, H, ndir;
  const int* lengths;     // [B] device, sorted by decreasing length
  const int* row_off;     // [T] device: packed row of (t, b) = row_off[t] + b
  const float* gin;       // [N, ndir*4H] input projections incl. both biases (N = sum of lengths)
  const float* whh;       // [ndir][4H][H]
  const float* whh_t;     // [ndir][H][4H]   (backward)
  const float* h0;        // [ndir][H] initial state (broadcast over the batch)
  const float* c0;
  float* hs;              // [2 parity][ndir][B][H] running hidden state
  float* cs;              // [2 parity][ndir][B][H] running cell state / running dc (backward)
  float* y;               // [N, ndir*H] layer output
  float* gates;           // [N, ndir*4H] post-activation gates i,f,g,o (saved for backward)
  float* csave;           // [N, ndir*H] c_t
  float* hprev;           // [N, ndir*H] h_{t-1} that entered step t (for dW_hh)
  // backward
  const float* dy;        // [N, ndir*H]
  float* dg;              // [N, ndir*4H] gradient wrt pre-activation gates
  int step;
  int ksplit, kiter;  // forward: K = H split over `ksplit` waves, `kiter` steps of 16 each
};

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.f / (1.f + expf(-x)); }

// time index visited by row b at recurrence step s, or -1 when the row is inactive
__device__ __forceinline__ int time_of(int dir, int s, int len) {
  if (s >= len) return -1;
  return dir == 0 ? s : len - 1 - s;
}

// ---- forward step -----------------------------------------------------------------------------------
// Workgroup = (4 hidden units = 16 gate rows) x (16 batch rows); its 4 waves split K = H four
// ways, every wave streams its A (h_{t-1}) and B (W_hh) fragments straight from L2 into registers
// (all loads of the step in flight at once, no LDS staging: each element is used by exactly one
// MFMA), the partial 16x16 tiles are reduced through LDS and 64 threads apply the cell update.
// Grid: (H/4 * ceil(nact/16), ndir) -> 1024 workgroups for H = 512 while all 64 rows are active.
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(LstmArgs a) {
  __shared__ float P[4][16][17];
  const int H = a.H, B = a.B, G4 = 4 * H;
  const int dir = blockIdx.y;
  const int ngroups = H / FW_UNITS;
  const int j0 = (blockIdx.x % ngroups) * FW_UNITS;
  const int b0 = (blockIdx.x / ngroups) * 16;
  const int par = a.step & 1;
  const float* whh = a.whh + (size_t)dir * G4 * H;
  const float* hprev = a.hs + ((size_t)par * a.ndir + dir) * B * H;
  const float* cprev = a.cs + ((size_t)par * a.ndir + dir) * B * H;
  float* hnext = a.hs + ((size_t)(par ^ 1) * a.ndir + dir) * B * H;
  float* cnext = a.cs + ((size_t)(par ^ 1) * a.ndir + dir) * B * H;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;

  // elementwise operands of this workgroup's 16 rows x 4 units: issue their loads first
  const int bl = threadIdx.x >> 2, u = threadIdx.x & 3;
  const int b = b0 + bl, j = j0 + u;
  const bool ew = threadIdx.x < 64 && b < B;
  int t = -1;
  float hp_v = 0.f, cp_v = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
  size_t r = 0;
  if (ew) {
    t = time_of(dir, a.step, a.lengths[b]);
    hp_v = hprev[(size_t)b * H + j];
    cp_v = cprev[(size_t)b * H + j];
    if (t >= 0) {
      r = (size_t)a.row_off[t] + b;
      const float* gi = a.gin + r * (size_t)(a.ndir * G4) + (size_t)dir * G4 + j;
      g0 = gi[0]; g1 = gi[H]; g2 = gi[2 * H]; g3 = gi[3 * H];
    }
  }

  const int row = b0 + lr;
  const bool rok = row < B;
  // K = H is split over the first a.ksplit waves, a.kiter steps of 16 k each
  const int kiter = wv < a.ksplit ? a.kiter : 0;
  const int kbase = wv * (16 * a.kiter) + 4 * kg;
  const float* hp = hprev + (size_t)(rok ? row : 0) * H + (kiter ? kbase : 0);
  const float* wp = whh + (size_t)((lr >> 2) * H + j0 + (lr & 3)) * H + (kiter ? kbase : 0);
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
    float hn = hp_v, cn = cp_v;
    if (t >= 0) {
      auto gate = [&](int n) { return (P[0][bl][n] + P[1][bl][n]) + (P[2][bl][n] + P[3][bl][n]); };
      const float ig = sigmoidf_acc(gate(u) + g0), fg = sigmoidf_acc(gate(4 + u) + g1),
                  gg = tanhf(gate(8 + u) + g2), og = sigmoidf_acc(gate(12 + u) + g3);
      cn = fg * cp_v + ig * gg;
      hn = og * tanhf(cn);
      a.y[r * (size_t)(a.ndir * H) + (size_t)dir * H + j] = hn;
      if (a.gates) {
        float* gs = a.gates + r * (size_t)(a.ndir * G4) + (size_t)dir * G4 + j;
        gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
        a.csave[r * (size_t)(a.ndir * H) + (size_t)dir * H + j] = cn;
        a.hprev[r * (size_t)(a.ndir * H) + (size_t)dir * H + j] = hp_v;
      }
    }
    hnext[(size_t)b * H + j] = hn;
    cnext[(size_t)b * H + j] = cn;
  }
}

// state init: hs/cs[parity 0][dir][b][:] = h0/c0[dir][:]
__global__ void lstm_init_state_kernel(const float* __restrict__ h0, const float* __restrict__ c0,
                                       float* __restrict__ hs, float* __restrict__ cs, int ndir, int B, int H) {
  const int64_t n = (int64_t)ndir * B * H;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % H);
    const int d = (int)(i / ((int64_t)B * H));
    hs[i] = h0 ? h0[d * H + j] : 0.f;
    cs[i] = c0 ? c0[d * H + j] : 0.f;
  }
}

// ---- backward step ----------------------------------------------------------------------------------
// Processes recurrence step s = a.step (called with s = T-1 ... 0). For row b active at s:
//   dh = dy[t] + dG[t_{s+1}] W_hh   (second term only if the row is active at s+1)
//   standard LSTM cell gradients -> dG[t], running dc (cs buffers, parity by step)
// Workgroup = 16 hidden units x 16 batch rows; the 4 waves split the K = 4H gate rows, operands
// stream from L2 into registers in chunks of 8 k-steps.  Grid (H/16 * ceil(nact/16), ndir).
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(LstmArgs a) {
  __shared__ float P[4][16][17];
  const int H = a.H, B = a.B, G4 = 4 * H;
  const int dir = blockIdx.y;
  const int ngroups = H / BW_UNITS;
  const int j0 = (blockIdx.x % ngroups) * BW_UNITS;
  const int b0 = (blockIdx.x / ngroups) * 16;
  const int par = a.step & 1;
  const float* wt = a.whh_t + (size_t)dir * H * G4;
  const float* dc_in = a.cs + ((size_t)(par ^ 1) * a.ndir + dir) * B * H;
  float* dc_out = a.cs + ((size_t)par * a.ndir + dir) * B * H;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  const size_t ldg = (size_t)a.ndir * G4, ldh = (size_t)a.ndir * H;

  // elementwise operands (thread -> batch row bl, unit n): load early
  const int bl = threadIdx.x >> 4, n = threadIdx.x & 15;
  const int b = b0 + bl, j = j0 + n;
  int t = -1;
  float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, ct = 0.f, cp = 0.f, dyv = 0.f, dcin = 0.f;
  size_t r = 0;
  if (b < B) {
    const int len = a.lengths[b];
    t = time_of(dir, a.step, len);
    if (t >= 0) {
      r = (size_t)a.row_off[t] + b;
      const float* gs = a.gates + r * ldg + (size_t)dir * G4 + j;
      ig = gs[0]; fg = gs[H]; gg = gs[2 * H]; og = gs[3 * H];
      ct = a.csave[r * ldh + (size_t)dir * H + j];
      const int tp = time_of(dir, a.step - 1, len);
      cp = (a.step > 0 && tp >= 0) ? a.csave[((size_t)a.row_off[tp] + b) * ldh + (size_t)dir * H + j]
                                   : (a.c0 ? a.c0[dir * H + j] : 0.f);
      dyv = a.dy[r * ldh + (size_t)dir * H + j];
      dcin = dc_in[(size_t)b * H + j];
    }
  }

  // dh_rec tile: A = dG of the next recurrence step (rows), B = W_hh^T rows of our 16 units
  const int row = b0 + lr;
  bool has_next = false;
  const float* dgp = a.dg;
  if (row < B) {
    const int tn = time_of(dir, a.step + 1, a.lengths[row]);
    if (tn >= 0) {
      has_next = true;
      dgp = a.dg + ((size_t)a.row_off[tn] + row) * ldg + (size_t)dir * G4;
    }
  }
  const int kiter = H / 16;            // 4H gate rows / 4 waves / 16 per step
  const int kbase = wv * H + 4 * kg;  // this wave's quarter of the 4H gate rows
  dgp += kbase;
  const float* wp = wt + (size_t)(j0 + lr) * G4 + kbase;
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
    float dc_keep = 0.f;
    if (t >= 0) {
      const float dhr = (P[0][bl][n] + P[1][bl][n]) + (P[2][bl][n] + P[3][bl][n]);
      const float tc = tanhf(ct);
      const float dh = dyv + dhr;
      const float dcv = dh * og * (1.f - tc * tc) + dcin;
      float* dgo = a.dg + r * ldg + (size_t)dir * G4 + j;
      dgo[0] = dcv * gg * ig * (1.f - ig);
      dgo[H] = dcv * cp * fg * (1.f - fg);
      dgo[2 * H] = dcv * ig * (1.f - gg * gg);
      dgo[3 * H] = dh * tc * og * (1.f - og);
      dc_keep = dcv * fg;
    }
    dc_out[(size_t)b * H + j] = dc_keep;
  }
}

// hn / cn of row b sit in the parity written by its last active step: (len_b & 1)
__global__ void rnn_final_state_kernel(const float* __restrict__ st, const int* __restrict__ lengths,
                                       float* __restrict__ out, int ndir, int B, int H) {
  const int64_t n = (int64_t)ndir * B * H;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)((i / H) % B);
    out[i] = st[(int64_t)(lengths[b] & 1) * n + i];
  }
}

}  // namespace itts

using namespace itts;

static int lstm_check(const int* h_lengths, int T, int B, int H, int ndir) {
  ITTS_REQUIRE(T >= 1 && B >= 1 && (ndir == 1 || ndir == 2), "bad sizes");
  ITTS_REQUIRE(H >= 16 && H % 16 == 0 && H <= 4096, "hidden size must be a multiple of 16");
  ITTS_REQUIRE(h_lengths != nullptr, "host copy of the lengths is required");
  ITTS_REQUIRE(h_lengths[0] == T && h_lengths[B - 1] >= 1, "T must be the longest length, all lengths >= 1");
  for (int b = 1; b < B; ++b) ITTS_REQUIRE(h_lengths[b] <= h_lengths[b - 1], "rows must be sorted by decreasing length");
  return ITTS_OK;
}

// number of rows still active at recurrence step s (lengths sorted decreasingly); `p` carries the
// previous answer so that a whole sweep costs O(B + T)
static inline int active_rows(const int* h_lengths, int B, int s, int* p) {
  while (*p > 0 && h_lengths[*p - 1] <= s) --*p;
  while (*p < B && h_lengths[*p] > s) ++*p;
  return *p;
}

extern "C" int64_t itts_lstm_state_bytes(int B, int H, int ndir) {
  if (B <= 0 || H <= 0 || ndir <= 0) return 0;
  return (int64_t)2 * 2 * ndir * B * H * 4;  // hs + cs, two parities each
}

// Runs the recurrence of one (bi)directional LSTM layer over T steps (packed rows, see the top).
extern "C" int itts_lstm_layer_fwd(const float* d_gin, const float* d_whh, const float* d_h0,
                                   const float* d_c0, const int* d_lengths, const int* h_lengths,
                                   const int* d_row_off, int T, int B, int H, int ndir, float* d_y,
                                   float* d_gates, float* d_csave, float* d_hprev, float* d_hn,
                                   float* d_cn, void* d_state, void* stream) {
  ITTS_REQUIRE(d_gin && d_whh && d_lengths && d_row_off && d_y && d_state, "null pointer");
  ITTS_REQUIRE((d_gates == nullptr) == (d_csave == nullptr) && (d_gates == nullptr) == (d_hprev == nullptr),
               "gates / csave / hprev must be given together (training) or all NULL (inference)");
  int rc = lstm_check(h_lengths, T, B, H, ndir);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  LstmArgs a{};
  a.T = T; a.B = B; a.H = H; a.ndir = ndir; a.lengths = d_lengths; a.row_off = d_row_off; a.gin = d_gin;
  a.whh = d_whh; a.h0 = d_h0; a.c0 = d_c0; a.y = d_y; a.gates = d_gates; a.csave = d_csave; a.hprev = d_hprev;
  const size_t st = (size_t)2 * ndir * B * H;
  a.hs = reinterpret_cast<float*>(d_state);
  a.cs = a.hs + st;
  const int64_t n = (int64_t)ndir * B * H;
  const dim3 eg((unsigned)std::min<int64_t>((n + 255) / 256, 1024));
  hipLaunchKernelGGL(lstm_init_state_kernel, eg, dim3(256), 0, s, d_h0, d_c0, a.hs, a.cs, ndir, B, H);
  ITTS_LAUNCH_CHECK();
  a.ksplit = (H % 64 == 0) ? 4 : ((H % 32 == 0) ? 2 : 1);
  a.kiter = H / (16 * a.ksplit);
  int p = B;
  for (int step = 0; step < T; ++step) {
    a.step = step;
    const int nact = active_rows(h_lengths, B, step, &p);
    hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3((H / FW_UNITS) * ((nact + 15) / 16), ndir), dim3(256), 0, s, a);
  }
  ITTS_LAUNCH_CHECK();
  if (d_hn) hipLaunchKernelGGL(rnn_final_state_kernel, eg, dim3(256), 0, s, a.hs, d_lengths, d_hn, ndir, B, H);
  if (d_cn) hipLaunchKernelGGL(rnn_final_state_kernel, eg, dim3(256), 0, s, a.cs, d_lengths, d_cn, ndir, B, H);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

// Backward recurrence: fills d_dg [N, ndir*4H] from d_dy and the saved forward tensors.
// d_whh_t is W_hh transposed per direction ([ndir][H][4H]).
extern "C" int itts_lstm_layer_bwd(const float* d_dy, const float* d_whh_t, const float* d_c0,
                                   const float* d_gates, const float* d_csave, const int* d_lengths,
                                   const int* h_lengths, const int* d_row_off, int T, int B, int H,
                                   int ndir, float* d_dg, void* d_state, void* stream) {
  ITTS_REQUIRE(d_dy && d_whh_t && d_gates && d_csave && d_lengths && d_row_off && d_dg && d_state, "null pointer");
  int rc = lstm_check(h_lengths, T, B, H, ndir);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  LstmArgs a{};
  a.T = T; a.B = B; a.H = H; a.ndir = ndir; a.lengths = d_lengths; a.row_off = d_row_off; a.whh_t = d_whh_t;
  a.c0 = d_c0; a.gates = const_cast<float*>(d_gates); a.csave = const_cast<float*>(d_csave); a.dy = d_dy;
  a.dg = d_dg;
  const size_t st = (size_t)2 * ndir * B * H;
  a.hs = reinterpret_cast<float*>(d_state);
  a.cs = a.hs + st;
  ITTS_HIP_CHECK(hipMemsetAsync(a.cs, 0, st * 4, s));   // running dc of rows that are not active yet
  int p = 0;
  for (int step = T - 1; step >= 0; --step) {
    a.step = step;
    const int nact = active_rows(h_lengths, B, step, &p);
    hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3((H / BW_UNITS) * ((nact + 15) / 16), ndir), dim3(256), 0, s, a);
  }
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}
