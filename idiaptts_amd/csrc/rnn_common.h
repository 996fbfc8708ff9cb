// Shared pieces of the LSTM / GRU recurrence kernels (lstm.hip, gru.hip).
//
// Memory layouts that make every operand load of a step a fully coalesced 1 KB wave access.
// v_mfma_f32_16x16x4_f32 wants lane (lr = lane & 15, kg = lane >> 4) to hold row lr, k = kg; with
// the K-permutation trick a lane loads 4 consecutive k (16 bytes) per MFMA quartet.  In a row-major
// matrix the 16 rows of a tile are kilobytes apart, so a wave load touches 64 separate 16-byte
// pieces and the L1 serialises them (measured: ~90 clocks per load, 6 us per step for 40 loads).
// Hence:
//   * running state (h, c, the backward pass's carried dG) lives in K-BLOCKED form
//       blocked(b, k) = ((k >> 2) * B + b) * 4 + (k & 3)          [K/4][B][4]
//     so the 16 rows of a tile are 16 consecutive float4;
//   * W_hh is re-tiled once per layer call into [unit group][k block][16 tile rows][4], the exact
//     order the lanes read it.
// These are private scratch layouts inside d_state; the C ABI keeps torch's layouts.
#pragma once
#include <algorithm>

#include "common.h"

namespace itts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ size_t blocked(int b, int k, int B) {
  return ((size_t)(k >> 2) * B + b) * 4 + (k & 3);
}

// The cell's sigmoid and tanh sit on the recurrence's critical path (five of them per LSTM step and
// thread, one after the products): the hardware's exp2 and reciprocal (1 ulp each) instead of the
// library's expf / IEEE division / tanhf -- 4 and 6 instructions against ~30 and ~40.  Absolute error
// below 2e-7 everywhere (tanh is formed as 1 - 2 / (1 + e^{2x}): no relative accuracy near 0, none needed:
// it multiplies a gate); saturates to 0 / 1 / -1 for large arguments like the library forms.
__device__ __forceinline__ float sigmoid_acc(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
__device__ __forceinline__ float tanh_cell(float x) {
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * 2.88539008177792681f));
}

// Forward tiling of W_hh [ndir][G*H][H] (G = 3 or 4 gates) for workgroups of 4 hidden units:
//   out[dir][jg = H/4][kb = H/4][lr = gate*4 + unit][4] = W[dir][gate*H + jg*4 + unit][kb*4 ..]
// tile rows of a missing 4th gate are zero.
static __global__ void rnn_pack_w_fwd_kernel(const float* __restrict__ w, float* __restrict__ out, int ndir,
                                      int G, int H) {
  const int64_t n = (int64_t)ndir * (H / 4) * (H / 4) * 16;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int lr = (int)(i & 15);
    const int kb = (int)((i >> 4) % (H / 4));
    const int jg = (int)(((i >> 4) / (H / 4)) % (H / 4));
    const int d = (int)((i >> 4) / ((int64_t)(H / 4) * (H / 4)));
    const int g = lr >> 2, u = lr & 3;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < G) v = *reinterpret_cast<const float4*>(w + ((size_t)d * G * H + (size_t)g * H + jg * 4 + u) * H + kb * 4);
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// Backward tiling (W_hh^T rows of 16 hidden units against the K = G*H gate rows):
//   out[dir][jg = H/16][kb = G*H/4][lr][kk] = W[dir][kb*4 + kk][jg*16 + lr]
static __global__ void rnn_pack_w_bwd_kernel(const float* __restrict__ w, float* __restrict__ out, int ndir,
                                      int G, int H) {
  const int KB = G * H / 4;
  const int64_t n = (int64_t)ndir * (H / 16) * KB * 16;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int lr = (int)(i & 15);
    const int kb = (int)((i >> 4) % KB);
    const int jg = (int)(((i >> 4) / KB) % (H / 16));
    const int d = (int)((i >> 4) / ((int64_t)KB * (H / 16)));
    const float* src = w + ((size_t)d * G * H + (size_t)kb * 4) * H + jg * 16 + lr;
    reinterpret_cast<float4*>(out)[i] = make_float4(src[0], src[H], src[2 * (size_t)H], src[3 * (size_t)H]);
  }
}

// state[parity 0][dir] (blocked) = init[dir][:] broadcast over the batch (zeros when init is NULL)
static __global__ void rnn_init_state_kernel(const float* __restrict__ init, float* __restrict__ st, int ndir,
                                      int B, int H) {
  const int64_t n = (int64_t)ndir * B * H;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i / ((int64_t)B * H));
    const int64_t e = i % ((int64_t)B * H);          // blocked offset inside the direction
    const int j = (int)(e / (4 * (int64_t)B)) * 4 + (int)(e & 3);
    st[i] = init ? init[d * H + j] : 0.f;
  }
}

// out[dir][b][j] = state of row b in the parity written by its last active step, (len_b & 1)
static __global__ void rnn_final_state_kernel(const float* __restrict__ st, const int* __restrict__ lengths,
                                       float* __restrict__ out, int ndir, int B, int H) {
  const int64_t n = (int64_t)ndir * B * H;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % H);
    const int b = (int)((i / H) % B);
    const int d = (int)(i / ((int64_t)B * H));
    out[i] = st[((int64_t)(lengths[b] & 1) * ndir + d) * B * H + blocked(b, j, B)];
  }
}

}  // namespace itts

static inline int rnn_check(const int* h_lengths, int T, int B, int H, int ndir) {
  ITTS_REQUIRE(T >= 1 && B >= 1 && (ndir == 1 || ndir == 2), "bad sizes");
  ITTS_REQUIRE(H >= 16 && H % 16 == 0 && H <= 4096, "hidden size must be a multiple of 16");
  ITTS_REQUIRE(h_lengths != nullptr, "host copy of the lengths is required");
  ITTS_REQUIRE(h_lengths[0] == T && h_lengths[B - 1] >= 1, "T must be the longest length, all lengths >= 1");
  for (int b = 1; b < B; ++b) ITTS_REQUIRE(h_lengths[b] <= h_lengths[b - 1], "rows must be sorted by decreasing length");
  return ITTS_OK;
}

// number of rows still active at recurrence step s (lengths sorted decreasingly); `p` carries the
// previous answer so that a whole sweep costs O(B + T)
static inline int rnn_active_rows(const int* h_lengths, int B, int s, int* p) {
  while (*p > 0 && h_lengths[*p - 1] <= s) --*p;
  while (*p < B && h_lengths[*p] > s) ++*p;
  return *p;
}

static inline dim3 rnn_ew_grid(int64_t n) { return dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)); }
