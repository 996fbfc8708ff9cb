// Padded mini-batches <-> rows stored back to back, on the device.
//
// The reference builds every mini-batch on the host: ModularModelHandlerPyTorch.prepare_batch (:388-465)
// pads the utterances of a batch with torch.nn.utils.rnn.pad_sequence, writes a float sequence mask
// (sequence_mask :467-491) and the training loop copies the result to the GPU (:745-760).  Here the
// normalised, length matched rows of the utterances stay in HBM between the epochs (data_preparation/
// DeviceBatchCache.py) and a batch is one launch per stream:
//
//   itts_batch_pad_gather_f32   padded[b, t, :] = rows[starts[b] + t, :]  (t < lens[b]),  else the fill row / zeros
//                               (+ the [B, T, 1] mask); also pad_packed_sequence for the frame-independent
//                               layers, which run on the valid rows only (nn/functional.py ValidRows)
//   itts_batch_pack_rows_f32    its adjoint: rows[starts[b] + t, :] = padded[b, t, :]
//   itts_batch_concat_rows_f32  rows[dst_starts[b] + t, :] = arena[src_starts[b] + t, :]: a mini-batch's valid frames
//                               back to back (FrameShard.gather, the flat feed-forward step's input)
//   itts_batch_pad_colsum_f32   the sum over the padding positions of a padded batch, per column (the gradient of
//                               the fill row), in a fixed order: partial sums per slab of rows, then one pass over
//                               the slabs
//
// HBM bound, one wave per row: a row is 425 or 187 floats (LJSpeech shape) whose pitch is not a multiple of
// 16 bytes, so lanes move one dword each -- 256 contiguous bytes per wave instruction, the shape that the
// guide's table of plain stores gives at 6.0-6.2 TB/s; rows of 16-byte multiples on both sides take float4.
#include <algorithm>

#include "common.h"

namespace itts {

static inline int aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// padded position r -> (b, t); the entry points keep n_utts * t_max below 2^31, so 32-bit division does
__device__ __forceinline__ void split_row(int64_t r, int64_t t_max, int n_utts, int batch_first, int& b, int64_t& t) {
  const uint32_t r32 = (uint32_t)r;
  if (batch_first) {
    const uint32_t q = r32 / (uint32_t)t_max;
    b = (int)q;
    t = r32 - q * (uint32_t)t_max;
  } else {
    const uint32_t q = r32 / (uint32_t)n_utts;
    t = q;
    b = (int)(r32 - q * (uint32_t)n_utts);
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void batch_pad_gather_kernel(const float* __restrict__ src, int64_t ld_src,
                                                               int64_t n_src, const int64_t* __restrict__ starts,
                                                               const int64_t* __restrict__ lens, int n_utts,
                                                               int64_t t_max, int width, int batch_first,
                                                               const float* __restrict__ fill, int64_t rep_pos,
                                                               const float* __restrict__ rep_row,
                                                               float* __restrict__ dst, int64_t ld_dst,
                                                               float* __restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int64_t n_rows = (int64_t)n_utts * t_max;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), wn = (int64_t)gridDim.x * 4;
  for (int64_t r = w0; r < n_rows; r += wn) {
    int b;
    int64_t t;
    split_row(r, t_max, n_utts, batch_first, b, t);
    const int64_t i = starts[b] + t;
    const bool valid = t < lens[b] && i >= 0 && i < n_src;
    const float* s = valid ? src + i * ld_src : (r == rep_pos ? rep_row : fill);
    float* d = dst + r * ld_dst;
    if (VEC) {
      for (int c = lane; c < (width >> 2); c += 64) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s) v = reinterpret_cast<const float4*>(s)[c];
        reinterpret_cast<float4*>(d)[c] = v;
      }
    } else {
      for (int c = lane; c < width; c += 64) d[c] = s ? s[c] : 0.f;
    }
    if (mask && lane == 0) mask[r] = valid ? 1.f : 0.f;
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void batch_pack_rows_kernel(const float* __restrict__ src, int64_t ld_src,
                                                              const int64_t* __restrict__ starts,
                                                              const int64_t* __restrict__ lens, int n_utts,
                                                              int64_t t_max, int width, int batch_first,
                                                              float* __restrict__ dst, int64_t ld_dst, int dst_width,
                                                              int64_t rep_pos, int64_t rep_dst_row) {
  const int lane = threadIdx.x & 63;
  const int64_t n_rows = (int64_t)n_utts * t_max;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), wn = (int64_t)gridDim.x * 4;
  for (int64_t r = w0; r < n_rows; r += wn) {
    int b;
    int64_t t;
    split_row(r, t_max, n_utts, batch_first, b, t);
    if (t >= lens[b] && r != rep_pos) continue;
    const float* s = src + r * ld_src;
    float* d = dst + (t < lens[b] ? starts[b] + t : rep_dst_row) * ld_dst;
    if (VEC) {
      for (int c = lane; c < (dst_width >> 2); c += 64) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < (width >> 2)) v = reinterpret_cast<const float4*>(s)[c];
        reinterpret_cast<float4*>(d)[c] = v;
      }
    } else {
      for (int c = lane; c < dst_width; c += 64) d[c] = c < width ? s[c] : 0.f;
    }
  }
}

// dst[dst_starts[b] + t, :] = src[src_starts[b] + t, :] for t < lens[b]: the utterances of a mini-batch out of an
// arena, back to back in batch order (FrameShard.gather: the packed valid frames the flat feed-forward step takes).
// Walks the B x t_max grid of positions like the kernels above; positions beyond an utterance's length leave at once.
template <bool VEC>
__global__ __launch_bounds__(256) void batch_concat_rows_kernel(const float* __restrict__ src, int64_t ld_src,
                                                                int64_t n_src, const int64_t* __restrict__ src_starts,
                                                                const int64_t* __restrict__ dst_starts,
                                                                const int64_t* __restrict__ lens, int n_utts,
                                                                int64_t t_max, int width, float* __restrict__ dst,
                                                                int64_t ld_dst, int dst_width) {
  const int lane = threadIdx.x & 63;
  const int64_t n_rows = (int64_t)n_utts * t_max;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), wn = (int64_t)gridDim.x * 4;
  for (int64_t r = w0; r < n_rows; r += wn) {
    int b;
    int64_t t;
    split_row(r, t_max, n_utts, 1, b, t);
    if (t >= lens[b]) continue;
    const int64_t i = src_starts[b] + t;
    const float* s = (i >= 0 && i < n_src) ? src + i * ld_src : nullptr;
    float* d = dst + (dst_starts[b] + t) * ld_dst;
    if (VEC) {
      for (int c = lane; c < (dst_width >> 2); c += 64) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s && c < (width >> 2)) v = reinterpret_cast<const float4*>(s)[c];
        reinterpret_cast<float4*>(d)[c] = v;
      }
    } else {
      for (int c = lane; c < dst_width; c += 64) d[c] = (s && c < width) ? s[c] : 0.f;
    }
  }
}

constexpr int kColsumSlabRows = 64;      // rows per workgroup of stage 1: 16 per wave

// stage 1: workgroup s sums the padding rows among its 64 consecutive rows -- wave w owns rows 16 w .. 16 w + 15
// of the slab, its first 16 lanes work out (one division each, side by side) which of them are padding, the wave
// then walks the set bits in row order, lane l adding the columns l, l + 64, ...; the four waves' sums meet in LDS
// in wave order.  A fixed order throughout: the same batch gives the same bits.
__global__ __launch_bounds__(256) void batch_pad_colsum_partial_kernel(const float* __restrict__ x, int64_t ld,
                                                                       const int64_t* __restrict__ lens, int n_utts,
                                                                       int64_t t_max, int width, int batch_first,
                                                                       float* __restrict__ partial) {
  extern __shared__ float sm[];      // [4][width]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_rows = (int64_t)n_utts * t_max;
  const int64_t r0 = (int64_t)blockIdx.x * kColsumSlabRows + wave * 16;
  bool pad = false;
  if (lane < 16 && r0 + lane < n_rows) {
    int b;
    int64_t t;
    split_row(r0 + lane, t_max, n_utts, batch_first, b, t);
    pad = t >= lens[b];
  }
  const unsigned long long pads = __ballot(pad);
  for (int c0 = 0; c0 < width; c0 += 64 * 8) {      // eight columns per lane and sweep
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (unsigned long long m = pads; m; m &= m - 1) {
      const float* s = x + (r0 + __builtin_ctzll(m)) * ld;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j * 64 + lane;
        if (c < width) acc[j] += s[c];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j * 64 + lane;
      if (c < width) sm[wave * width + c] = acc[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < width; c += 256)
    partial[(int64_t)blockIdx.x * width + c] = ((sm[c] + sm[width + c]) + sm[2 * width + c]) + sm[3 * width + c];
}

// stage 2: one wave per column -- lane l adds the slabs l, l + 64, ... in that order, the 64 sums meet in the
// fixed butterfly of wave_sum
__global__ __launch_bounds__(256) void batch_pad_colsum_final_kernel(const float* __restrict__ partial, int n_slabs,
                                                                     int width, float* __restrict__ out,
                                                                     int out_width) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= out_width) return;
  float acc = 0.f;
  if (c < width)
    for (int s = lane; s < n_slabs; s += 64) acc += partial[(int64_t)s * width + c];
  acc = wave_sum(acc);
  if (lane == 0) out[c] = acc;
}

static unsigned row_grid(int64_t n_rows) { return (unsigned)std::min<int64_t>((n_rows + 3) / 4, 16384); }

}  // namespace itts

using namespace itts;

extern "C" int itts_batch_pad_gather_f32(const float* d_src, int64_t ld_src, int64_t n_src, const int64_t* d_starts,
                                         const int64_t* d_lens, int n_utts, int64_t t_max, int width, int batch_first,
                                         const float* d_fill_row, int64_t rep_pos, const float* d_rep_row,
                                         float* d_dst, int64_t ld_dst, float* d_mask, void* stream) {
  ITTS_REQUIRE(n_utts >= 0 && t_max >= 0 && width >= 0 && n_src >= 0 && ld_src >= width && ld_dst >= width, "bad sizes");
  const int64_t n_rows = (int64_t)n_utts * t_max;
  if (n_rows == 0) return ITTS_OK;
  ITTS_REQUIRE(n_rows < ((int64_t)1 << 31), "more than 2^31 positions");
  ITTS_REQUIRE(d_starts && d_lens && d_dst && (d_src || n_src == 0), "null pointer");
  ITTS_REQUIRE(rep_pos < n_rows && (rep_pos < 0 || d_rep_row), "bad representative position");
  const bool vec = width % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && aligned16(d_src) && aligned16(d_dst) &&
                   (!d_fill_row || aligned16(d_fill_row)) && (rep_pos < 0 || aligned16(d_rep_row));
  if (vec)
    hipLaunchKernelGGL(batch_pad_gather_kernel<true>, dim3(row_grid(n_rows)), dim3(256), 0, as_stream(stream), d_src,
                       ld_src, n_src, d_starts, d_lens, n_utts, t_max, width, batch_first, d_fill_row, rep_pos, d_rep_row,
                       d_dst, ld_dst, d_mask);
  else
    hipLaunchKernelGGL(batch_pad_gather_kernel<false>, dim3(row_grid(n_rows)), dim3(256), 0, as_stream(stream), d_src,
                       ld_src, n_src, d_starts, d_lens, n_utts, t_max, width, batch_first, d_fill_row, rep_pos, d_rep_row,
                       d_dst, ld_dst, d_mask);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_batch_pack_rows_f32(const float* d_src, int64_t ld_src, const int64_t* d_starts,
                                        const int64_t* d_lens, int n_utts, int64_t t_max, int width, int batch_first,
                                        float* d_dst, int64_t ld_dst, int dst_width, int64_t rep_pos,
                                        int64_t rep_dst_row, void* stream) {
  ITTS_REQUIRE(n_utts >= 0 && t_max >= 0 && width >= 0 && dst_width >= width && ld_src >= width && ld_dst >= dst_width,
               "bad sizes");
  const int64_t n_rows = (int64_t)n_utts * t_max;
  if (n_rows == 0 || dst_width == 0) return ITTS_OK;
  ITTS_REQUIRE(n_rows < ((int64_t)1 << 31), "more than 2^31 positions");
  ITTS_REQUIRE(d_starts && d_lens && d_src && d_dst, "null pointer");
  ITTS_REQUIRE(rep_pos < n_rows && (rep_pos < 0 || rep_dst_row >= 0), "bad representative position");
  const bool vec = width % 4 == 0 && dst_width % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && aligned16(d_src) &&
                   aligned16(d_dst);
  if (vec)
    hipLaunchKernelGGL(batch_pack_rows_kernel<true>, dim3(row_grid(n_rows)), dim3(256), 0, as_stream(stream), d_src,
                       ld_src, d_starts, d_lens, n_utts, t_max, width, batch_first, d_dst, ld_dst, dst_width, rep_pos,
                       rep_dst_row);
  else
    hipLaunchKernelGGL(batch_pack_rows_kernel<false>, dim3(row_grid(n_rows)), dim3(256), 0, as_stream(stream), d_src,
                       ld_src, d_starts, d_lens, n_utts, t_max, width, batch_first, d_dst, ld_dst, dst_width, rep_pos,
                       rep_dst_row);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_batch_concat_rows_f32(const float* d_src, int64_t ld_src, int64_t n_src, const int64_t* d_src_starts,
                                          const int64_t* d_dst_starts, const int64_t* d_lens, int n_utts, int64_t t_max,
                                          int width, float* d_dst, int64_t ld_dst, int dst_width, void* stream) {
  ITTS_REQUIRE(n_utts >= 0 && t_max >= 0 && width >= 0 && dst_width >= width && n_src >= 0 && ld_src >= width &&
                   ld_dst >= dst_width, "bad sizes");
  const int64_t n_rows = (int64_t)n_utts * t_max;
  if (n_rows == 0 || dst_width == 0) return ITTS_OK;
  ITTS_REQUIRE(n_rows < ((int64_t)1 << 31), "more than 2^31 positions");
  ITTS_REQUIRE(d_src_starts && d_dst_starts && d_lens && d_dst && (d_src || n_src == 0), "null pointer");
  const bool vec = width % 4 == 0 && dst_width % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && aligned16(d_src) &&
                   aligned16(d_dst);
  if (vec)
    hipLaunchKernelGGL(batch_concat_rows_kernel<true>, dim3(row_grid(n_rows)), dim3(256), 0, as_stream(stream), d_src,
                       ld_src, n_src, d_src_starts, d_dst_starts, d_lens, n_utts, t_max, width, d_dst, ld_dst, dst_width);
  else
    hipLaunchKernelGGL(batch_concat_rows_kernel<false>, dim3(row_grid(n_rows)), dim3(256), 0, as_stream(stream), d_src,
                       ld_src, n_src, d_src_starts, d_dst_starts, d_lens, n_utts, t_max, width, d_dst, ld_dst, dst_width);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int64_t itts_batch_pad_colsum_workspace_bytes(int64_t n_rows, int width) {
  const int64_t n_slabs = (std::max<int64_t>(n_rows, 0) + kColsumSlabRows - 1) / kColsumSlabRows;
  return std::max<int64_t>(n_slabs, 1) * std::max(width, 0) * (int64_t)sizeof(float);
}

extern "C" int itts_batch_pad_colsum_f32(const float* d_x, int64_t ld_x, const int64_t* d_lens, int n_utts,
                                         int64_t t_max, int width, int batch_first, float* d_out, int out_width,
                                         void* d_workspace, void* stream) {
  ITTS_REQUIRE(n_utts >= 0 && t_max >= 0 && width >= 0 && ld_x >= width && out_width >= width, "bad sizes");
  if (out_width == 0) return ITTS_OK;
  ITTS_REQUIRE(d_out && d_workspace, "null pointer");
  ITTS_REQUIRE(width <= 4096, "at most 4096 columns");      // 4 x width floats of LDS
  const int64_t n_rows = (int64_t)n_utts * t_max;
  if (n_rows == 0) {
    ITTS_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(float) * out_width, as_stream(stream)));
    return ITTS_OK;
  }
  ITTS_REQUIRE(d_x && d_lens, "null pointer");
  ITTS_REQUIRE(n_rows < ((int64_t)1 << 31), "more than 2^31 positions");
  const int n_slabs = (int)((n_rows + kColsumSlabRows - 1) / kColsumSlabRows);
  float* partial = static_cast<float*>(d_workspace);
  hipLaunchKernelGGL(batch_pad_colsum_partial_kernel, dim3(n_slabs), dim3(256), sizeof(float) * 4 * width,
                     as_stream(stream), d_x, ld_x, d_lens, n_utts, t_max, width, batch_first, partial);
  ITTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(batch_pad_colsum_final_kernel, dim3((out_width + 3) / 4), dim3(256), 0, as_stream(stream),
                     partial, n_slabs, width, d_out, out_width);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}
