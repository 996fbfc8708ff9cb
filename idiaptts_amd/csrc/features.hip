// Frame-feature tail of WORLD feature extraction on the device: everything the reference does on
// the host between pyworld / pysptk and the files it writes.
//
//   lf0 / V-UV        WorldFeatLabelGen.world_extract_features  (world/WorldFeatLabelGen.py:798-802)
//   interpolate_lin   misc/utils.py:40-86  (sequential gap filling, reproduced bit for bit)
//   deltas            misc/utils.py:103-105 (np.gradient in float32) + the stream layout of
//                     save_output (WorldFeatLabelGen.py:1121-1172): [sp, d, dd | lf0, d, dd | vuv | bap, d, dd]
//   statistics        MeanCovarianceExtractor / MeanStdDevExtractor.add_sample
//                     (misc/normalisation/*.py): sum x, sum x x^T (or sum x^2), here in fp64
//
// All of it is HBM-bound streaming over [frames x features] matrices; nothing is GEMM-shaped
// except sum x x^T, which runs on the fp64 matrix cores (products of two f32 values are exact in
// fp64, so the sums carry no rounding from the multiply).
#include <algorithm>
#include <cmath>

#include "common.h"
#include "context.h"

namespace itts {

// ------------------------------------------------------------------------------ interpolate_lin
// The reference walks the contour once: a frame <= 0 opens a gap [i, j) up to the next frame > 0
// (j = last index when there is none).  If j < n - 1: the gap is filled by linear interpolation
// between data[i-1] and data[j] -- with step (data[j] - data[i-1]) / (j - i), so the target value
// is reached one frame early -- or, without a voiced frame in front, with data[j].  Otherwise
// (the next voiced frame is the LAST frame, or there is none) everything from i to the end,
// including that last frame, becomes the last voiced value (0 if there was none).  All arithmetic
// is float32 without contraction.  Filled frames are > 0, so gaps never interact: frame k only
// needs p = previous voiced index and q = next voiced index -- two block scans per utterance.

constexpr int IL_THREADS = 256;

// inclusive max-scan over the block's threads, `lds` holds IL_THREADS / 64 ints
__device__ __forceinline__ int block_scan_max(int v, int* lds, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(v, off, 64);
    if (lane >= off) v = max(v, o);
  }
  __syncthreads();
  if (lane == 63) lds[w] = v;
  __syncthreads();
  int carry = -1, tot = -1;
  for (int i = 0; i < IL_THREADS / 64; ++i) {
    if (i < w) carry = max(carry, lds[i]);
    tot = max(tot, lds[i]);
  }
  *total = tot;
  return max(v, carry);
}

// FROM_F0: contour = float32 log of the clipped f0 with everything <= log(threshold) set to
// lf0_zero (WorldFeatLabelGen.py:798-800); otherwise the contour is read from `in`.
template <bool FROM_F0>
__global__ __launch_bounds__(IL_THREADS) void interpolate_lin_kernel(
    const double* __restrict__ f0, const float* __restrict__ in, const int64_t* __restrict__ off,
    float log_thr, float lf0_zero, float* __restrict__ ip, float* __restrict__ vuv,
    int* __restrict__ prev) {
  __shared__ int lds[IL_THREADS / 64];
  const int u = blockIdx.x;
  const int64_t t0 = off[u];
  const int T = (int)(off[u + 1] - t0);
  if (T <= 0) return;
  float* d = ip + t0;       // staging: the original contour, gaps are overwritten in place
  int* pv = prev + t0;
  // pass 1: contour, V/UV, previous voiced index (forward max-scan)
  int carry = -1;
  for (int base = 0; base < T; base += IL_THREADS) {
    const int t = base + (int)threadIdx.x;
    float v = 0.f;
    if (t < T) {
      if (FROM_F0) {
        const float x = (float)fmax(f0[t0 + t], 1e-10);
        v = (float)log((double)x);
        if (v <= log_thr) v = lf0_zero;
      } else {
        v = in[t0 + t];
      }
      d[t] = v;
      vuv[t0 + t] = v > 0.f ? 1.f : 0.f;
    }
    int tot;
    const int s = max(block_scan_max((t < T && v > 0.f) ? t : -1, lds, &tot), carry);
    if (t < T) pv[t] = s;
    carry = max(carry, tot);
  }
  __syncthreads();
  const bool last_voiced = d[T - 1] > 0.f;
  // pass 2: next voiced index (max-scan in reversed coordinates) and the fill
  carry = -1;
  for (int base = 0; base < T; base += IL_THREADS) {
    const int r = base + (int)threadIdx.x;
    const int k = T - 1 - r;
    const bool inside = r < T;
    const float v = inside ? d[k] : 0.f;
    int tot;
    const int s = max(block_scan_max((inside && v > 0.f) ? r : -1, lds, &tot), carry);
    carry = max(carry, tot);
    if (inside && !(v > 0.f)) {
      const int q = s < 0 ? -1 : T - 1 - s;
      const int p = pv[k];
      float out;
      if (q >= 0 && q < T - 1) {
        if (p >= 0) {
          const int i = p + 1;
          const float dp = d[p];
          const float step = __fdiv_rn(__fsub_rn(d[q], dp), (float)(q - i));
          out = __fadd_rn(dp, __fmul_rn(step, (float)(k - i + 1)));
        } else {
          out = d[q];
        }
      } else {
        out = p >= 0 ? d[p] : 0.f;
      }
      d[k] = out;      // only unvoiced frames are written; p and q are voiced frames
    }
  }
  // the tail fill also overwrites a voiced LAST frame when the frame before it is unvoiced
  __syncthreads();
  if (threadIdx.x == 0 && T >= 2 && last_voiced) {
    // frame T-2 unvoiced in the original contour <=> its previous-voiced index is not itself
    const int p = pv[T - 2];
    if (p != T - 2) d[T - 1] = p >= 0 ? d[p] : 0.f;
  }
}

// ---------------------------------------------------------------------------- deltas + layout
// np.gradient(f, axis=0) for float32: (f[t+1] - f[t-1]) / 2 inside, one-sided differences at the
// ends; a one-frame utterance gives 0 (numpy raises there, the reference never hits it).
struct CmpArgs {
  const float* sp;
  int64_t ld_sp;
  int n_sp;
  const float* lf0;
  const float* vuv;
  const float* bap;
  int64_t ld_bap;
  int n_bap;
  const int64_t* off;
  int n_utts;
  float* out;
  int64_t ld_out;
  int64_t t_total;
};

__device__ __forceinline__ float grad_at(const float* c, int64_t ld, int t, int T) {
  if (T == 1) return 0.f;
  if (t == 0) return __fsub_rn(c[ld], c[0]);
  if (t == T - 1) return __fsub_rn(c[(int64_t)t * ld], c[(int64_t)(t - 1) * ld]);
  return __fmul_rn(__fsub_rn(c[(int64_t)(t + 1) * ld], c[(int64_t)(t - 1) * ld]), 0.5f);
}

template <bool DELTAS>
__global__ __launch_bounds__(256) void assemble_cmp_kernel(CmpArgs a) {
  // one thread per (frame, static column); columns: n_sp coded-sp, lf0, vuv, n_bap bap
  const int ncol = a.n_sp + 2 + a.n_bap;
  const int64_t n = a.t_total * ncol;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / ncol;
    const int c = (int)(idx - row * ncol);
    // utterance of this frame: binary search in the offsets
    int lo = 0, hi = a.n_utts;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.off[mid] <= row) lo = mid; else hi = mid;
    }
    const int64_t t0 = a.off[lo];
    const int T = (int)(a.off[lo + 1] - t0);
    const int t = (int)(row - t0);
    const float* col;
    int64_t ld;
    int width, ocol;          // stream width, first output column of the stream, index in stream
    int j;
    if (c < a.n_sp) {
      col = a.sp + t0 * a.ld_sp + c; ld = a.ld_sp; width = a.n_sp; ocol = 0; j = c;
    } else if (c == a.n_sp) {
      col = a.lf0 + t0; ld = 1; width = 1; ocol = (DELTAS ? 3 : 1) * a.n_sp; j = 0;
    } else if (c == a.n_sp + 1) {
      a.out[row * a.ld_out + (DELTAS ? 3 : 1) * (a.n_sp + 1)] = a.vuv[row];
      continue;
    } else {
      j = c - a.n_sp - 2;
      col = a.bap + t0 * a.ld_bap + j; ld = a.ld_bap; width = a.n_bap;
      ocol = (DELTAS ? 3 : 1) * (a.n_sp + 1) + 1;
    }
    float* o = a.out + row * a.ld_out + ocol + j;
    o[0] = col[(int64_t)t * ld];
    if (DELTAS) {
      const float g = grad_at(col, ld, t, T);
      float gg;
      if (T == 1) {
        gg = 0.f;
      } else if (t == 0) {
        gg = __fsub_rn(grad_at(col, ld, 1, T), g);
      } else if (t == T - 1) {
        gg = __fsub_rn(g, grad_at(col, ld, t - 1, T));
      } else {
        gg = __fmul_rn(__fsub_rn(grad_at(col, ld, t + 1, T), grad_at(col, ld, t - 1, T)), 0.5f);
      }
      o[width] = g;
      o[2 * width] = gg;
    }
  }
}

// ------------------------------------------------------------------------------- statistics
// sum x (and sum x^2) per column: partial sums of row chunks, then a fixed-order reduction
constexpr int ST_CHUNKS = 256;

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t ld,
                                                             int64_t n_rows, int col0, int width,
                                                             double* __restrict__ part_sum,
                                                             double* __restrict__ part_sq) {
  // block = row chunk; thread = column (looping when width > 256)
  const int64_t per = (n_rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = blockIdx.x * per, r1 = min(n_rows, r0 + per);
  for (int c = threadIdx.x; c < width; c += blockDim.x) {
    double s = 0.0, q = 0.0;
    for (int64_t r = r0; r < r1; ++r) {
      const double v = (double)x[r * ld + col0 + c];
      s += v;
      q += v * v;
    }
    part_sum[(int64_t)blockIdx.x * width + c] = s;
    if (part_sq) part_sq[(int64_t)blockIdx.x * width + c] = q;
  }
}

__global__ void reduce_parts_kernel(const double* __restrict__ part, int n_parts, int64_t n,
                                    double* __restrict__ out, int accumulate) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int p = 0; p < n_parts; ++p) s += part[(int64_t)p * n + i];
  out[i] = accumulate ? out[i] + s : s;
}

// sum x x^T on the fp64 matrix cores: one wave per (16x16 output tile, row chunk).
// v_mfma_f64_16x16x4_f64: A [16 x 4] lane l holds A[l % 16][l / 16]; B [4 x 16] lane l holds
// B[l / 16][l % 16]; D lane l holds D[(l / 16) + 4 * i][l % 16], i = 0..3.
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int XTX_CHUNKS = 64;

__global__ __launch_bounds__(64) void xtx_partial_kernel(const float* __restrict__ x, int64_t ld,
                                                         int64_t n_rows, int col0, int width,
                                                         double* __restrict__ part) {
  const int tiles = (width + 15) / 16;
  const int ti = blockIdx.x / tiles, tj = blockIdx.x % tiles;
  if (tj < ti) return;                     // upper triangle only; mirrored by the reduction
  const int chunk = blockIdx.y;
  const int64_t per = ((n_rows + XTX_CHUNKS - 1) / XTX_CHUNKS + 3) / 4 * 4;
  const int64_t r0 = chunk * per, r1 = min(n_rows, r0 + per);
  const int lane = threadIdx.x;
  const int ci = ti * 16 + (lane & 15), cj = tj * 16 + (lane & 15);
  const bool oki = ci < width, okj = cj < width;
  const float* xi = x + col0 + ci;
  const float* xj = x + col0 + cj;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int64_t rb = r0; rb < r1; rb += 4) {     // wave-uniform trip count
    const int64_t r = rb + (lane >> 4);
    const bool in = r < r1;
    const double a = (in && oki) ? (double)xi[r * ld] : 0.0;
    const double b = (in && okj) ? (double)xj[r * ld] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  double* p = part + (int64_t)chunk * width * width;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = ti * 16 + (lane >> 4) + 4 * i;
    if (row < width && okj) p[(int64_t)row * width + cj] = acc[i];
  }
}

__global__ void xtx_reduce_kernel(const double* __restrict__ part, int width,
                                  double* __restrict__ out, int accumulate) {
  const int64_t n = (int64_t)width * width;
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int r = (int)(i / width), c = (int)(i % width);
  // partial tiles exist for 16x16 tiles with tj >= ti; inside a diagonal tile both triangles are
  // computed, below-diagonal tiles are mirrored
  const int64_t src = (r / 16 <= c / 16) ? (int64_t)r * width + c : (int64_t)c * width + r;
  double s = 0.0;
  for (int p = 0; p < XTX_CHUNKS; ++p) s += part[(int64_t)p * n + src];
  out[i] = accumulate ? out[i] + s : s;
}

}  // namespace itts

using namespace itts;

static int interpolate_launch(const double* d_f0, const float* d_in, const int64_t* h_off, int n_utts,
                              double thr, float lf0_zero, float* d_ip, float* d_vuv, hipStream_t s) {
  if (n_utts == 0 || h_off[n_utts] == 0) return ITTS_OK;
  const int64_t total = h_off[n_utts];
  int64_t* d_off = nullptr;
  int rc = upload_i64(h_off, n_utts + 1, &d_off, s);
  if (rc != ITTS_OK) return rc;
  int* d_prev = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_prev, total * sizeof(int), s));
  if (d_f0)
    hipLaunchKernelGGL(interpolate_lin_kernel<true>, dim3(n_utts), dim3(IL_THREADS), 0, s, d_f0,
                       nullptr, d_off, (float)std::log(thr), lf0_zero, d_ip, d_vuv, d_prev);
  else
    hipLaunchKernelGGL(interpolate_lin_kernel<false>, dim3(n_utts), dim3(IL_THREADS), 0, s, nullptr,
                       d_in, d_off, 0.f, 0.f, d_ip, d_vuv, d_prev);
  ITTS_LAUNCH_CHECK();
  ITTS_HIP_CHECK(itts::scratch_free(d_prev, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_off, s));
  return ITTS_OK;
}

extern "C" int itts_lf0_vuv(const double* d_f0, const int64_t* h_f_off, int n_utts,
                            double f0_silence_threshold, float lf0_zero, float* d_lf0,
                            float* d_vuv, void* stream) {
  ITTS_REQUIRE(h_f_off && n_utts >= 0, "null offsets");
  ITTS_REQUIRE(n_utts == 0 || h_f_off[n_utts] == 0 || (d_f0 && d_lf0 && d_vuv), "null pointer");
  ITTS_REQUIRE(f0_silence_threshold > 0.0, "f0_silence_threshold must be positive");
  for (int u = 0; u < n_utts; ++u)
    ITTS_REQUIRE(h_f_off[u + 1] >= h_f_off[u] && h_f_off[u + 1] - h_f_off[u] < (1ll << 24),
                 "utterance lengths must be in [0, 2^24)");
  return interpolate_launch(d_f0, nullptr, h_f_off, n_utts, f0_silence_threshold, lf0_zero, d_lf0,
                            d_vuv, as_stream(stream));
}

// amp_sp = sqrt(sp) in place (WorldFeatLabelGen.py:795, `np.sqrt` of pyworld's power envelope): IEEE square
// roots, so the same bits as numpy's; 0.5 ms of one host core per 6.6-s utterance otherwise
__global__ __launch_bounds__(256) void sqrt_inplace_kernel(double* __restrict__ x, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] = __dsqrt_rn(x[i]);
}

// sp = amp_sp^2 in place (WorldFeatLabelGen.py:925, `np.square(amp_sp, dtype=np.float64)` in front of the synthesis):
// one IEEE product per value, numpy's bits; 0.3 ms of one host core per 6.6-s utterance otherwise
__global__ __launch_bounds__(256) void square_inplace_kernel(double* __restrict__ x, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const double v = x[i];
    x[i] = __dmul_rn(v, v);
  }
}

extern "C" int itts_square_inplace_f64(double* d_x, int64_t n, void* stream) {
  ITTS_REQUIRE(n >= 0 && (n == 0 || d_x), "bad arguments");
  if (n == 0) return ITTS_OK;
  const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(square_inplace_kernel, dim3(grid), dim3(256), 0, as_stream(stream), d_x, n);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_sqrt_inplace_f64(double* d_x, int64_t n, void* stream) {
  ITTS_REQUIRE(n >= 0 && (n == 0 || d_x), "bad arguments");
  if (n == 0) return ITTS_OK;
  const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(sqrt_inplace_kernel, dim3(grid), dim3(256), 0, as_stream(stream), d_x, n);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_interpolate_lin_f32(const float* d_in, const int64_t* h_off, int n_utts,
                                        float* d_ip, float* d_vuv, void* stream) {
  ITTS_REQUIRE(h_off && n_utts >= 0, "null offsets");
  ITTS_REQUIRE(n_utts == 0 || h_off[n_utts] == 0 || (d_in && d_ip && d_vuv), "null pointer");
  for (int u = 0; u < n_utts; ++u)
    ITTS_REQUIRE(h_off[u + 1] >= h_off[u] && h_off[u + 1] - h_off[u] < (1ll << 24),
                 "utterance lengths must be in [0, 2^24)");
  return interpolate_launch(nullptr, d_in, h_off, n_utts, 1.0, 0.f, d_ip, d_vuv, as_stream(stream));
}

extern "C" int itts_assemble_cmp_f32(const float* d_sp, int64_t ld_sp, int n_sp, const float* d_lf0,
                                     const float* d_vuv, const float* d_bap, int64_t ld_bap,
                                     int n_bap, const int64_t* h_f_off, int n_utts, int add_deltas,
                                     float* d_out, int64_t ld_out, void* stream) {
  ITTS_REQUIRE(h_f_off && n_utts >= 0, "null offsets");
  ITTS_REQUIRE(n_sp >= 0 && n_bap >= 0 && ld_sp >= n_sp && ld_bap >= n_bap, "bad sizes");
  const int width = (add_deltas ? 3 : 1) * (n_sp + 1 + n_bap) + 1;
  ITTS_REQUIRE(ld_out >= width, "ld_out smaller than the feature width");
  if (n_utts == 0 || h_f_off[n_utts] == 0) return ITTS_OK;
  ITTS_REQUIRE(d_lf0 && d_vuv && d_out && (n_sp == 0 || d_sp) && (n_bap == 0 || d_bap),
               "null pointer");
  hipStream_t s = as_stream(stream);
  itts::ScratchScope scratch_scope(s);
  int64_t* d_off = nullptr;
  int rc = upload_i64(h_f_off, n_utts + 1, &d_off, s);
  if (rc != ITTS_OK) return rc;
  CmpArgs a{d_sp, ld_sp, n_sp, d_lf0, d_vuv, d_bap, ld_bap, n_bap, d_off, n_utts, d_out, ld_out,
            h_f_off[n_utts]};
  const int64_t n = a.t_total * (n_sp + 2 + n_bap);
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 16384);
  if (add_deltas)
    hipLaunchKernelGGL(assemble_cmp_kernel<true>, dim3(blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(assemble_cmp_kernel<false>, dim3(blocks), dim3(256), 0, s, a);
  ITTS_LAUNCH_CHECK();
  ITTS_HIP_CHECK(itts::scratch_free(d_off, s));
  return ITTS_OK;
}

extern "C" int64_t itts_feature_stats_workspace_bytes(int width, int want_cov) {
  if (width <= 0) return 0;
  int64_t b = (int64_t)ST_CHUNKS * width * 2 * sizeof(double);
  if (want_cov) b += (int64_t)XTX_CHUNKS * width * width * sizeof(double);
  return b;
}

extern "C" int itts_feature_stats(const float* d_x, int64_t ld_x, int64_t n_rows, int col0,
                                  int width, int want_cov, int accumulate, double* d_sum,
                                  double* d_second, void* d_workspace, void* stream) {
  ITTS_REQUIRE(width > 0 && col0 >= 0 && ld_x >= col0 + width && n_rows >= 0, "bad sizes");
  ITTS_REQUIRE(d_sum && d_second && d_workspace && (n_rows == 0 || d_x), "null pointer");
  hipStream_t s = as_stream(stream);
  itts::ScratchScope scratch_scope(s);
  if (n_rows == 0) {
    if (!accumulate) {
      ITTS_HIP_CHECK(hipMemsetAsync(d_sum, 0, width * sizeof(double), s));
      ITTS_HIP_CHECK(hipMemsetAsync(d_second, 0, (want_cov ? (int64_t)width * width : width) *
                                    sizeof(double), s));
    }
    return ITTS_OK;
  }
  double* part_sum = (double*)d_workspace;
  double* part_sq = part_sum + (int64_t)ST_CHUNKS * width;
  double* part_xtx = part_sq + (int64_t)ST_CHUNKS * width;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(ST_CHUNKS), dim3(256), 0, s, d_x, ld_x, n_rows,
                     col0, width, part_sum, want_cov ? (double*)nullptr : part_sq);
  ITTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_parts_kernel, dim3((width + 255) / 256), dim3(256), 0, s, part_sum,
                     ST_CHUNKS, (int64_t)width, d_sum, accumulate);
  ITTS_LAUNCH_CHECK();
  if (!want_cov) {
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((width + 255) / 256), dim3(256), 0, s, part_sq,
                       ST_CHUNKS, (int64_t)width, d_second, accumulate);
    ITTS_LAUNCH_CHECK();
    return ITTS_OK;
  }
  const int tiles = (width + 15) / 16;
  hipLaunchKernelGGL(xtx_partial_kernel, dim3(tiles * tiles, XTX_CHUNKS), dim3(64), 0, s, d_x, ld_x,
                     n_rows, col0, width, part_xtx);
  ITTS_LAUNCH_CHECK();
  const int64_t n = (int64_t)width * width;
  hipLaunchKernelGGL(xtx_reduce_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, s, part_xtx,
                     width, d_second, accumulate);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}
