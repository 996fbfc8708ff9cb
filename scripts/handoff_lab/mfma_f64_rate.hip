// Issue rate of v_mfma_f64_16x16x4_f64 (and of v_fma_f64 beside it): clocks per instruction with 1, 2
// and 4 waves per SIMD, eight independent accumulators per wave.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_f64_rate mfma_f64_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void mfma_kernel(double* out, long long* clk, int iters) {
  f64x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0.0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void fma_kernel(double* out, long long* clk, int iters) {
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  const long long t1 = clock64();
  double s = 0.0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
  double* out; long long* clk;
  hipMalloc(&out, 1 << 24); hipMalloc(&clk, 4096 * 8);
  const int iters = 20000;
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
    const int blocks = 256 * (256 * waves_per_simd / threads);
    for (int kind = 0; kind < 2; ++kind) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
      else hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double n_instr = (double)blocks * (threads / 64) * iters * 8;        // wave instructions
      const double flop = n_instr * (kind == 0 ? 2048.0 : 128.0);
      printf("%s, %d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s, %.1f ns per wave instruction and SIMD\n",
             kind == 0 ? "v_mfma_f64_16x16x4_f64" : "v_fma_f64             ", waves_per_simd, ms, flop / ms / 1e9,
             ms * 1e6 / (iters * 8.0 * waves_per_simd));
    }
  }
  return 0;
}
