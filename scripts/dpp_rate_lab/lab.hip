// Issue rate of v_fmac_f64 with a DPP row_newbcast operand against the plain form (the product of
// mcls_solve_dpp_kernel): 32 independent accumulators per lane, 4 waves per SIMD, 4096 products per accumulator.
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 -o /tmp/dpp_rate lab.hip && /tmp/dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <bool DPP>
__global__ __launch_bounds__(256) void k(double* out, const double* in, int iters) {
  double acc[32];
  const double p = in[threadIdx.x], f = in[threadIdx.x + 256];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = in[i];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (DPP) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(p), "v"(f));
      else asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(p), "v"(f));
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  double *in, *out;
  hipMalloc(&in, 4096 * 8);
  hipMalloc(&out, 1024 * 4 * 256 * 8);
  hipMemset(in, 0, 4096 * 8);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 4096, blocks = 256 * 4;      // 4 workgroups of 4 waves per CU
  for (int rep = 0; rep < 2; ++rep) {
    for (int dpp = 0; dpp < 2; ++dpp) {
      hipEventRecord(a);
      if (dpp) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
      else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      const double flops = 2.0 * 64 * 32 * iters * 4.0 * blocks;
      printf("%s: %.3f ms, %.1f TFLOP/s\n", dpp ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64               ", ms, flops / ms / 1e9);
    }
  }
  return 0;
}
