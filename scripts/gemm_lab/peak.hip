// What the fp32 matrix pipe delivers in a bare register loop on this device, and at which clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(256) void peak_kernel(float* out, int iters, uint64_t* stamps, float seed) {
  const uint64_t t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    uint32_t x = (threadIdx.x * 8 + i + blockIdx.x * 2048) * 2654435761u;
    x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 13;
    a[i] = ((float)(x >> 8) / 16777216.f - 0.5f) * seed;
    b[i] = ((float)((x * 31u) >> 8) / 16777216.f - 0.5f);
  }
  if (SHAPE == 32) {
    f32x16 c0 = {0}, c1 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[i], a[i], c1, 0, 0, 0);
      }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i], a[i], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[i], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i], b[i], c3, 0, 0, 0);
      }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
  }
}
int main() {
  float* out; uint64_t* st;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&st, 4096 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape : {32, 16}) for (int wgs : {256, 512, 1024}) for (float seed : {1.f, 0.f}) {
    const int iters = 2000, reps = 30;
    auto launch = [&]() {
      if (shape == 32) hipLaunchKernelGGL(peak_kernel<32>, dim3(wgs), dim3(256), 0, 0, out, iters, st, seed);
      else hipLaunchKernelGGL(peak_kernel<16>, dim3(wgs), dim3(256), 0, 0, out, iters, st, seed);
    };
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = shape == 32 ? 16.0 * 32 * 32 * 2 * 2 : 32.0 * 16 * 16 * 4 * 2;
    const double flops = (double)wgs * 4 * iters * per * reps;
    std::vector<uint64_t> h(2 * wgs);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < wgs; ++i) ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double cyc_per_mfma = (double)h[0] / (iters * (shape == 32 ? 16.0 : 32.0));
    printf("mfma %dx%d  %4d WGs (%.0f waves/SIMD) operands %s: %.1f TF, clock median %.3f GHz, %.1f cycles per MFMA (wg 0), %.2f ms per launch\n",
           shape, shape, wgs, wgs / 256.0, seed ? "random" : "a=0", flops / ms * 1e-9, ghz[wgs / 2], cyc_per_mfma, ms / reps);
  }
  return 0;
}
