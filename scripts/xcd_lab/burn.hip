// A kernel that keeps the matrix units of all CUs busy at a chosen duty cycle for a chosen time, behind a C entry
// point (loaded with ctypes by scripts/load_step_probe.py): does a GEMM that follows a stretch of LIGHT load run slower
// because of the load step, and does matrix work of no use during that stretch prevent it?
// build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC burn.hip -o libburn.so
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// every workgroup: until `ticks` of the 100-MHz clock have passed, `on` bursts of 32 fp32 MFMAs (16x16x4: about 1 us
// of the unit per wave) followed by `off` sleeps of ~0.5 us
__global__ __launch_bounds__(256) void burn_kernel(unsigned long long ticks, int on, int off, float* sink) {
  const unsigned long long t0 = wall_clock64();
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const float a = (float)threadIdx.x * 1e-9f, b = 1.0f + (float)blockIdx.x * 1e-9f;
  while (wall_clock64() - t0 < ticks) {
    for (int i = 0; i < on; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
      }
    }
    for (int i = 0; i < off; ++i) __builtin_amdgcn_s_sleep(16);      // 16 x 64 clocks
  }
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) sink[0] = 1.f;
}

extern "C" int burn_launch(double ms, int on, int off, int workgroups, void* d_sink, void* stream) {
  hipLaunchKernelGGL(burn_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (unsigned long long)(ms * 1e5), on,
                     off, (float*)d_sink);
  return (int)hipGetLastError();
}
