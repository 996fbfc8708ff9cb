// Dependent-issue latency of the fp64 vector instructions on gfx950 (round 5: why a three-operation recurrence costs
// 120 cycles a step).   hipcc -O3 --offload-arch=gfx950 lat.hip -o lat && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int MODE>
__global__ void chain(double* out, long long* cyc, double a, double b, int active) {
  if ((int)threadIdx.x >= active) return;
  double x = out[threadIdx.x], y = out[threadIdx.x + 64];
  float xf = (float)x, af = (float)a, bf = (float)b;
  long long t0 = clock64();
#pragma unroll 64
  for (int i = 0; i < N; ++i) {
    if (MODE == 0) x = __builtin_fma(x, a, b);                    // v_fma_f64 chain
    if (MODE == 1) x = x * a;                                     // v_mul_f64 chain
    if (MODE == 2) x = x + a;                                     // v_add_f64 chain
    if (MODE == 3) xf = __builtin_fmaf(xf, af, bf);               // v_fma_f32 chain
    if (MODE == 4) { x = __builtin_fma(x, a, b); y = __builtin_fma(y, a, b); }   // two independent chains
    if (MODE == 5) { x = __builtin_fma(-a, x, y); y = x * b; }     // fma -> mul alternating
    asm volatile("" : "+v"(x), "+v"(y), "+v"(xf));
  }
  long long t1 = clock64();
  out[threadIdx.x] = x + y + xf;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* d; long long* c; hipMalloc(&d, 4096); hipMalloc(&c, 64); hipMemset(d, 0, 4096);
  long long h[2];
  const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32", "2 x v_fma_f64 independent (per pair)", "fma_f64 + mul_f64 (per pair)"};
  for (int active : {64, 16}) for (int waves : {1, 2}) {
    for (int m = 0; m < 6; ++m) {
      for (int rep = 0; rep < 2; ++rep) {
        switch (m) {
          case 0: chain<0><<<1, 64 * waves>>>(d, c, 0.999, 1e-3, active + 64 * (waves - 1)); break;
          case 1: chain<1><<<1, 64 * waves>>>(d, c, 0.999, 1e-3, active + 64 * (waves - 1)); break;
          case 2: chain<2><<<1, 64 * waves>>>(d, c, 0.999, 1e-3, active + 64 * (waves - 1)); break;
          case 3: chain<3><<<1, 64 * waves>>>(d, c, 0.999, 1e-3, active + 64 * (waves - 1)); break;
          case 4: chain<4><<<1, 64 * waves>>>(d, c, 0.999, 1e-3, active + 64 * (waves - 1)); break;
          case 5: chain<5><<<1, 64 * waves>>>(d, c, 0.999, 1e-3, active + 64 * (waves - 1)); break;
        }
        hipDeviceSynchronize();
      }
      hipMemcpy(h, c, 8, hipMemcpyDeviceToHost);
      printf("lanes %2d waves %d  %-40s %.1f cycles per step\n", active, waves, names[m], (double)h[0] / N);
    }
  }
  return 0;
}
