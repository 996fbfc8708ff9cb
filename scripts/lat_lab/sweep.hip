// What a frame of the MLPG backward recurrence costs a lone wave, by variant (round 5).
//   hipcc -O3 --offload-arch=gfx950 sweep.hip -o sweep && ./sweep
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int PF = 8;
// LANES doubles a frame in LDS, FR frames; wave 0 sweeps, the other waves sleep-poll an LDS word like the kernel's helpers
template <int LANES, int MODE, bool POLLERS>
__global__ __launch_bounds__(512) void sweep(double* out, long long* cyc, double sl1, double sl2, double sd, int frames, int reps) {
  extern __shared__ double ring[];
  int* word = reinterpret_cast<int*>(ring + (size_t)frames * LANES);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < frames * LANES; i += blockDim.x) ring[i] = 1e-3 * (i % 97);
  if (threadIdx.x == 0) *word = 0;
  __syncthreads();
  if (wave != 0) {
    if (POLLERS) while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(2);
    return;
  }
  if (lane >= LANES) return;
  double x1 = 0.1, x2 = 0.2;
  const double a1 = sl1 * sd, a2 = sl2 * sd;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    double* p = ring + lane;
    double A[PF], B[PF];
    auto chain = [&](const double (&yv)[PF], double* q) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        double x;
        if (MODE == 0) x = (yv[i] - sl1 * x1 - sl2 * x2) * sd;                   // the kernel's expression: 3 dependent operations
        if (MODE == 1) x = ((yv[i] - sl2 * x2) - sl1 * x1) * sd;                 // x2 term first: 2 dependent operations
        if (MODE == 2) x = __builtin_fma(-a1, x1, __builtin_fma(-a2, x2, yv[i] * sd));   // scaled coefficients: 1 dependent operation
        q[i * LANES] = x;
        x2 = x1;
        x1 = x;
      }
    };
    const int nblocks = frames / PF;
#pragma unroll
    for (int i = 0; i < PF; ++i) A[i] = p[i * LANES];
    int k = 0;
    for (; k + 2 <= nblocks; k += 2) {
#pragma unroll
      for (int i = 0; i < PF; ++i) B[i] = p[(PF + i) * LANES];
      chain(A, p);
      double* pn = k + 2 < nblocks ? p + 2 * PF * LANES : p;
#pragma unroll
      for (int i = 0; i < PF; ++i) A[i] = pn[i * LANES];
      chain(B, p + PF * LANES);
      p += 2 * PF * LANES;
    }
  }
  long long t1 = clock64();
  out[lane] = x1 + x2;
  if (lane == 0) { cyc[0] = t1 - t0; __hip_atomic_store(word, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
}
template <int LANES, int MODE, bool POLLERS>
void run(const char* name, double* d, long long* c) {
  const int frames = 256 * 16 / LANES * 4 / 4;   // 16 lanes: 256 frames .. 64 lanes: 64 .. keep <= 64 KB
  const int fr = LANES == 16 ? 512 : (LANES == 32 ? 256 : 128);
  (void)frames;
  const int reps = 64;
  long long h = 0;
  for (int rep = 0; rep < 2; ++rep) {
    sweep<LANES, MODE, POLLERS><<<1, 512, (size_t)fr * LANES * 8 + 64>>>(d, c, 0.3, 0.1, 0.9, fr, reps);
    hipDeviceSynchronize();
  }
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-58s %.1f cycles per frame\n", name, (double)h / ((double)fr * reps));
}
int main() {
  double* d; long long* c;
  if (hipMalloc(&d, 4096) != hipSuccess || hipMalloc(&c, 64) != hipSuccess) return 1;
  run<16, 0, false>("16 lanes, 3-operation chain, other waves gone", d, c);
  run<16, 0, true>("16 lanes, 3-operation chain, 7 waves polling", d, c);
  run<16, 1, true>("16 lanes, 2-operation chain, 7 waves polling", d, c);
  run<16, 2, true>("16 lanes, 1-operation chain, 7 waves polling", d, c);
  run<64, 0, false>("64 lanes, 3-operation chain, other waves gone", d, c);
  run<64, 0, true>("64 lanes, 3-operation chain, 7 waves polling", d, c);
  run<64, 1, true>("64 lanes, 2-operation chain, 7 waves polling", d, c);
  run<64, 2, true>("64 lanes, 1-operation chain, 7 waves polling", d, c);
  run<32, 0, true>("32 lanes, 3-operation chain, 7 waves polling", d, c);
  return 0;
}
