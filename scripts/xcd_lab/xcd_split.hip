// Can two kernels share the chip by XCD?  (Groundwork for running the weight-gradient products of a recurrent
// layer beside the persistent recurrence of the next one: DESIGN.md section 13.)
//
// Kernel `occupy`: 256 workgroups x 512 threads with 130 KB of LDS each (one per CU, nothing else fits beside it
// on the LDS side); those that find themselves on XCDs 4 .. 7 leave at once, the others spin for `ms`
// milliseconds -- the footprint of a persistent recurrence confined to four XCDs.
// Kernel `probe`: launched on ANOTHER stream while `occupy` spins, 512 workgroups x 256 threads with 72 KB of LDS
// (the GEMM's footprint, two per CU); every workgroup stamps the wall clock and its XCC_ID, the ones on XCDs
// 4 .. 7 then work for 200 us, the others leave.
// Questions: (1) do the probe's workgroups for XCDs 4 .. 7 start while XCDs 0 .. 3 are full, or does the
// dispatcher stall behind the first workgroup it cannot place?  (2) when does the probe kernel END (its
// workgroups for XCDs 0 .. 3 can only be placed once `occupy` leaves)?  (3) is blockIdx % 8 the XCC_ID for both?
// build: hipcc -O3 --offload-arch=gfx950 xcd_split.hip -o xcd_split ; run: ./xcd_split [ms]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  return id & 0xf;
}

struct Stamp { unsigned long long t0, t1; unsigned xcc, pad; };

__global__ __launch_bounds__(512) void occupy(Stamp* st, unsigned long long ticks, int keep_below) {
  extern __shared__ char lds[];
  const unsigned x = xcc_id();
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) { st[blockIdx.x].t0 = t0; st[blockIdx.x].xcc = x; }
  if ((int)x < keep_below) {
    lds[threadIdx.x] = 1;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  }
  if (threadIdx.x == 0) st[blockIdx.x].t1 = wall_clock64();
}

__global__ __launch_bounds__(256) void probe(Stamp* st, unsigned long long ticks, int work_from) {
  extern __shared__ char lds[];
  const unsigned x = xcc_id();
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) { st[blockIdx.x].t0 = t0; st[blockIdx.x].xcc = x; }
  if ((int)x >= work_from) {
    lds[threadIdx.x] = 1;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  }
  if (threadIdx.x == 0) st[blockIdx.x].t1 = wall_clock64();
}

static void report(const char* name, const std::vector<Stamp>& s, unsigned long long origin) {
  int mism = 0;
  for (size_t b = 0; b < s.size(); ++b) mism += (s[b].xcc != (unsigned)(b % 8));
  printf("%s: %zu workgroups, blockIdx %% 8 != XCC_ID for %d of them\n", name, s.size(), mism);
  for (unsigned x = 0; x < 8; ++x) {
    double a0 = 1e30, a1 = -1e30, e1 = -1e30;
    int n = 0;
    for (const Stamp& q : s)
      if (q.xcc == x) {
        a0 = std::min(a0, (double)(long long)(q.t0 - origin) / 100.0);
        a1 = std::max(a1, (double)(long long)(q.t0 - origin) / 100.0);
        e1 = std::max(e1, (double)(long long)(q.t1 - origin) / 100.0);
        ++n;
      }
    printf("  XCD %u: %3d workgroups, first start %9.1f us, last start %9.1f us, last end %9.1f us\n", x, n, a0, a1, e1);
  }
}

int main(int argc, char** argv) {
  const double ms = argc > 1 ? atof(argv[1]) : 5.0;
  Stamp *so, *sp, *sp2;
  CHECK(hipMalloc((void**)&so, 256 * sizeof(Stamp)));
  CHECK(hipMalloc((void**)&sp, 512 * sizeof(Stamp)));
  CHECK(hipMalloc((void**)&sp2, 512 * sizeof(Stamp)));
  CHECK(hipFuncSetAttribute((const void*)occupy, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024));
  CHECK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
  hipStream_t a, b;
  CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipMemset(so, 0, 256 * sizeof(Stamp)));
    CHECK(hipMemset(sp, 0, 512 * sizeof(Stamp)));
    CHECK(hipMemset(sp2, 0, 512 * sizeof(Stamp)));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(occupy, dim3(256), dim3(512), 130 * 1024, a, so, (unsigned long long)(ms * 1e5), 4);   // 100 MHz clock
    hipLaunchKernelGGL(probe, dim3(512), dim3(256), 72 * 1024, b, sp, 20000ull, 4);
    hipLaunchKernelGGL(probe, dim3(512), dim3(256), 72 * 1024, b, sp2, 20000ull, 4);    // behind the first on its stream
    CHECK(hipDeviceSynchronize());
    std::vector<Stamp> ho(256), hp(512), hp2(512);
    CHECK(hipMemcpy(ho.data(), so, 256 * sizeof(Stamp), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hp.data(), sp, 512 * sizeof(Stamp), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hp2.data(), sp2, 512 * sizeof(Stamp), hipMemcpyDeviceToHost));
    unsigned long long origin = ~0ull;
    for (const Stamp& q : ho) origin = std::min(origin, q.t0);
    printf("---- pass %d: occupy spins %.1f ms on XCDs 0 .. 3; times relative to its first workgroup\n", rep, ms);
    report("occupy", ho, origin);
    report("probe (other stream)", hp, origin);
    report("second probe (behind the first)", hp2, origin);
  }
  return 0;
}
