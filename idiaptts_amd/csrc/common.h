// Shared host/device helpers for the gfx950 kernels of libidiaptts_amd.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/idiaptts_amd.h"

namespace itts {

void set_error(const std::string& msg);

#define ITTS_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ::itts::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));               \
      return ITTS_E_HIP;                                                                  \
    }                                                                                     \
  } while (0)

#define ITTS_REQUIRE(cond, msg)                                                           \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      ::itts::set_error(std::string(__func__) + ": " + (msg));                            \
      return ITTS_E_INVALID;                                                              \
    }                                                                                     \
  } while (0)

#define ITTS_LAUNCH_CHECK() ITTS_HIP_CHECK(hipGetLastError())

// Host wait for everything queued on `s`, by polling: hipStreamSynchronize puts the thread to
// sleep when the wait gets long (a millisecond-scale kernel in front) and the wake-up then costs
// up to a millisecond of idle GPU; data-dependent loops that need a count back from the device
// between launches (mcep trip counts) poll an event instead.
static inline hipError_t itts_spin_sync(hipStream_t s) {
  hipEvent_t ev;
  hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e != hipSuccess) return e;
  e = hipEventRecord(ev, s);
  if (e == hipSuccess)
    while ((e = hipEventQuery(ev)) == hipErrorNotReady) {
    }
  (void)hipEventDestroy(ev);
  return e;
}

constexpr int kWave = 64;  // gfx950 wavefront

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum for blockDim.x a multiple of 64 (<= 1024). `red` needs 16 doubles of LDS.
// Result valid in every thread.
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < nw; ++i) s += red[i];  // same order in every thread: deterministic
  return s;
}

// Scratch memory of the entry points.  hipMallocAsync's pool hands its blocks back to the driver
// whenever a request does not fit what it has cached (one Harvest call empties it), and a fresh
// device allocation occasionally takes hundreds of milliseconds to seconds on this platform
// (measured: an analysis pass of 65 ms took 2.3 s right after the pool had been emptied; a whole
// bench process ran its analysis section at 330 ms per pass).  So the library keeps its own
// blocks: scratch_malloc returns a cached block of the request's size class (plain hipMalloc the
// first time), scratch_free marks it free behind an event on the caller's stream; a block that
// moves to another stream waits for that event first (stream-ordered like hipMallocAsync /
// hipFreeAsync).  At most ITTS_POOL_KEEP_GB (default 64) stay cached; itts_release_scratch() frees
// everything that is not in use.
hipError_t scratch_malloc(void** out, size_t bytes, hipStream_t s);
hipError_t scratch_free(void* p, hipStream_t s);

// Every entry point that takes scratch opens a ScratchScope first: blocks obtained (on this thread)
// while it is alive and not handed back by scratch_free -- the early returns of ITTS_REQUIRE /
// ITTS_HIP_CHECK / ITTS_LAUNCH_CHECK between a scratch_malloc and its scratch_free -- are released
// by its destructor, behind the work already queued on the stream.  Scopes nest (an entry point
// calling another one).
class ScratchScope {
 public:
  explicit ScratchScope(hipStream_t s);
  ~ScratchScope();
  ScratchScope(const ScratchScope&) = delete;
  ScratchScope& operator=(const ScratchScope&) = delete;
  void track(void* p);
  void untrack(void* p);

 private:
  hipStream_t stream_;
  ScratchScope* prev_;
  void* live_[64];
  int n_live_ = 0;
};

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace itts
