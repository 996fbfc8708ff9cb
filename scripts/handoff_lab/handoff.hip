// Cross-workgroup hand-off inside one XCD, measured: is "plain stores + L1-bypassing loads" enough
// when producer and consumer share an L2, and what does a step of such an exchange cost?
// (Groundwork for a persistent recurrence kernel)
//
// 256 persistent workgroups; group = blockIdx % 8 (the XCD under round-robin dispatch; the kernel
// records the XCC_ID it really runs on), 32 workgroups per group.  Step s: every workgroup writes
// its 1 KB slice of the group's 32 KB buffer (parity s & 1), waits for its stores, adds to the
// group's counter; one lane polls the counter until all 32 have added; then all four waves read
// the whole 32 KB and compare with what the producers must have written.  Variants: stores plain or
// sc1, loads plain or sc1.  Output: stale 16-byte granules seen, microseconds per step.
// build: hipcc -O3 --offload-arch=gfx950 handoff.hip -o handoff ; run: ./handoff [steps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <bool SC1>
__device__ __forceinline__ void store16(u32x4* p, u32x4 v) {
  // s_nop behind the store: a VALU write of the data registers right after a 16-byte store reaches
  // the last lanes' data before the store has read it (the hazard the compiler pads for its own
  // stores but cannot see inside an asm statement; seen here as lanes 12-15 of every 16 holding the
  // next value)
  if (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 3" ::"v"(p), "v"(v) : "memory");
}
template <bool SC1>
__device__ __forceinline__ u32x4 load16(const u32x4* p) {
  u32x4 v;
  // load and wait in ONE statement: the compiler does not know that the result of a bare load asm
  // is not there yet and may copy the destination registers before a separate s_waitcnt
  if (SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}


// eight 16-byte loads in flight together, one wait (a wave's share of a step: 8 producers)
template <bool SC1>
__device__ __forceinline__ void load16x8(const u32x4* p, int stride, u32x4 (&v)[8]) {
  const u32x4 *p0 = p, *p1 = p + stride, *p2 = p + 2 * stride, *p3 = p + 3 * stride, *p4 = p + 4 * stride,
              *p5 = p + 5 * stride, *p6 = p + 6 * stride, *p7 = p + 7 * stride;
  if (SC1)
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7) : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %9, off\n\t"
                 "global_load_dwordx4 %2, %10, off\n\tglobal_load_dwordx4 %3, %11, off\n\t"
                 "global_load_dwordx4 %4, %12, off\n\tglobal_load_dwordx4 %5, %13, off\n\t"
                 "global_load_dwordx4 %6, %14, off\n\tglobal_load_dwordx4 %7, %15, off\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7) : "memory");
}

struct Args {
  u32x4* buf;            // [8 groups][2 parities][32 producers][64 granules]
  unsigned* counter;     // [8 groups], 64 B apart
  unsigned long long* stale;   // [256] per workgroup
  unsigned long long* torn;    // [256] per workgroup (inband_kernel)
  unsigned* xcc;         // [256] XCC_ID seen
  unsigned* gave_up;
  int steps;
};

template <bool ST_SC1, bool LD_SC1>
__global__ __launch_bounds__(256) void handoff_kernel(Args a) {
  const int group = blockIdx.x & 7, rank = blockIdx.x >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    a.xcc[blockIdx.x] = id & 0xf;
  }
  unsigned* ctr = a.counter + group * 16;
  unsigned long long stale = 0;
  for (int s = 0; s < a.steps; ++s) {
    u32x4* base = a.buf + ((size_t)(group * 2 + (s & 1)) * 32) * 64;
    if (wave == 0) {
      u32x4 v = {(unsigned)s, (unsigned)rank, (unsigned)lane, (unsigned)(s * 2654435761u + rank * 97u + lane)};
      store16<ST_SC1>(base + rank * 64 + lane, v);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = 32u * (unsigned)(s + 1);
      int budget = 1 << 22;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && --budget > 0) __builtin_amdgcn_s_sleep(2);
      if (budget <= 0) atomicExch(a.gave_up, 1u);
    }
    __syncthreads();
    // every wave reads a quarter of the 32 producers' slices, its eight loads in flight together
    {
      u32x4 v[8];
      load16x8<LD_SC1>(base + wave * 64 + lane, 4 * 64, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = wave + 4 * i;
        const unsigned expect = (unsigned)(s * 2654435761u + p * 97u + lane);
        if (v[i].x != (unsigned)s || v[i].y != (unsigned)p || v[i].z != (unsigned)lane || v[i].w != expect) ++stale;
      }
    }
    if (*a.gave_up) break;
  }
  for (int off = 32; off > 0; off >>= 1) stale += __shfl_xor(stale, off, 64);
  if (lane == 0) atomicAdd(a.stale + blockIdx.x, stale);
}


// No counter at all: every 16-byte granule carries its step number, and a consumer lane re-reads a
// granule until the number is the one it waits for ("tagged granules"; a 16-byte store was never
// seen torn on this chip, MI355X_MICROARCH.md).  The parity buffers make a step's granule differ
// from the one two steps earlier, so a stale L2 / L1 line cannot pass for the new one.
template <bool ST_SC1, bool LD_SC1>
__global__ __launch_bounds__(256) void tagged_kernel(Args a) {
  const int group = blockIdx.x & 7, rank = blockIdx.x >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long stale = 0;      // here: granules whose payload did not match although the tag did
  unsigned long long polls = 0;
  for (int s = 0; s < a.steps; ++s) {
    u32x4* base = a.buf + ((size_t)(group * 2 + (s & 1)) * 32) * 64;
    if (wave == 0) {
      u32x4 v = {(unsigned)s, (unsigned)rank, (unsigned)lane, (unsigned)(s * 2654435761u + rank * 97u + lane)};
      store16<ST_SC1>(base + rank * 64 + lane, v);
    }
    {
      int budget = 1 << 20;
      u32x4 v[8];
      for (;;) {
        load16x8<LD_SC1>(base + wave * 64 + lane, 4 * 64, v);
        ++polls;
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) ok = ok && v[i].x == (unsigned)s;
        if (__all(ok) || --budget <= 0) break;
        __builtin_amdgcn_s_sleep(4);
      }
      if (budget <= 0 && lane == 0) atomicExch(a.gave_up, 1u);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = wave + 4 * i;
        const unsigned expect = (unsigned)(s * 2654435761u + p * 97u + lane);
        if (v[i].x != (unsigned)s || v[i].y != (unsigned)p || v[i].z != (unsigned)lane || v[i].w != expect) ++stale;
      }
    }
    // a workgroup may not overwrite parity (s & 1) at step s + 2 before every reader of step s is done:
    // a workgroup writes step s + 1 only after ALL its waves have read step s (the barrier), and
    // every workgroup reads step s + 1 before it writes step s + 2 -- so by the time anybody writes
    // step s + 2, everybody has finished reading step s
    __syncthreads();
    if (*a.gave_up) break;
  }
  for (int off = 32; off > 0; off >>= 1) stale += __shfl_xor(stale, off, 64);
  if (lane == 0) atomicAdd(a.stale + blockIdx.x, stale);
  if (lane == 0 && wave == 0) a.xcc[blockIdx.x] = (unsigned)(polls / (unsigned long long)(a.steps > 0 ? a.steps : 1));
}


// Round 5: the product's in-band format (csrc/rnn_persist.h, persist_use_tag).  Every 32-bit word of a granule
// gives its lowest bit to a validity bit that is complementary between consecutive uses of a slot (1, 0, 1 ...;
// a cleared buffer reads 0 and the first use writes 1), so each word is validated by itself.  A consumer polls
// until all four bits are the current use's and then compares the payload: `stale` counts granules that passed
// the check with a wrong payload (must be 0, with or without 16-byte single-copy atomicity), `torn` counts
// polls that saw a granule whose four words did not all belong to the same use (information only).
template <bool ST_SC1, bool LD_SC1>
__global__ __launch_bounds__(256) void inband_kernel(Args a) {
  const int group = blockIdx.x & 7, rank = blockIdx.x >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long stale = 0, torn = 0, polls = 0;
  auto payload = [](int s, int p, int ln, int k) -> unsigned {
    return ((unsigned)s * 2654435761u + (unsigned)p * 40503u + (unsigned)ln * 97u + (unsigned)k * 0x9E3779B1u) & ~1u;
  };
  for (int s = 0; s < a.steps; ++s) {
    const unsigned bit = ((s >> 1) & 1) ? 0u : 1u;         // use index of parity slot s & 1 is s >> 1
    u32x4* base = a.buf + ((size_t)(group * 2 + (s & 1)) * 32) * 64;
    if (wave == 0) {
      u32x4 v = {payload(s, rank, lane, 0) | bit, payload(s, rank, lane, 1) | bit, payload(s, rank, lane, 2) | bit,
                 payload(s, rank, lane, 3) | bit};
      store16<ST_SC1>(base + rank * 64 + lane, v);
    }
    {
      int budget = 1 << 20;
      u32x4 v[8];
      for (;;) {
        load16x8<LD_SC1>(base + wave * 64 + lane, 4 * 64, v);
        ++polls;
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned bits = (v[i].x & 1u) + (v[i].y & 1u) + (v[i].z & 1u) + (v[i].w & 1u);
          ok = ok && bits == 4u * bit;
          if (bits != 0u && bits != 4u) ++torn;
        }
        if (__all(ok) || --budget <= 0) break;
        __builtin_amdgcn_s_sleep(4);
      }
      if (budget <= 0 && lane == 0) atomicExch(a.gave_up, 1u);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = wave + 4 * i;
        if (v[i].x != (payload(s, p, lane, 0) | bit) || v[i].y != (payload(s, p, lane, 1) | bit) ||
            v[i].z != (payload(s, p, lane, 2) | bit) || v[i].w != (payload(s, p, lane, 3) | bit)) ++stale;
      }
    }
    __syncthreads();
    if (*a.gave_up) break;
  }
  for (int off = 32; off > 0; off >>= 1) { stale += __shfl_xor(stale, off, 64); torn += __shfl_xor(torn, off, 64); }
  if (lane == 0) { atomicAdd(a.stale + blockIdx.x, stale); atomicAdd(a.torn + blockIdx.x, torn); }
  if (lane == 0 && wave == 0) a.xcc[blockIdx.x] = (unsigned)(polls / (unsigned long long)(a.steps > 0 ? a.steps : 1));
}

template <bool ST_SC1, bool LD_SC1, int TAGGED = 0>
static void run(const char* name, Args a) {
  CHECK(hipMemset(a.buf, TAGGED == 2 ? 0 : 0xff, (size_t)8 * 2 * 32 * 64 * 16));
  CHECK(hipMemset(a.torn, 0, 256 * 8));
  CHECK(hipMemset(a.counter, 0, 8 * 64));
  CHECK(hipMemset(a.stale, 0, 256 * 8));
  CHECK(hipMemset(a.gave_up, 0, 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  if (TAGGED == 2) hipLaunchKernelGGL((inband_kernel<ST_SC1, LD_SC1>), dim3(256), dim3(256), 0, 0, a);
  else if (TAGGED) hipLaunchKernelGGL((tagged_kernel<ST_SC1, LD_SC1>), dim3(256), dim3(256), 0, 0, a);
  else hipLaunchKernelGGL((handoff_kernel<ST_SC1, LD_SC1>), dim3(256), dim3(256), 0, 0, a);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> st(256);
  std::vector<unsigned> xcc(256);
  unsigned gave = 0;
  CHECK(hipMemcpy(st.data(), a.stale, 256 * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(xcc.data(), a.xcc, 256 * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&gave, a.gave_up, 4, hipMemcpyDeviceToHost));
  unsigned long long total = 0;
  for (auto v : st) total += v;
  int mismapped = 0;
  if (!TAGGED) for (int b = 0; b < 256; ++b) mismapped += (xcc[b] != xcc[b & 7]);   // same group -> same XCC?
  if (TAGGED) mismapped = (int)xcc[0];      // polls per step of workgroup 0's first wave instead
  if (TAGGED == 2) {
    std::vector<unsigned long long> tn(256);
    CHECK(hipMemcpy(tn.data(), a.torn, 256 * 8, hipMemcpyDeviceToHost));
    unsigned long long torn = 0;
    for (auto v : tn) torn += v;
    printf("%-28s steps %d  %.2f us/step  granules that passed the check with a wrong payload %llu of %llu  "
           "(polls that saw a granule with mixed validity bits: %llu; polls per step %d, gave up: %u)\n",
           name, a.steps, ms * 1e3 / a.steps, total, (unsigned long long)a.steps * 256 * 32 * 64, torn, mismapped, gave);
    return;
  }
  printf("%-28s steps %d  %.2f us/step  stale granules %llu of %llu  (groups split over XCCs: %d workgroups, gave up: %u)\n",
         name, a.steps, ms * 1e3 / a.steps, total, (unsigned long long)a.steps * 256 * 32 * 64, mismapped, gave);
}

int main(int argc, char** argv) {
  Args a{};
  a.steps = argc > 1 ? atoi(argv[1]) : 20000;
  CHECK(hipMalloc((void**)&a.buf, (size_t)8 * 2 * 32 * 64 * 16));
  CHECK(hipMalloc((void**)&a.counter, 8 * 64));
  CHECK(hipMalloc((void**)&a.stale, 256 * 8));
  CHECK(hipMalloc((void**)&a.torn, 256 * 8));
  CHECK(hipMalloc((void**)&a.xcc, 256 * 4));
  CHECK(hipMalloc((void**)&a.gave_up, 4));
  for (int rep = 0; rep < 2; ++rep) {
    run<true, true>("stores sc1,   loads sc1", a);
    run<false, true>("stores plain, loads sc1", a);
    run<false, false>("stores plain, loads plain", a);
    run<true, false>("stores sc1,   loads plain", a);
    run<true, true, 1>("tagged: sc1 / sc1", a);
    run<false, true, 1>("tagged: plain / sc1", a);
    run<false, true, 2>("in-band bits: plain / sc1", a);
    run<true, true, 2>("in-band bits: sc1 / sc1", a);
  }
  return 0;
}
