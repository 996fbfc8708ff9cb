// Persistent recurrences of the LSTM / GRU layers (included by lstm.hip and gru.hip): what
// torch.nn.LSTM / GRU do between pack_padded_sequence and pad_packed_sequence in
// rnn_dyn/RNNWrapper.py:89-102 and what autograd does on the way back -- one launch per layer for
// the forward recurrence, one for the backward recurrence (the step kernels of lstm.hip / gru.hip
// remain for other sizes and as the fallback).
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <map>
#include <mutex>

#include "context.h"
#include "rnn_common.h"

namespace itts {

// ---- persistent forward recurrence, one (direction, 16-row batch tile) per XCD --------------------
// Taken when H = 512 and the device has 256 CUs (ITTS_RNN_PERSISTENT=0 keeps the step kernels);
// 8 / ndir batch tiles of 16 rows per launch, larger batches in rounds; G = 4: LSTM, G = 3: GRU.  The
// step kernels pay 3.4 us of launch boundary per step for the grid-wide exchange of h; here
// the 32 workgroups an XCD holds (blockIdx % 8 = XCD under round-robin dispatch) keep one
// recurrence to themselves for all T steps: workgroup c owns hidden units 16 c .. 16 c + 15 (all G
// gates: a G x 32 KB image of its W_hh rows stays in LDS in the order the MFMA lanes read it), the
// cell state stays in registers, and h travels through the XCD's L2 as self-validating 16-byte
// granules (four values, each carrying a validity bit in its lowest mantissa bit -- persist_use_tag below;
// until round 4: pairs (P, P ^ mask(step))) that a consumer lane re-reads until all four bits are the
// current use's -- no counter, no fence (scripts/handoff_lab: 1.2 us per step).  A polling budget turns a missing workgroup into an abort
// flag instead of a hang.
constexpr int PH = 512;                          // hidden size this kernel is built for
constexpr int P_PART_FLOATS = 4 * 4 * 16 * 17;   // partial gate sums [wave][gate][row][unit + pad]
constexpr int persist_w_bytes(int G) { return G * 4 * 8 * 64 * 16; }   // W_hh image: [gate][k quarter][quad][lane] float4
constexpr int persist_lds_bytes(int G) { return persist_w_bytes(G) + P_PART_FLOATS * 4; }

struct RnnPersistArgs {
  const float* gin;
  const float* whh;
  const float* bhh;     // GRU: [ndir][3H]
  const float* h0;
  const float* c0;
  const int* lengths;
  const int* row_off;
  const int* rev_row;
  float* y;
  float* gates;
  float* csave;
  float* hn;
  float* cn;
  uint4* xchg;      // [8 groups][4 step slots][32 producers][64 granules]
  int* abort_flag;
  int T, B, ndir, ntiles;
  int tile0;        // first batch tile of this launch (batches of more than 8 / ndir tiles take several)
};

// asm operands must be native 128-bit vectors (a struct type such as uint4 gives the register
// allocator no reason to keep its four components in consecutive registers)
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));

// one granule of one producer: what a waiting wave re-reads (full-width loads per poll from every waiting
// wave of 32 CUs crowd the L2 the producers' stores have to get through)
__device__ __forceinline__ void persist_load_one(const uint4* p, pu32x4& v) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
}

// In-band validity bits (both recurrences): the lowest mantissa bit of EACH of a granule's four values says which
// use of its slot the value belongs to -- consecutive uses of a slot carry complementary bits (0xF, 0x0, 0xF ... over
// the granule, the first use after a clear being 0xF), so every one of the four words is validated by itself: a torn
// or partly stale 16-byte read, whatever mixture of the previous and the current use it holds, fails the check in the
// stale words.  (Round 4 used a 4-bit counter spread over the words; consecutive uses differed in ONE word only, so a
// torn read could pass -- ADVICE r4.)  A value from TWO uses back carries the current bit again; a reader cannot see
// one: it has itself read the use in between from the same address, and reads of one location served by the L2
// (sc1: the L1 is bypassed) never go back in that location's modification order.  A cleared buffer reads as 0x0 and
// is what the second use expects -- by then every reader has seen the first use's 0xF at that address, same argument.
// Assumes nothing about 16-byte single-copy atomicity (scripts/handoff_lab re-run on this format: profiles/r5_handoff_lab.txt).
// Inf in an exchanged value becomes NaN when its bit is set: a recurrence that produced Inf is lost either way.
__device__ __forceinline__ unsigned persist_use_tag(int use) { return (use & 1) ? 0u : 0xFu; }
__device__ __forceinline__ unsigned persist_tag_of(const pu32x4 g) {
  return (g.x & 1u) | ((g.y & 1u) << 1) | ((g.z & 1u) << 2) | ((g.w & 1u) << 3);
}
// eight 16-byte L1-bypassing loads in flight together (one granule of each of eight producers' tiles)
__device__ __forceinline__ void persist_load8(const uint4* p, int stride, pu32x4 (&v)[8]) {
  const uint4 *p0 = p, *p1 = p + stride, *p2 = p + 2 * stride, *p3 = p + 3 * stride, *p4 = p + 4 * stride,
              *p5 = p + 5 * stride, *p6 = p + 6 * stride, *p7 = p + 7 * stride;
  asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
               "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
               "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
               "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
               : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)
               : "memory");
}

// Loads and stores of data touched once (gates, gradients): marked non-temporal so that they do not
// push the exchange tiles, which are re-used every other step, out of the XCD's L2.
typedef float pf32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 persist_stream_load4(const float4* p) {
  const pf32x4 v = __builtin_nontemporal_load(reinterpret_cast<const pf32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// Workgroup barrier that only waits for this wave's LDS traffic: __syncthreads() also waits for every
// global load and store in flight (vmcnt(0)), which puts a trip to HBM in front of the barrier when
// loads for the NEXT step were just requested.
__device__ __forceinline__ void persist_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifndef PERSIST_TRACE
#define PERSIST_TRACE 0
#endif
template <int G>
__global__ __launch_bounds__(256) void rnn_persist_fwd_kernel(RnnPersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) char psm[];
  float4* Wl = reinterpret_cast<float4*>(psm);
  float* Pp = reinterpret_cast<float*>(psm + persist_w_bytes(G));
  const int group = blockIdx.x & 7, cu = blockIdx.x >> 3;
  const int tiles_per_dir = 8 / a.ndir;
  const int dir = group / tiles_per_dir, tile = a.tile0 + group % tiles_per_dir;
  if (tile >= a.ntiles) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int H = PH, G4 = G * PH;      // (G4: gate rows per direction)
  const int j0 = cu * 16;

  // the image of this workgroup's W_hh rows
  {
    const float* W = a.whh + (size_t)dir * G4 * H;
    for (int idx = threadIdx.x; idx < G * 4 * 8 * 64; idx += 256) {
      const int ln = idx & 63, qd = (idx >> 6) & 7, w2 = (idx >> 9) & 3, gg = idx >> 11;
      const float* src = W + (size_t)(gg * H + j0 + (ln & 15)) * H + 128 * w2 + 16 * qd + (ln >> 4);
      Wl[idx] = make_float4(src[0], src[4], src[8], src[12]);
    }
  }
  // this thread's (row, unit) of the cell update
  const int r = threadIdx.x >> 4, u = threadIdx.x & 15;
  const int b = tile * 16 + r, j = j0 + u;
  const bool valid = b < a.B;
  const int len = valid ? a.lengths[b] : 0;
  const int t_tile = a.lengths[tile * 16];          // the tile's longest row: how long this group runs
  float c = (valid && a.c0) ? a.c0[dir * H + j] : 0.f;
  float h = (valid && a.h0) ? a.h0[dir * H + j] : 0.f;
  float bh0 = 0.f, bh1 = 0.f, bh2 = 0.f;
  if (G == 3 && a.bhh) { bh0 = a.bhh[dir * G4 + j]; bh1 = a.bhh[dir * G4 + H + j]; bh2 = a.bhh[dir * G4 + 2 * H + j]; }
  (void)bh0; (void)bh1; (void)bh2;
  uint4* xg = a.xchg + (size_t)group * 4 * 32 * 64;
  // Every wave publishes the h of its own four rows: thread (r, u) = lane (r & 3) * 16 + u of wave
  // r >> 2 holds h[r][u]; granule (row r', k quarter kq) of this workgroup's block wants units
  // kq, 4 + kq, 8 + kq, 12 + kq of row r' -- four lanes of the same wave, fetched by shuffles; lanes
  // 0 .. 15 store the wave's sixteen granules (P and the check copy).  No LDS, no barrier: a wave
  // of this workgroup cannot get past its next poll before all four have published.
  auto publish = [&](int step, float hval) {
    const int rl = (lane >> 2) & 3, kq = lane & 3;          // meaningful for lanes 0 .. 15
    const float p0 = __shfl(hval, rl * 16 + kq, 64), p1 = __shfl(hval, rl * 16 + 4 + kq, 64),
                p2 = __shfl(hval, rl * 16 + 8 + kq, 64), p3 = __shfl(hval, rl * 16 + 12 + kq, 64);
    if (lane < 16) {
      // the validity bit rides in the lowest mantissa bit of each of the four values (until round 4 a check copy
      // P ^ mask(step) travelled beside every granule: twice the bytes to poll)
      const unsigned t = persist_use_tag(step >> 2);
      const uint4 P = make_uint4((__float_as_uint(p0) & ~1u) | (t & 1u), (__float_as_uint(p1) & ~1u) | ((t >> 1) & 1u),
                                 (__float_as_uint(p2) & ~1u) | ((t >> 2) & 1u), (__float_as_uint(p3) & ~1u) | ((t >> 3) & 1u));
      uint4* dst = xg + ((size_t)(step & 3) * 32 + cu) * 64 + kq * 16 + 4 * wv + rl;
      dst[0] = P;
    }
  };
  publish(0, h);

  const size_t ldg = (size_t)a.ndir * G4, ldh = (size_t)a.ndir * H;
  auto row_at = [&](int st) -> size_t {
    return dir == 0 ? (size_t)(a.row_off[st] + b) : (size_t)a.rev_row[(size_t)st * a.B + b];
  };
  // The row index of a step is requested two steps ahead as ONE load of a raw int (forward direction:
  // row_off[step], + b when it is used; reverse: rev_row[step][b]): anything computed from it at once
  // would wait for it at once -- in front of the products, with the projections' loads in flight.
  const int* rtab = dir == 0 ? a.row_off : a.rev_row + b;
  const size_t rstride = dir == 0 ? 1 : (size_t)a.B;
  const int radd = dir == 0 ? b : 0;
  size_t row_cur = 0;
  int rnew = 0;
  float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f, gc3 = 0.f;
  if (0 < len) {
    row_cur = row_at(0);
    const float* gi = a.gin + row_cur * ldg + (size_t)dir * G4 + j;
    gc0 = gi[0]; gc1 = gi[H]; gc2 = gi[2 * H]; gc3 = G == 4 ? gi[3 * H] : 0.f;
  }
  if (1 < len) rnew = rtab[rstride];
  unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tprev = 0;
#define PT(i) do { if (PERSIST_TRACE) { const unsigned long long tn = wall_clock64(); tacc[i] += tn - tprev; tprev = tn; } } while (0)
  if (PERSIST_TRACE) tprev = wall_clock64();
  for (int s = 0; s < t_tile; ++s) {
    const bool act = s < len;
    // packed row and input projections of this thread's element: requested one and two steps ahead,
    // BEHIND the poll of the step before (the poll's wait covers every load in flight, so a load
    // issued in front of it would hold it back by a trip to HBM)
    const size_t row = row_cur;
    const float g0 = gc0, g1 = gc1, g2 = gc2, g3 = gc3;
    // h_{s-1} of the whole tile: this wave's k quarter comes from producers 8 wv .. 8 wv + 7
    pu32x4 pv[8];
    {
      const uint4* src = xg + ((size_t)(s & 3) * 32 + 8 * wv) * 64 + lane;
      const unsigned m = persist_use_tag(s >> 2);
      int budget = 1 << 16;      // ~50 ms of polling at most
      for (;;) {
        // optimistic: usually everything is there (one trip); otherwise wait on one granule per
        // producer (lane i & 7 watches producer 8 wv + (i & 7)) and fetch again
        persist_load8(src, 64, pv);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) ok = ok && persist_tag_of(pv[i]) == m;
        if (__all(ok)) break;
        const uint4* watch = xg + ((size_t)(s & 3) * 32 + 8 * wv + (lane & 7)) * 64;
        bool gave_up = false;
        for (;;) {
          pu32x4 wp;
          persist_load_one(watch, wp);
          if (__all(persist_tag_of(wp) == m)) break;
          if (--budget <= 0 || *reinterpret_cast<volatile int*>(a.abort_flag)) { gave_up = true; break; }
          __builtin_amdgcn_s_sleep(4);
        }
        if (gave_up || --budget <= 0) {
          if (lane == 0) atomicExch(a.abort_flag, 1);
          break;
        }
      }
    }
    // step s + 1's input projections (their row index arrived a step ago; pinned HERE, where the poll
    // has just waited for everything in flight -- the compiler does not know that and would otherwise
    // wait at its first use, behind the loads below) and step s + 2's row index
    int rraw = rnew;
    asm volatile("" : "+v"(rraw));
    const size_t row_n1 = (size_t)(rraw + radd);
    float gn0 = 0.f, gn1 = 0.f, gn2 = 0.f, gn3 = 0.f;
    if (s + 1 < len) {
      const float* gi = a.gin + row_n1 * ldg + (size_t)dir * G4 + j;
      gn0 = gi[0]; gn1 = gi[H]; gn2 = gi[2 * H]; gn3 = G == 4 ? gi[3 * H] : 0.f;
    }
    if (s + 2 < len) rnew = rtab[(size_t)(s + 2) * rstride];
    PT(0);
    f32x4 acc[G];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) acc[gg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qd = 0; qd < 8; ++qd) {
      const float ax = __uint_as_float(pv[qd].x), ay = __uint_as_float(pv[qd].y), az = __uint_as_float(pv[qd].z),
                  aw = __uint_as_float(pv[qd].w);
      float4 bw[G];
#pragma unroll
      for (int gg = 0; gg < G; ++gg) bw[gg] = Wl[((gg * 4 + wv) * 8 + qd) * 64 + lane];
      // the four gates are four independent accumulator chains: an MFMA never waits for the one before it
#pragma unroll
      for (int gg = 0; gg < G; ++gg) acc[gg] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bw[gg].x, acc[gg], 0, 0, 0);
#pragma unroll
      for (int gg = 0; gg < G; ++gg) acc[gg] = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, bw[gg].y, acc[gg], 0, 0, 0);
#pragma unroll
      for (int gg = 0; gg < G; ++gg) acc[gg] = __builtin_amdgcn_mfma_f32_16x16x4f32(az, bw[gg].z, acc[gg], 0, 0, 0);
#pragma unroll
      for (int gg = 0; gg < G; ++gg) acc[gg] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, bw[gg].w, acc[gg], 0, 0, 0);
    }
#pragma unroll
    for (int gg = 0; gg < G; ++gg)
#pragma unroll
      for (int e = 0; e < 4; ++e) Pp[((wv * 4 + gg) * 16 + 4 * (lane >> 4) + e) * 17 + (lane & 15)] = acc[gg][e];
    PT(1);
    persist_lds_barrier();
    PT(2);
    if (act) {
      auto pre = [&](int gg) {
        return (Pp[((0 * 4 + gg) * 16 + r) * 17 + u] + Pp[((1 * 4 + gg) * 16 + r) * 17 + u]) +
               (Pp[((2 * 4 + gg) * 16 + r) * 17 + u] + Pp[((3 * 4 + gg) * 16 + r) * 17 + u]);
      };
      if (G == 4) {
        const float ig = sigmoid_acc(pre(0) + g0), fg = sigmoid_acc(pre(1) + g1), gg_ = tanh_cell(pre(2) + g2),
                    og = sigmoid_acc(pre(3) + g3);
        c = fg * c + ig * gg_;
        h = og * tanh_cell(c);
        a.y[row * ldh + (size_t)dir * H + j] = h;
        if (a.gates) {
          reinterpret_cast<float4*>(a.gates)[(row * a.ndir + dir) * H + j] = make_float4(ig, fg, gg_, og);
          a.csave[row * ldh + (size_t)dir * H + j] = c;
        }
      } else {
        // torch.nn.GRU: r, z from both projections, n = tanh(gin_n + r (W_hn h + b_hn)); saved for
        // backward as (r, z, n, W_hn h + b_hn)
        const float rg = sigmoid_acc(g0 + pre(0) + bh0), zg = sigmoid_acc(g1 + pre(1) + bh1);
        const float hnp = pre(2) + bh2;
        const float ng = tanh_cell(g2 + rg * hnp);
        h = (1.f - zg) * ng + zg * h;
        a.y[row * ldh + (size_t)dir * H + j] = h;
        if (a.gates) reinterpret_cast<float4*>(a.gates)[(row * a.ndir + dir) * H + j] = make_float4(rg, zg, ng, hnp);
      }
      if (s == len - 1) {
        if (a.hn) a.hn[((size_t)dir * a.B + b) * H + j] = h;
        if (G == 4 && a.cn) a.cn[((size_t)dir * a.B + b) * H + j] = c;
      }
    }
    row_cur = row_n1;
    gc0 = gn0; gc1 = gn1; gc2 = gn2; gc3 = gn3;
    PT(3);
    if (s + 1 < t_tile) publish(s + 1, h);
    PT(4);
  }
  if (PERSIST_TRACE && lane == 0 && (blockIdx.x < 8 || blockIdx.x == 100) )
    printf("block %3d wave %d: steps %d  poll %.2f  mfma %.2f  barrier %.2f  cell %.2f  barrier+publish %.2f us per step\n", (int)blockIdx.x, wv, t_tile,
           tacc[0] / 100.0 / t_tile, tacc[1] / 100.0 / t_tile, tacc[2] / 100.0 / t_tile, tacc[3] / 100.0 / t_tile, tacc[4] / 100.0 / t_tile);
#undef PT
}


// Runs the forward recurrence with the persistent kernel where it applies.  Returns 1 when it did
// (results complete), 0 when the caller has to run the step kernels (not applicable, switched off,
// or a launch that gave up waiting), -1 on a HIP error.  It needs all 256 workgroups resident at
// once, which nobody can promise (another process may hold CUs): every wait carries a budget, the
// launch is followed by a read-back of the abort flag, and a launch that gave up switches the
// persistent path off for the rest of the process.
// Read-back of a launch's abort flag: the copy is queued behind the kernel, and the host spins on the
// page-locked word it lands in instead of blocking in hipStreamSynchronize (which wakes up tens of
// microseconds after the event: six such waits per training step); 0.2 s without the word changing
// and it blocks after all.  Returns the flag, or -1 on a runtime error.
static int persist_read_flag(int64_t* slot, const int* d_flag, hipStream_t s) {
  volatile int* w = reinterpret_cast<volatile int*>(slot);
  constexpr int kPending = 0x7fffffff;
  w[0] = kPending;
  if (hipMemcpyAsync(slot, d_flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; w[0] == kPending; ++spins) {
    __builtin_ia32_pause();
    if ((spins & 0xffff) == 0xffff &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.2) {
      if (hipStreamSynchronize(s) != hipSuccess) return -1;
      break;
    }
  }
  return w[0] == kPending ? -1 : w[0];
}

// Per-device bookkeeping of a persistent kernel: CU count, the dynamic-LDS attribute (set once per
// device, not once per process), and a cool-down -- a launch that gave up waiting (CUs held by another
// stream or process) sends the next calls to the step kernels and is then tried again: 64 calls after
// the first give-up, twice as many after every further one in a row (cap 1 << 16: a device whose CUs
// are masked or shared for good costs one 50-ms polling budget every 65 536 calls, not every 65th),
// back to 64 after a launch that ran.  The message is printed for the first three give-ups in a row only.
constexpr int kPersistCooldown = 64, kPersistCooldownCap = 1 << 16;
struct PersistDevice {
  int n_cu = 0;
  bool attr_set = false;
  int cooldown = 0;
  int next_cooldown = kPersistCooldown;
  int give_ups = 0;       // in a row
};
// Returns 1 when the persistent kernel may be launched on the current device, 0 when the caller has to
// take the step kernels (fewer than 256 CUs, cooling down, or the attribute could not be set).
static int persist_device_ready(const void* kernel, int lds_bytes, std::map<int, PersistDevice>& table, std::mutex& mu,
                                int* dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  *dev_out = dev;
  std::lock_guard<std::mutex> lock(mu);
  PersistDevice& d = table[dev];
  if (d.n_cu == 0) {
    if (hipDeviceGetAttribute(&d.n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { d.n_cu = 0; return 0; }
  }
  if (d.n_cu != 256) return 0;
  if (d.cooldown > 0) { --d.cooldown; return 0; }
  if (!d.attr_set) {
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    d.attr_set = true;
  }
  return 1;
}
// A launch gave up (or could not be made): returns the number of calls the step kernels take now, negative when the
// message for it should not be printed any more.
static int persist_cool_down(std::map<int, PersistDevice>& table, std::mutex& mu, int dev) {
  std::lock_guard<std::mutex> lock(mu);
  PersistDevice& d = table[dev];
  d.cooldown = d.next_cooldown;
  d.next_cooldown = std::min(2 * d.next_cooldown, kPersistCooldownCap);
  ++d.give_ups;
  return d.give_ups <= 3 ? d.cooldown : -d.cooldown;
}
static void persist_ran(std::map<int, PersistDevice>& table, std::mutex& mu, int dev) {
  std::lock_guard<std::mutex> lock(mu);
  PersistDevice& d = table[dev];
  d.next_cooldown = kPersistCooldown;
  d.give_ups = 0;
}

template <int G>
static int rnn_persist_forward(RnnPersistArgs p, int H, hipStream_t s) {
  static std::map<int, PersistDevice> devices;
  static std::mutex devices_mu;
  const char* pe = getenv("ITTS_RNN_PERSISTENT");       // read per call: tests switch it
  if (pe && pe[0] == '0') return 0;
  p.ntiles = (p.B + 15) / 16;
  if (H != PH) return 0;
  int dev = 0;
  if (!persist_device_ready(reinterpret_cast<const void*>(&rnn_persist_fwd_kernel<G>), persist_lds_bytes(G), devices,
                            devices_mu, &dev))
    return 0;
  DeviceContext* ctx = get_context();
  if (!ctx) return 0;
  const size_t xbytes = (size_t)8 * 4 * 32 * 64 * sizeof(uint4);
  char* blk = nullptr;
  itts::ScratchScope scope(s);
  if (itts::scratch_malloc((void**)&blk, xbytes + 64, s) != hipSuccess) return -1;
  if (hipMemsetAsync(blk, 0, xbytes + 64, s) != hipSuccess) return -1;
  p.xchg = reinterpret_cast<uint4*>(blk);
  p.abort_flag = reinterpret_cast<int*>(blk + xbytes);
  if (getenv("ITTS_RNN_PERSIST_TEST_ABORT"))       // test hook: the launch finds the abort flag raised
    if (hipMemsetAsync(p.abort_flag, 1, sizeof(int), s) != hipSuccess) return -1;
  // 8 / ndir batch tiles per launch (one XCD each); a larger batch takes its tiles in rounds, longest
  // rows first (the exchange slots of a round are all overwritten before anybody of the next round
  // can mistake them: its first wait is for step 0's mask, which no later step of the round before
  // left behind -- masks are unique per step -- unless that round was one step long; the buffer is
  // therefore cleared between rounds)
  for (p.tile0 = 0; p.tile0 < p.ntiles; p.tile0 += 8 / p.ndir) {
    if (p.tile0 > 0 && hipMemsetAsync(blk, 0, xbytes, s) != hipSuccess) return -1;
    hipLaunchKernelGGL(rnn_persist_fwd_kernel<G>, dim3(256), dim3(256), persist_lds_bytes(G), s, p);
    if (hipGetLastError() != hipSuccess) {          // nothing of this round ran: the step kernels redo the layer
      persist_cool_down(devices, devices_mu, dev);
      return 0;
    }
  }
  const int gave_up = persist_read_flag(pinned_slot(ctx), p.abort_flag, s);
  if (gave_up < 0) return -1;
  if (itts::scratch_free(blk, s) != hipSuccess) return -1;
  if (gave_up == 0) { persist_ran(devices, devices_mu, dev); return 1; }
  const int calls = persist_cool_down(devices, devices_mu, dev);
  if (calls > 0)
    fprintf(stderr, "libidiaptts_amd: the persistent recurrence gave up waiting (are all 256 CUs available to "
                    "this process?); per-step kernels for the next %d calls\n", calls);
  return 0;
}


// ---- persistent backward recurrence, same placement (G = 4: LSTM, G = 3: GRU) ---------------------
// Step s (T_tile - 1 ... 0) of row b:  dh = dy + dG(s + 1) W_hh (+ the GRU's dh z carry),  then the
// cell gradients (lstm.hip, lstm_step_bwd_kernel; gru.hip, gru_step_bwd_kernel).  dG is G gates
// wide, so handing it around as the forward kernel hands h around would multiply the exchange by G;
// instead the product is split along K: workgroup c multiplies ITS OWN 16 G gate values per row (kept
// in LDS, never exchanged) with its 16 G rows of W_hh (a G x 32 KB LDS image: [column tile 32][gate]
// [lane] float4) into a PARTIAL dh for all 512 units, publishes the 16 x 16 tile of every unit block
// to the workgroup that owns it (reduce-scatter: 32 KB out, 32 KB in, per CU and step), and sums the
// 32 partial tiles it receives.  Polling, budget and fallback as above, but the granules carry their
// tag INSIDE: the lowest mantissa bit of each of a granule's four values is that value's validity bit
// (persist_use_tag: complementary between consecutive uses of a slot), so a tile's 256 values are 64
// granules and the two step slots of an XCD's 32 x 32 tiles are 2 MB of its 4 MB L2.  History
// (profiles/r3h_rnn_traffic.txt, r4a_rnn_bwd_variants.txt; one layer, 64 rows, T = 1981, against
// 3 GB of gate and gradient data): value + check-copy pairs, 4 MB per XCD: 22.8 GB written and 14 GB
// read per launch -- every step's exchange went through HBM; three values + a tag word per granule,
// 2.75 MB: reads fit (1.8 GB) but 15.8 GB were still written back -- the slots and the streamed
// gates / gradients together overflow the L2's ways; at 2 MB the write-back drops to 4.0 GB.  The
// price is the last bit of each PARTIAL sum (<= 1 ulp of a 64-term fp32 dot product that already
// carries several ulp of rounding; 32 such partials are added per element): the layer's gradients
// stay inside the budget of tests/test_gpu_rnn_long.py unchanged.
struct RnnPersistBwdArgs {
  const float* dy;
  const float* whh;
  const float* c0;      // LSTM: initial cell state [ndir][H] or NULL
  const float* gates;   // LSTM (i, f, g, o), GRU (r, z, n, hn_pre) per row, direction, unit
  const float* aux;     // LSTM: the saved cell states; GRU: h_prev per packed row
  const int* lengths;
  const int* row_off;
  const int* rev_row;
  float* dg;            // LSTM: dG; GRU: dGi
  float* dg2;           // GRU: dGh (da_n * r in the third gate)
  float* d0;            // [ndir][B][H] or NULL: LSTM dc * f, GRU dh * z after step 0
  uint4* xchg;          // [8 groups][2 slots][32 consumers][32 producers][PT granules]
  int* abort_flag;
  int T, B, ndir, ntiles, tile0;
};

constexpr int PT = 64;    // granules per partial tile: one per lane, four values each, tag in the four low mantissa bits
constexpr int persist_bwd_lds_bytes(int G) { return 32 * G * 64 * 16 + 4 * 16 * 17 * 4 + G * 16 * 16 * 4; }

#ifndef PERSIST_BWD_TRACE
#define PERSIST_BWD_TRACE 0
#endif
template <int G>
__global__ __launch_bounds__(256) void rnn_persist_bwd_kernel(RnnPersistBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char psm[];
  float4* Wb = reinterpret_cast<float4*>(psm);                               // [ct 32][gate G][lane 64]
  float* Pp = reinterpret_cast<float*>(psm + 32 * G * 64 * 16);              // [wave 4][row 16][17]
  float* dgs = Pp + 4 * 16 * 17;                                             // [gate G][row 16][unit 16]
  const int group = blockIdx.x & 7, cu = blockIdx.x >> 3;
  const int tiles_per_dir = 8 / a.ndir;
  const int dir = group / tiles_per_dir, tile = a.tile0 + group % tiles_per_dir;
  if (tile >= a.ntiles) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int H = PH, GH = G * PH;
  const int j0 = cu * 16;
  {
    // image of this workgroup's 16 G rows of W_hh: lane (col, kq) of column tile ct, gate g holds
    // W[g H + j0 + 4 i + kq][16 ct + col], i = 0 .. 3
    const float* W = a.whh + (size_t)dir * GH * H;
    for (int idx = threadIdx.x; idx < 32 * G * 64; idx += 256) {
      const int ln = idx & 63, gg = (idx >> 6) % G, ct = (idx >> 6) / G;
      const float* src = W + (size_t)(gg * H + j0 + (ln >> 4)) * H + 16 * ct + (ln & 15);
      Wb[idx] = make_float4(src[0], src[4 * (size_t)H], src[8 * (size_t)H], src[12 * (size_t)H]);
    }
  }
  const int r = threadIdx.x >> 4, u = threadIdx.x & 15;
  const int b = tile * 16 + r, j = j0 + u;
  const bool valid = b < a.B;
  const int len = valid ? a.lengths[b] : 0;
  const int t_tile = a.lengths[tile * 16];
  const size_t ldg = (size_t)a.ndir * GH, ldh = (size_t)a.ndir * H;
  uint4* xg = a.xchg + (size_t)group * 2 * 32 * 32 * PT;
  auto row_at = [&](int st) -> size_t {
    return dir == 0 ? (size_t)(a.row_off[st] + b) : (size_t)a.rev_row[(size_t)st * a.B + b];
  };
  const float c_init = (G == 4 && valid && a.c0) ? a.c0[dir * H + j] : 0.f;
  // Two sets of pipeline registers used alternately (no copy of a value still in flight: a copy would
  // wait for the load): the current step's values in one, the next step's requested into the other.
  // LSTM: v1 = c_t, v2 = c_{t-1} (the saved cell state of the step before); GRU: v1 = h_prev
  struct Vals { float4 g; float dy, v1, v2; size_t row; };
  const int s0 = t_tile - 1;
  Vals va{make_float4(0.f, 0.f, 0.f, 0.f), 0.f, 0.f, 0.f, 0}, vb = va;
  // the row index of a step is requested three steps ahead as ONE load of a raw int (forward
  // direction: row_off[step], + b when it is used; reverse: rev_row[step][b]) -- anything computed from
  // it at once would wait for it at once
  const int* rtab = dir == 0 ? a.row_off : a.rev_row + b;
  const size_t rstride = dir == 0 ? 1 : (size_t)a.B;
  const int radd = dir == 0 ? b : 0;
  int rnew = 0;
  if (s0 < len) {
    va.row = row_at(s0);
    va.g = reinterpret_cast<const float4*>(a.gates)[(va.row * a.ndir + dir) * H + j];
    va.dy = a.dy[va.row * ldh + (size_t)dir * H + j];
    va.v1 = a.aux[va.row * ldh + (size_t)dir * H + j];
  }
  if (s0 - 1 >= 0 && s0 - 1 < len) vb.row = row_at(s0 - 1);
  if (s0 - 2 >= 0 && s0 - 2 < len) rnew = rtab[(size_t)(s0 - 2) * rstride];
  if (G == 4 && s0 < len) va.v2 = s0 > 0 ? a.aux[vb.row * ldh + (size_t)dir * H + j] : c_init;
  float carry = 0.f;       // LSTM: dc * f, GRU: dh * z of the step processed before (s + 1)
  __syncthreads();         // the W image is complete
  unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = 0;
#define PTB(i) do { if (PERSIST_BWD_TRACE) { const unsigned long long tn = wall_clock64(); tacc[i] += tn - tprev; tprev = tn; } } while (0)
  if (PERSIST_BWD_TRACE) tprev = wall_clock64();

  auto step = [&](const int s, Vals& X, Vals& Y) __attribute__((always_inline)) {
    const bool act = s < len;
    // dh_rec: the 32 partial tiles published at step s + 1
    float dhr = 0.f;
    if (s < s0) {
      pu32x4 pv[8];
      const uint4* src = xg + (((size_t)((s + 1) & 1) * 32 + cu) * 32 + 8 * wv) * PT;
      const unsigned m = persist_use_tag((s0 - (s + 1)) >> 1);
      int budget = 1 << 16;
      for (;;) {
        persist_load8(src + lane, PT, pv);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) ok = ok && persist_tag_of(pv[i]) == m;
        if (__all(ok)) break;
        const uint4* watch = src + (lane & 7) * PT;
        bool gave_up = false;
        for (;;) {
          pu32x4 wp;
          persist_load_one(watch, wp);
          if (__all(persist_tag_of(wp) == m)) break;
          if (--budget <= 0 || *reinterpret_cast<volatile int*>(a.abort_flag)) { gave_up = true; break; }
          __builtin_amdgcn_s_sleep(4);
        }
        if (gave_up || --budget <= 0) {
          if (lane == 0) atomicExch(a.abort_flag, 1);
          break;
        }
      }
      // this wave's eight tiles summed (rows 4 kq + e, column lane & 15), then across the waves
      float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sx += __uint_as_float(pv[i].x); sy += __uint_as_float(pv[i].y); sz += __uint_as_float(pv[i].z);
        sw += __uint_as_float(pv[i].w);
      }
      const int rq = 4 * (lane >> 4), cc = lane & 15;
      Pp[(wv * 16 + rq + 0) * 17 + cc] = sx; Pp[(wv * 16 + rq + 1) * 17 + cc] = sy;
      Pp[(wv * 16 + rq + 2) * 17 + cc] = sz; Pp[(wv * 16 + rq + 3) * 17 + cc] = sw;
    }
    // row of step s - 2, requested a step ago; pinned HERE, where the poll has just waited for everything
    // in flight (the compiler does not know that, and would otherwise wait at the first use below,
    // behind this step's stores)
    int rraw = rnew;
    asm volatile("" : "+v"(rraw));
    const size_t r2 = (size_t)(rraw + radd);
    PTB(0);
    persist_lds_barrier();
    PTB(1);
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    if (act) {
      if (s < s0 && s + 1 < len)
        dhr = (Pp[(0 * 16 + r) * 17 + u] + Pp[(1 * 16 + r) * 17 + u]) + (Pp[(2 * 16 + r) * 17 + u] + Pp[(3 * 16 + r) * 17 + u]);
      if (G == 4) {
        const float ig = X.g.x, fg = X.g.y, gg = X.g.z, og = X.g.w;
        const float tc = tanh_cell(X.v1);
        const float dh = X.dy + dhr;
        const float dcv = dh * og * (1.f - tc * tc) + carry;
        d[0] = dcv * gg * ig * (1.f - ig); d[1] = dcv * X.v2 * fg * (1.f - fg);
        d[2] = dcv * ig * (1.f - gg * gg); d[3] = dh * tc * og * (1.f - og);
        float* dgo = a.dg + X.row * ldg + (size_t)dir * GH + j;
        __builtin_nontemporal_store(d[0], dgo); __builtin_nontemporal_store(d[1], dgo + H);
        __builtin_nontemporal_store(d[2], dgo + 2 * H); __builtin_nontemporal_store(d[3], dgo + 3 * H);
        carry = dcv * fg;
      } else {
        const float rg = X.g.x, zg = X.g.y, ng = X.g.z, hnp = X.g.w;
        const float dh = X.dy + dhr + carry;
        const float dn = dh * (1.f - zg);
        const float dz = dh * (X.v1 - ng);
        const float dan = dn * (1.f - ng * ng);
        const float dar = dan * hnp * rg * (1.f - rg);
        const float daz = dz * zg * (1.f - zg);
        float* gi = a.dg + X.row * ldg + (size_t)dir * GH + j;
        float* gh = a.dg2 + X.row * ldg + (size_t)dir * GH + j;
        __builtin_nontemporal_store(dar, gi); __builtin_nontemporal_store(daz, gi + H); __builtin_nontemporal_store(dan, gi + 2 * H);
        d[0] = dar; d[1] = daz; d[2] = dan * rg;
        __builtin_nontemporal_store(d[0], gh); __builtin_nontemporal_store(d[1], gh + H); __builtin_nontemporal_store(d[2], gh + 2 * H);
        carry = dh * zg;
      }
    } else {
      carry = 0.f;      // (only ever read once the row is active; rows are active from their last frame down)
    }
    if (s == 0 && valid && a.d0) a.d0[((size_t)dir * a.B + b) * H + j] = carry;
#pragma unroll
    for (int gg2 = 0; gg2 < G; ++gg2) dgs[(gg2 * 16 + r) * 16 + u] = d[gg2];
    persist_lds_barrier();
    PTB(2);
    // values of step s - 1 (their row indices arrived a step ago) and the row index of step s - 3:
    // requested here, in front of the products, so that neither a barrier nor the next poll waits for them
    Y.g = make_float4(0.f, 0.f, 0.f, 0.f);
    Y.dy = 0.f; Y.v2 = 0.f;
    Y.v1 = X.v2;                   // LSTM: c_t of step s - 1 is c_{t-1} of step s ...
    if (s - 1 >= 0 && s - 1 < len) {
      Y.g = persist_stream_load4(reinterpret_cast<const float4*>(a.gates) + (Y.row * a.ndir + dir) * H + j);
      Y.dy = __builtin_nontemporal_load(a.dy + Y.row * ldh + (size_t)dir * H + j);
      if (G == 4) {
        Y.v2 = s - 1 > 0 ? __builtin_nontemporal_load(a.aux + r2 * ldh + (size_t)dir * H + j) : c_init;
        if (!act) Y.v1 = __builtin_nontemporal_load(a.aux + Y.row * ldh + (size_t)dir * H + j);   // ... unless the row only starts there
      } else {
        Y.v1 = __builtin_nontemporal_load(a.aux + Y.row * ldh + (size_t)dir * H + j);
      }
    }
    X.row = r2;                    // X is the current set again at step s - 2
    if (s - 3 >= 0 && s - 3 < len) rnew = rtab[(size_t)(s - 3) * rstride];
    if (s > 0) {
      // partial dh of step s - 1: [16 rows x 16 G own gate values] x [16 G x 512], this wave's 8 column tiles
      const int rr = lane & 15, kq = lane >> 4;
      float4 af[G];
#pragma unroll
      for (int gg2 = 0; gg2 < G; ++gg2)
        af[gg2] = make_float4(dgs[(gg2 * 16 + rr) * 16 + kq], dgs[(gg2 * 16 + rr) * 16 + 4 + kq],
                              dgs[(gg2 * 16 + rr) * 16 + 8 + kq], dgs[(gg2 * 16 + rr) * 16 + 12 + kq]);
      const unsigned m = persist_use_tag((s0 - s) >> 1);
      const unsigned t0 = m & 1u, t1 = (m >> 1) & 1u, t2 = (m >> 2) & 1u, t3 = (m >> 3) & 1u;
      // Software pipeline over the tiles: the values of tile i - 1 are tagged and stored BETWEEN the
      // products of tile i (a wave issues in order: behind the last product of a chain the read of its
      // result and the store would otherwise leave the matrix unit idle)
      f32x4 accp = {0.f, 0.f, 0.f, 0.f};
      const float4* wsrc = Wb + (size_t)(8 * wv * G) * 64 + lane;      // this wave's 8 G operand quadruples
      float4 bw = wsrc[0];
#pragma unroll
      for (int i = 0; i <= 8; ++i) {
        const int ct = 8 * wv + i;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        uint4* dst = xg + (((size_t)(s & 1) * 32 + (ct - 1)) * 32 + cu) * PT;
#pragma unroll
        for (int gg2 = 0; gg2 < G; ++gg2) {
          float4 bwn = bw;
          if (i * G + gg2 + 1 < 8 * G) bwn = wsrc[(i * G + gg2 + 1) * 64];      // requested one group ahead
          if (i < 8) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[gg2].x, bw.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[gg2].y, bw.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[gg2].z, bw.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[gg2].w, bw.w, acc, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (i > 0 && gg2 == 0)
            dst[lane] = make_uint4((__float_as_uint(accp[0]) & ~1u) | t0, (__float_as_uint(accp[1]) & ~1u) | t1,
                                   (__float_as_uint(accp[2]) & ~1u) | t2, (__float_as_uint(accp[3]) & ~1u) | t3);
          __builtin_amdgcn_sched_barrier(0);
          bw = bwn;
        }
        accp = acc;
      }
    }
    PTB(3);
  };
  for (int s = s0; s >= 0; s -= 2) {
    step(s, va, vb);
    if (s >= 1) step(s - 1, vb, va);
  }
  if (PERSIST_BWD_TRACE && lane == 0 && (blockIdx.x < 8 || blockIdx.x == 100))
    printf("bwd block %3d wave %d: steps %d  poll %.2f  loads+barrier %.2f  cell+barrier %.2f  mfma+publish %.2f us per step\n",
           (int)blockIdx.x, wv, t_tile, tacc[0] / 100.0 / t_tile, tacc[1] / 100.0 / t_tile, tacc[2] / 100.0 / t_tile,
           tacc[3] / 100.0 / t_tile);
#undef PTB
}

// Backward counterpart of rnn_persist_forward: 1 = done, 0 = run the step kernels.
template <int G>
static int rnn_persist_backward(RnnPersistBwdArgs p, const int* h_lengths, int H, hipStream_t s) {
  static std::map<int, PersistDevice> devices;
  static std::mutex devices_mu;
  const char* pe = getenv("ITTS_RNN_PERSISTENT");          // read per call: tests switch it
  const char* pb = getenv("ITTS_RNN_PERSISTENT_BWD");      // ... and this one keeps the forward half on
  if ((pe && pe[0] == '0') || (pb && pb[0] == '0')) return 0;
  p.ntiles = (p.B + 15) / 16;
  if (H != PH) return 0;
  int dev = 0;
  if (!persist_device_ready(reinterpret_cast<const void*>(&rnn_persist_bwd_kernel<G>), persist_bwd_lds_bytes(G),
                            devices, devices_mu, &dev))
    return 0;
  DeviceContext* ctx = get_context();
  if (!ctx) return 0;
  const size_t xbytes = (size_t)8 * 2 * 32 * 32 * PT * sizeof(uint4);
  const size_t lbytes = ((size_t)p.B * sizeof(int) + 63) / 64 * 64;
  char* blk = nullptr;
  itts::ScratchScope scope(s);
  if (itts::scratch_malloc((void**)&blk, xbytes + 64 + lbytes, s) != hipSuccess) return -1;
  if (hipMemsetAsync(blk, 0, xbytes + 64, s) != hipSuccess) return -1;
  if (staged_upload(blk + xbytes + 64, h_lengths, (size_t)p.B * sizeof(int), s) != ITTS_OK) return -1;
  p.xchg = reinterpret_cast<uint4*>(blk);
  p.abort_flag = reinterpret_cast<int*>(blk + xbytes);
  p.lengths = reinterpret_cast<const int*>(blk + xbytes + 64);
  if (getenv("ITTS_RNN_PERSIST_TEST_ABORT"))       // test hook: the launch finds the abort flag raised
    if (hipMemsetAsync(p.abort_flag, 1, sizeof(int), s) != hipSuccess) return -1;
  for (p.tile0 = 0; p.tile0 < p.ntiles; p.tile0 += 8 / p.ndir) {
    if (p.tile0 > 0 && hipMemsetAsync(blk, 0, xbytes, s) != hipSuccess) return -1;
    hipLaunchKernelGGL(rnn_persist_bwd_kernel<G>, dim3(256), dim3(256), persist_bwd_lds_bytes(G), s, p);
    if (hipGetLastError() != hipSuccess) {
      persist_cool_down(devices, devices_mu, dev);
      return 0;
    }
  }
  const int gave_up = persist_read_flag(pinned_slot(ctx), p.abort_flag, s);
  if (gave_up < 0) return -1;
  if (itts::scratch_free(blk, s) != hipSuccess) return -1;
  if (gave_up == 0) { persist_ran(devices, devices_mu, dev); return 1; }
  const int calls = persist_cool_down(devices, devices_mu, dev);
  if (calls > 0)
    fprintf(stderr, "libidiaptts_amd: the persistent backward recurrence gave up waiting; per-step kernels for the "
                    "next %d calls\n", calls);
  return 0;
}

}  // namespace itts
