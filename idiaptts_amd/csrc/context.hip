// Device context: twiddle table and SPTK frequency-warping matrices (see context.h).
#include <cstdlib>

#include "context.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "world_dev.h"

namespace itts {

static std::mutex g_ctx_mutex;
static std::map<int, DeviceContext*> g_contexts;

static int create_context(DeviceContext* ctx) {
  const int n = wd::TW_N / 2;
  std::vector<double2> tw(n);
  for (int k = 0; k < n; ++k) {
    const long double a = 2.0L * 3.141592653589793238462643383279502884L * k / wd::TW_N;
    tw[k] = make_double2((double)cosl(a), (double)sinl(a));
  }
  ITTS_HIP_CHECK(hipMalloc((void**)&ctx->twiddles, n * sizeof(double2)));
  ITTS_HIP_CHECK(hipMemcpy(ctx->twiddles, tw.data(), n * sizeof(double2), hipMemcpyHostToDevice));
  // The entry points take their scratch from the device's stream-ordered pool.  By default the
  // pool hands everything back to the driver at the next synchronisation, which turns every call
  // into fresh multi-hundred-MB allocations; let it keep up to 64 GB (of 288) between calls.
  for (int L = 5; (1 << L) <= wd::TW_N; ++L) {
    const int m = (1 << L) / 2, stride = wd::TW_N / (1 << L);
    std::vector<double2> c(m);
    for (int k = 0; k < m; ++k) c[k] = tw[(size_t)k * stride];
    ITTS_HIP_CHECK(hipMalloc((void**)&ctx->tw_compact[L], m * sizeof(double2)));
    ITTS_HIP_CHECK(hipMemcpy(ctx->tw_compact[L], c.data(), m * sizeof(double2), hipMemcpyHostToDevice));
  }
  ITTS_HIP_CHECK(hipHostMalloc((void**)&ctx->pinned, 64 * sizeof(int64_t), hipHostMallocDefault));
  return ITTS_OK;
}


}  // namespace itts


// ---- scratch blocks ---------------------------------------------------------------------------------
namespace itts {

struct ScratchBlock {
  void* p;
  size_t bytes;
  hipEvent_t ev;       // recorded on `last` when the block was handed back
  hipStream_t last;
  bool busy;
};
struct ScratchCache {
  std::vector<ScratchBlock> blocks;
  size_t total = 0;
};
static std::mutex g_scratch_mutex;
static std::map<int, ScratchCache> g_scratch;

static size_t scratch_class(size_t n) {
  if (n <= ((size_t)1 << 20)) {          // powers of two from 256 B: exact-class reuse
    size_t c = 256;
    while (c < n) c <<= 1;
    return c;
  }
  const size_t g = (size_t)2 << 20;      // multiples of 2 MB above
  return (n + g - 1) / g * g;
}

static size_t scratch_keep_bytes() {
  static const size_t keep = [] {
    size_t gb = 64;
    if (const char* e = getenv("ITTS_POOL_KEEP_GB")) gb = strtoull(e, nullptr, 10);
    return gb << 30;
  }();
  return keep;
}

// frees idle blocks (largest first) until at most `limit` bytes are held; the caller holds the mutex
static void scratch_evict(ScratchCache& c, size_t limit) {
  while (c.total > limit) {
    int pick = -1;
    for (int i = 0; i < (int)c.blocks.size(); ++i)
      if (!c.blocks[i].busy && (pick < 0 || c.blocks[i].bytes > c.blocks[pick].bytes)) pick = i;
    if (pick < 0) return;
    ScratchBlock b = c.blocks[pick];
    (void)hipEventSynchronize(b.ev);
    (void)hipFree(b.p);
    (void)hipEventDestroy(b.ev);
    c.total -= b.bytes;
    c.blocks.erase(c.blocks.begin() + pick);
  }
}

static thread_local ScratchScope* t_scope = nullptr;

ScratchScope::ScratchScope(hipStream_t s) : stream_(s), prev_(t_scope) { t_scope = this; }
ScratchScope::~ScratchScope() {
  t_scope = prev_;   // first: the frees below must not come back to this scope
  for (int i = 0; i < n_live_; ++i)
    if (live_[i]) (void)scratch_free(live_[i], stream_);
}
void ScratchScope::track(void* p) {
  if (n_live_ < 64) live_[n_live_++] = p;   // beyond 64 live blocks: as before (explicit frees only)
}
void ScratchScope::untrack(void* p) {
  for (int i = n_live_ - 1; i >= 0; --i)
    if (live_[i] == p) {
      live_[i] = nullptr;
      while (n_live_ > 0 && live_[n_live_ - 1] == nullptr) --n_live_;
      return;
    }
  if (prev_) prev_->untrack(p);   // obtained under an outer scope
}

static hipError_t scratch_malloc_impl(void** out, size_t bytes, hipStream_t s);
hipError_t scratch_malloc(void** out, size_t bytes, hipStream_t s) {
  const hipError_t e = scratch_malloc_impl(out, bytes, s);
  if (e == hipSuccess && *out && t_scope) t_scope->track(*out);
  return e;
}

static hipError_t scratch_malloc_impl(void** out, size_t bytes, hipStream_t s) {
  *out = nullptr;
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const size_t want = scratch_class(bytes);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  ScratchCache& c = g_scratch[dev];
  int best = -1;
  for (int i = 0; i < (int)c.blocks.size(); ++i) {
    const ScratchBlock& b = c.blocks[i];
    if (b.busy || b.bytes < want) continue;
    if (want <= ((size_t)1 << 20) ? b.bytes != want : b.bytes > want + want / 4 + ((size_t)4 << 20)) continue;
    if (best < 0 || b.bytes < c.blocks[best].bytes) best = i;
  }
  if (best >= 0) {
    ScratchBlock& b = c.blocks[best];
    if (b.last != s && (e = hipStreamWaitEvent(s, b.ev, 0)) != hipSuccess) return e;
    b.busy = true;
    *out = b.p;
    return hipSuccess;
  }
  void* p = nullptr;
  e = hipMalloc(&p, want);
  if (e != hipSuccess) {               // out of memory: give back what is idle and try once more
    (void)hipGetLastError();
    scratch_evict(c, 0);
    if ((e = hipMalloc(&p, want)) != hipSuccess) return e;
  }
  hipEvent_t ev;
  if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) {
    (void)hipFree(p);
    return e;
  }
  c.blocks.push_back(ScratchBlock{p, want, ev, s, true});
  c.total += want;
  if (c.total > scratch_keep_bytes()) scratch_evict(c, scratch_keep_bytes());
  *out = p;
  return hipSuccess;
}

hipError_t scratch_free(void* p, hipStream_t s) {
  if (!p) return hipSuccess;
  if (t_scope) t_scope->untrack(p);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  for (auto& kv : g_scratch)
    for (ScratchBlock& b : kv.second.blocks)
      if (b.p == p) {
        const hipError_t e = hipEventRecord(b.ev, s);
        b.last = s;
        b.busy = false;
        return e;
      }
  return hipErrorInvalidValue;
}

}  // namespace itts

extern "C" int itts_scratch_pool_stats(int64_t* reserved, int64_t* used, int64_t* keep_threshold) {
  int dev = -1;
  ITTS_HIP_CHECK(hipGetDevice(&dev));
  size_t r = 0, u = 0;
  {
    std::lock_guard<std::mutex> lock(itts::g_scratch_mutex);
    auto it = itts::g_scratch.find(dev);
    if (it != itts::g_scratch.end())
      for (const itts::ScratchBlock& b : it->second.blocks) {
        r += b.bytes;
        if (b.busy) u += b.bytes;
      }
  }
  if (reserved) *reserved = (int64_t)r;
  if (used) *used = (int64_t)u;
  if (keep_threshold) *keep_threshold = (int64_t)itts::scratch_keep_bytes();
  return ITTS_OK;
}

extern "C" int itts_release_scratch(void) {
  int dev = -1;
  ITTS_HIP_CHECK(hipGetDevice(&dev));
  ITTS_HIP_CHECK(hipDeviceSynchronize());
  {
    std::lock_guard<std::mutex> lock(itts::g_scratch_mutex);
    auto it = itts::g_scratch.find(dev);
    if (it != itts::g_scratch.end()) itts::scratch_evict(it->second, 0);
  }
  return ITTS_OK;
}

namespace itts {

int64_t* pinned_slot(DeviceContext* ctx) {
  static std::atomic<unsigned> next{0};
  return ctx->pinned + (next.fetch_add(1) & 63u);
}

DeviceContext* get_context() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error("hipGetDevice failed: no HIP device (there is no CPU fallback)");
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  auto it = g_contexts.find(dev);
  if (it != g_contexts.end()) return it->second;
  DeviceContext* ctx = new DeviceContext();
  ctx->device = dev;
  if (create_context(ctx) != ITTS_OK) {
    delete ctx;
    return nullptr;
  }
  g_contexts[dev] = ctx;
  return ctx;
}

// ---- WORLD randn(): GF(2) jump matrices of the xorshift128 step + the out-of-order generator ------
struct Mat128 {
  uint32_t col[128][4];
};
static void mat_apply(const Mat128& m, const uint32_t* v, uint32_t* out) {
  uint32_t r[4] = {0, 0, 0, 0};
  for (int j = 0; j < 128; ++j)
    if ((v[j >> 5] >> (j & 31)) & 1u)
      for (int c = 0; c < 4; ++c) r[c] ^= m.col[j][c];
  std::memcpy(out, r, sizeof(r));
}
static void mat_mul(const Mat128& a, const Mat128& b, Mat128* out) {  // out = a * b (apply b first)
  Mat128 r;
  for (int j = 0; j < 128; ++j) mat_apply(a, b.col[j], r.col[j]);
  *out = r;
}

// states[c] = B^c * seed for c < 2^RNG_TABLE_LOG2 (once per device)
__global__ __launch_bounds__(256) void rng_states_kernel(const JumpTable* __restrict__ jt) {
  const int chunk = blockIdx.x * 256 + threadIdx.x;
  uint32_t s[4] = {123456789u, 362436069u, 521288629u, 88675123u};
  for (int k = 0; k < RNG_TABLE_LOG2; ++k) {
    if ((chunk >> k) & 1) {
      uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
      for (int j = 0; j < 128; ++j) {
        if ((s[j >> 5] >> (j & 31)) & 1u) {
          r0 ^= jt->col[k][j][0];
          r1 ^= jt->col[k][j][1];
          r2 ^= jt->col[k][j][2];
          r3 ^= jt->col[k][j][3];
        }
      }
      s[0] = r0; s[1] = r1; s[2] = r2; s[3] = r3;
    }
  }
  jt->states[chunk] = make_uint4(s[0], s[1], s[2], s[3]);
}

const JumpTable* get_jump_table(DeviceContext* ctx) {
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  if (ctx->jump) return ctx->jump;
  Mat128 step;
  for (int j = 0; j < 128; ++j) {
    uint32_t s[4] = {0, 0, 0, 0};
    s[j >> 5] = 1u << (j & 31);
    uint32_t x = s[0], y = s[1], z = s[2], w = s[3];
    const uint32_t t = x ^ (x << 11);
    x = y; y = z; z = w;
    w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
    step.col[j][0] = x; step.col[j][1] = y; step.col[j][2] = z; step.col[j][3] = w;
  }
  // B = step^(12*RNG_CHUNK) by square-and-multiply
  Mat128 B, pw = step;
  bool have = false;
  for (int e = 12 * RNG_CHUNK; e > 0; e >>= 1) {
    if (e & 1) {
      if (!have) { B = pw; have = true; } else mat_mul(pw, B, &B);
    }
    mat_mul(pw, pw, &pw);
  }
  std::vector<JumpTable> jt(1);
  Mat128 cur = B;
  for (int k = 0; k < RNG_NJUMP; ++k) {
    std::memcpy(jt[0].col[k], cur.col, sizeof(cur.col));
    mat_mul(cur, cur, &cur);
  }
  JumpTable* d = nullptr;
  uint4* states = nullptr;
  const int n_states = 1 << RNG_TABLE_LOG2;
  if (hipMalloc((void**)&states, sizeof(uint4) * n_states) != hipSuccess ||
      hipMalloc((void**)&d, sizeof(JumpTable)) != hipSuccess) {
    set_error("could not create the RNG jump table");
    return nullptr;
  }
  jt[0].states = states;
  if (hipMemcpy(d, jt.data(), sizeof(JumpTable), hipMemcpyHostToDevice) != hipSuccess) {
    set_error("could not create the RNG jump table");
    return nullptr;
  }
  hipLaunchKernelGGL(rng_states_kernel, dim3(n_states / 256), dim3(256), 0, 0, d);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(0) != hipSuccess) {
    set_error("could not tabulate the RNG states");
    return nullptr;
  }
  ctx->jump = d;
  return d;
}

// The plain form (every lane stores its own chunk, 256 bytes from its neighbour's): ITTS_RANDN_DIRECT=1, kept as the
// other side of tests/test_gpu_world.py's bit-for-bit comparison.
__global__ __launch_bounds__(256) void randn_u32_direct_kernel(const int64_t* __restrict__ off,
                                                               const int64_t* __restrict__ len,
                                                               const JumpTable* __restrict__ jt,
                                                               uint32_t* __restrict__ R) {
  const int u = blockIdx.y;
  const int64_t chunk = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n0 = chunk * RNG_CHUNK;
  const int64_t n = len[u];
  if (n0 >= n) return;
  const uint4 st = rng_chunk_state(jt, chunk);
  uint32_t x = st.x, y = st.y, z = st.z, w = st.w;
  uint32_t* out = R + off[u] + n0;
  const int cnt = (int)(n - n0 < RNG_CHUNK ? n - n0 : RNG_CHUNK);
  for (int i = 0; i < cnt; ++i) {
    uint32_t tmp = 0;
    for (int q = 0; q < 12; ++q) {
      const uint32_t t = x ^ (x << 11);
      x = y; y = z; z = w;
      w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
      tmp += w >> 4;
    }
    out[i] = tmp;
  }
}

// One lane generates the RNG_CHUNK consecutive normals of its chunk, so a wave's stores would land 256 bytes
// apart; the values go through a padded LDS tile (half a chunk at a time) and leave as 128-byte runs.
__global__ __launch_bounds__(256) void randn_u32_kernel(const int64_t* __restrict__ off,
                                                        const int64_t* __restrict__ len,
                                                        const JumpTable* __restrict__ jt,
                                                        uint32_t* __restrict__ R) {
  constexpr int HALF = RNG_CHUNK / 2;
  __shared__ uint32_t tile[4][64 * (HALF + 1)];
  const int u = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t chunk0 = (int64_t)blockIdx.x * 256 + wv * 64;  // the wave's first chunk
  const int64_t n = len[u];
  if ((int64_t)blockIdx.x * 256 * RNG_CHUNK >= n) return;      // the whole block is past the end
  const int64_t chunk = chunk0 + lane;
  const bool live = chunk * RNG_CHUNK < n;
  uint32_t x = 0, y = 0, z = 0, w = 0;
  if (live) {
    const uint4 st = rng_chunk_state(jt, chunk);
    x = st.x; y = st.y; z = st.z; w = st.w;
  }
  uint32_t* t = tile[wv];
  uint32_t* out = R + off[u] + chunk0 * RNG_CHUNK;
  const int64_t rem = n - chunk0 * RNG_CHUNK;  // normals the wave owes (<= 0: none)
  for (int h = 0; h < 2; ++h) {
    if (live) {
#pragma unroll 4
      for (int i = 0; i < HALF; ++i) {
        uint32_t tmp = 0;
        for (int q = 0; q < 12; ++q) {
          const uint32_t s = x ^ (x << 11);
          x = y; y = z; z = w;
          w = (w ^ (w >> 19)) ^ (s ^ (s >> 8));
          tmp += w >> 4;
        }
        t[lane * (HALF + 1) + i] = tmp;
      }
    }
    __syncthreads();
    const int col = lane & (HALF - 1), sub = lane / HALF;
    for (int r = 0; r < 64; r += 64 / HALF) {
      const int row = r + sub;
      const int64_t idx = (int64_t)row * RNG_CHUNK + h * HALF + col;
      if (idx < rem) out[idx] = t[row * (HALF + 1) + col];
    }
    __syncthreads();
  }
}

int launch_randn_u32(DeviceContext* ctx, const int64_t* d_off, const int64_t* d_len, int n_utts,
                     int64_t max_len, uint32_t* d_R, hipStream_t s) {
  const JumpTable* jt = get_jump_table(ctx);
  if (!jt) return ITTS_E_HIP;
  ITTS_REQUIRE(max_len < ((int64_t)RNG_CHUNK << RNG_NJUMP), "utterance too long for the RNG jump table");
  if (n_utts == 0 || max_len <= 0) return ITTS_OK;
  const int64_t chunks = (max_len + RNG_CHUNK - 1) / RNG_CHUNK;
  const char* direct = std::getenv("ITTS_RANDN_DIRECT");
  const dim3 grid((unsigned)((chunks + 255) / 256), n_utts);
  if (direct && direct[0] == '1')
    hipLaunchKernelGGL(randn_u32_direct_kernel, grid, dim3(256), 0, s, d_off, d_len, jt, d_R);
  else
    hipLaunchKernelGGL(randn_u32_kernel, grid, dim3(256), 0, s, d_off, d_len, jt, d_R);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

// SPTK freqt applied to the unit vector e_idx of an (m1+1)-vector -> g[0..m2]
static void freqt_unit(int idx, int m2, double a, double* g, double* d) {
  const double b = 1.0 - a * a;
  std::memset(g, 0, sizeof(double) * (m2 + 1));
  std::memset(d, 0, sizeof(double) * (m2 + 1));
  // the recursion consumes c1[m1], ..., c1[0]; everything before c1[idx] leaves g = 0
  for (int i = -idx; i <= 0; ++i) {
    const double cin = (i == -idx) ? 1.0 : 0.0;
    d[0] = g[0];
    g[0] = cin + a * d[0];
    if (m2 >= 1) {
      d[1] = g[1];
      g[1] = b * d[0] + a * d[1];
    }
    for (int j = 2; j <= m2; ++j) {
      d[j] = g[j];
      g[j] = d[j - 1] + a * (d[j] - g[j - 1]);
    }
  }
}

static void frqtr_unit(int idx, int m2, double a, double* g, double* d) {
  std::memset(g, 0, sizeof(double) * (m2 + 1));
  std::memset(d, 0, sizeof(double) * (m2 + 1));
  for (int i = -idx; i <= 0; ++i) {
    const double cin = (i == -idx) ? 1.0 : 0.0;
    d[0] = g[0];
    g[0] = cin;
    for (int j = 1; j <= m2; ++j) {
      d[j] = g[j];
      g[j] = d[j - 1] + a * (d[j] - g[j - 1]);
    }
  }
}

// SPTK mgcep.c's b2c applied to the unit vector e_idx -> g[0..m2]
static void b2c_unit(int idx, int m2, double a, double* g, double* d) {
  const double k = 1.0 - a * a;
  std::memset(g, 0, sizeof(double) * (m2 + 1));
  std::memset(d, 0, sizeof(double) * (m2 + 1));
  for (int i = -idx; i <= 0; ++i) {
    const double cin = (i == -idx) ? 1.0 : 0.0;
    d[0] = g[0];
    g[0] = cin;
    if (m2 >= 1) {
      d[1] = g[1];
      g[1] = k * d[0] + a * d[1];
    }
    for (int j = 2; j <= m2; ++j) {
      d[j] = g[j];
      g[j] = d[j - 1] + a * (d[j] - g[j - 1]);
    }
  }
}

static int upload(const std::vector<double>& h, double** d) {
  ITTS_HIP_CHECK(hipMalloc((void**)d, h.size() * sizeof(double)));
  ITTS_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  return ITTS_OK;
}

const FreqtTables* get_freqt(DeviceContext* ctx, int m, int f2, double alpha, bool need_fwd_frq,
                             bool need_mgc, bool need_spec) {
  long long abits;
  std::memcpy(&abits, &alpha, sizeof(abits));
  const auto key = std::make_tuple(m, f2, abits);
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  FreqtTables& t = ctx->freqt[key];
  t.m = m;
  t.f2 = f2;
  t.alpha = alpha;
  const int m2 = 2 * m;
  std::vector<double> g(std::max(f2, m2) + 2), d(std::max(f2, m2) + 2);
  if (!t.invT) {
    std::vector<double> inv((size_t)(m + 1) * (f2 + 1));
    for (int j = 0; j <= m; ++j) {
      freqt_unit(j, f2, -alpha, g.data(), d.data());
      for (int i = 0; i <= f2; ++i) inv[(size_t)j * (f2 + 1) + i] = g[i];
    }
    if (upload(inv, &t.invT) != ITTS_OK) return nullptr;
  }
  if (need_fwd_frq && !t.fwdT) {
    std::vector<double> fwd((size_t)(f2 + 1) * (m + 1)), frq((size_t)(f2 + 1) * (m2 + 1));
    for (int i = 0; i <= f2; ++i) {
      freqt_unit(i, m, alpha, g.data(), d.data());
      for (int j = 0; j <= m; ++j) fwd[(size_t)i * (m + 1) + j] = g[j];
      frqtr_unit(i, m2, alpha, g.data(), d.data());
      for (int j = 0; j <= m2; ++j) frq[(size_t)i * (m2 + 1) + j] = g[j];
    }
    if (upload(fwd, &t.fwdT) != ITTS_OK) return nullptr;
    if (upload(frq, &t.frqT) != ITTS_OK) return nullptr;
    // crT = Ci^T . frq with Ci[n][k] = w_k cos(2 pi k n / 2 f2) / (2 f2), w_0 = w_f2 = 1, otherwise 2: the
    // inverse transform of a real spectrum, sample n <= f2.  Cosines from the reduced integer
    // argument, sums in long double.
    const int nfft = 2 * f2;
    std::vector<long double> cs(nfft);
    for (int i = 0; i < nfft; ++i) cs[i] = cosl(2.0L * 3.14159265358979323846264338327950288L * i / nfft);
    std::vector<double> crt((size_t)(f2 + 1) * (m2 + 1));
    std::vector<long double> accj(m2 + 1);
    for (int k = 0; k <= f2; ++k) {
      const long double wk = (k == 0 || k == f2) ? 1.0L : 2.0L;
      std::fill(accj.begin(), accj.end(), 0.0L);
      for (int n = 0; n <= f2; ++n) {
        const long double c = cs[(size_t)((int64_t)k * n % nfft)];
        const double* fr = frq.data() + (size_t)n * (m2 + 1);
        for (int j = 0; j <= m2; ++j) accj[j] += c * (long double)fr[j];
      }
      for (int j = 0; j <= m2; ++j) crt[(size_t)k * (m2 + 1) + j] = (double)(accj[j] * wk / nfft);
    }
    if (upload(crt, &t.crT) != ITTS_OK) return nullptr;
    if (m <= 63) {
      t.kpad = (f2 + 1 + 63) / 64 * 64;
      std::vector<double> crp((size_t)t.kpad * 128, 0.0);
      for (int k = 0; k <= f2; ++k)
        for (int j = 0; j <= m2; ++j) crp[(size_t)k * 128 + j] = crt[(size_t)k * (m2 + 1) + j];
      if (upload(crp, &t.crP) != ITTS_OK) return nullptr;
    }
    // initT = (H Ci)^T . fwd, H = diag(1/2, 1, ..., 1, 1/2): the initial mel-cepstrum from the log periodogram
    // (mcep.c: c = ifft(log x); c[0] /= 2; c[f2] /= 2; mc = freqt(c, +alpha))
    std::vector<double> ini((size_t)(f2 + 1) * (m + 1));
    std::vector<long double> accm(m + 1);
    for (int k = 0; k <= f2; ++k) {
      const long double wk = (k == 0 || k == f2) ? 1.0L : 2.0L;
      std::fill(accm.begin(), accm.end(), 0.0L);
      for (int n = 0; n <= f2; ++n) {
        const long double c = cs[(size_t)((int64_t)k * n % nfft)] * ((n == 0 || n == f2) ? 0.5L : 1.0L);
        const double* fw = fwd.data() + (size_t)n * (m + 1);
        for (int j = 0; j <= m; ++j) accm[j] += c * (long double)fw[j];
      }
      for (int j = 0; j <= m; ++j) ini[(size_t)k * (m + 1) + j] = (double)(accm[j] * wk / nfft);
    }
    if (upload(ini, &t.initT) != ITTS_OK) return nullptr;
    if (m <= 63) {
      const int kpad = (f2 + 1 + 63) / 64 * 64;
      std::vector<double> ip((size_t)kpad * 64, 0.0);
      for (int k = 0; k <= f2; ++k)
        for (int j = 0; j <= m; ++j) ip[(size_t)k * 64 + j] = ini[(size_t)k * (m + 1) + j];
      if (upload(ip, &t.initP) != ITTS_OK) return nullptr;
    }
  }
  if ((need_spec || need_fwd_frq) && !t.specT) {
    // specT = inv . C with C[n][k] = cos(2 pi k n / 2 f2): the real part of the one-sided transform of
    // c'[0 .. f2] (same cosines, same sums)
    const int nfft = 2 * f2;
    std::vector<long double> cs(nfft);
    for (int i = 0; i < nfft; ++i) cs[i] = cosl(2.0L * 3.14159265358979323846264338327950288L * i / nfft);
    std::vector<double> spec((size_t)(m + 1) * (f2 + 1));
    for (int j = 0; j <= m; ++j) {
      freqt_unit(j, f2, -alpha, g.data(), d.data());
      for (int k = 0; k <= f2; ++k) {
        long double acc = 0.0L;
        for (int n = 0; n <= f2; ++n) acc += (long double)g[n] * cs[(size_t)((int64_t)k * n % nfft)];
        spec[(size_t)j * (f2 + 1) + k] = (double)acc;
      }
    }
    if (upload(spec, &t.specT) != ITTS_OK) return nullptr;
    if (m <= 63) {
      const int kpad = (f2 + 1 + 63) / 64 * 64;
      std::vector<double> sp((size_t)64 * kpad, 0.0);
      for (int j = 0; j <= m; ++j)
        for (int k = 0; k <= f2; ++k) sp[(size_t)j * kpad + k] = spec[(size_t)j * (f2 + 1) + k];
      if (upload(sp, &t.specP) != ITTS_OK) return nullptr;
    }
  }
  if (need_mgc && !t.b1T) {
    std::vector<double> b1((size_t)(m + 1) * (f2 + 1)), p2((size_t)(f2 + 1) * (m2 + 1));
    for (int j = 0; j <= m; ++j) {
      b2c_unit(j, f2, -alpha, g.data(), d.data());
      for (int i = 0; i <= f2; ++i) b1[(size_t)j * (f2 + 1) + i] = g[i];
    }
    for (int i = 0; i <= f2; ++i) {
      b2c_unit(i, m2, alpha, g.data(), d.data());
      for (int j = 0; j <= m2; ++j) p2[(size_t)i * (m2 + 1) + j] = g[j];
    }
    if (upload(b1, &t.b1T) != ITTS_OK) return nullptr;
    if (upload(p2, &t.p2T) != ITTS_OK) return nullptr;
  }
  return &t;
}

namespace {
// A ring of page-locked staging slots per device.  32 slots: an entry point uploads two or three
// small tables, so the host may run ~10 entry points ahead of the GPU before it has to wait for a slot.
constexpr unsigned kStageSlots = 32;
struct StageSlot {
  void* p = nullptr;
  size_t cap = 0;
  hipEvent_t ev = nullptr;
  bool pending = false;    // a copy out of this slot may still be in flight (ev marks its end)
  bool busy = false;       // a thread is filling this slot right now
};
struct StageRing {
  StageSlot slot[kStageSlots];
  unsigned next = 0;
};
std::mutex g_stage_mutex;
std::map<int, StageRing> g_stage;
}  // namespace

// A free staging slot of the current device holding at least `bytes` (its last use waited for), marked busy.
static int stage_acquire(size_t bytes, const char* who, StageSlot** out) {
  int dev = 0;
  ITTS_HIP_CHECK(hipGetDevice(&dev));
  StageSlot* sl = nullptr;
  hipEvent_t wait_for = nullptr;
  {
    // the mutex only covers the choice of a slot: waiting for the slot's last copy (another stream's
    // queue may be long) and growing it happen outside, so that other threads keep uploading
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    StageRing& ring = g_stage[dev];
    for (unsigned tries = 0; tries < kStageSlots && !sl; ++tries) {
      StageSlot& c = ring.slot[ring.next++ % kStageSlots];
      if (!c.busy) sl = &c;
    }
    if (!sl) {
      set_error(std::string(who) + ": every staging slot is being filled by another thread");
      return ITTS_E_HIP;
    }
    sl->busy = true;
    if (sl->pending) wait_for = sl->ev;
  }
  hipError_t e = hipSuccess;
  if (wait_for) e = hipEventSynchronize(wait_for);        // the work that last used this slot has left it
  if (e == hipSuccess && sl->cap < bytes) {
    if (sl->p) (void)hipHostFree(sl->p);
    sl->p = nullptr;
    sl->cap = 0;
    const size_t cap = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 16);
    e = hipHostMalloc(&sl->p, cap, hipHostMallocDefault);
    if (e == hipSuccess) sl->cap = cap;
  }
  if (e == hipSuccess && !sl->ev) e = hipEventCreateWithFlags(&sl->ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    sl->pending = false;
    sl->busy = false;
    set_error(std::string(who) + ": " + hipGetErrorString(e));
    return ITTS_E_HIP;
  }
  *out = sl;
  return ITTS_OK;
}

// Hands the slot back: `pending` = work queued on s still reads it (an event marks its end).  If the event cannot
// be recorded the work is waited for on the spot rather than leaving the slot to be overwritten under it.
static int stage_release(StageSlot* sl, bool pending, hipStream_t s, const char* who) {
  hipError_t e = hipSuccess;
  if (pending) {
    e = hipEventRecord(sl->ev, s);
    if (e != hipSuccess) {
      (void)hipStreamSynchronize(s);
      pending = false;
    }
  }
  {
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    sl->pending = pending;
    sl->busy = false;
  }
  if (e != hipSuccess) {
    set_error(std::string(who) + ": " + hipGetErrorString(e));
    return ITTS_E_HIP;
  }
  return ITTS_OK;
}

int staged_upload(void* d_dst, const void* src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return ITTS_OK;
  StageSlot* sl = nullptr;
  {
    const int rc = stage_acquire(bytes, "staged_upload", &sl);
    if (rc) return rc;
  }
  std::memcpy(sl->p, src, bytes);
  const hipError_t e = hipMemcpyAsync(d_dst, sl->p, bytes, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) {
    (void)stage_release(sl, false, s, "staged_upload");
    set_error(std::string("staged_upload: ") + hipGetErrorString(e));
    return ITTS_E_HIP;
  }
  return stage_release(sl, true, s, "staged_upload");
}

int pinned_table_begin(const void* src, size_t bytes, PinnedTable* t) {
  StageSlot* sl = nullptr;
  const int rc = stage_acquire(bytes ? bytes : 1, "pinned_table", &sl);
  if (rc) return rc;
  std::memcpy(sl->p, src, bytes);
  t->p = sl->p;
  t->slot = sl;
  return ITTS_OK;
}

int pinned_table_end(PinnedTable* t, hipStream_t s) {
  if (!t->slot) return ITTS_OK;
  StageSlot* sl = static_cast<StageSlot*>(t->slot);
  t->slot = nullptr;
  return stage_release(sl, true, s, "pinned_table");
}

int upload_i64(const int64_t* h, int n, int64_t** d_out, hipStream_t s) {
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)d_out, (size_t)n * sizeof(int64_t), s));
  return staged_upload(*d_out, h, (size_t)n * sizeof(int64_t), s);
}

}  // namespace itts
