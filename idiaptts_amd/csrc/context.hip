// Device context: twiddle table and SPTK frequency-warping matrices (see context.h).
#include "context.h"

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
  return ITTS_OK;
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

static int upload(const std::vector<double>& h, double** d) {
  ITTS_HIP_CHECK(hipMalloc((void**)d, h.size() * sizeof(double)));
  ITTS_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  return ITTS_OK;
}

const FreqtTables* get_freqt(DeviceContext* ctx, int m, int f2, double alpha, bool need_fwd_frq) {
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
  }
  return &t;
}

int upload_i64(const int64_t* h, int n, int64_t** d_out, hipStream_t s) {
  ITTS_HIP_CHECK(hipMallocAsync((void**)d_out, (size_t)n * sizeof(int64_t), s));
  ITTS_HIP_CHECK(hipMemcpyAsync(*d_out, h, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, s));
  return ITTS_OK;
}

}  // namespace itts
