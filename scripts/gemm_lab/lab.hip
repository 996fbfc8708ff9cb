// GEMM lab: the LDS-DMA ring kernel (csrc/gemm_ring.h) against the library's register-staged
// kernels and an fp64 reference, on the eight GEMMs of one FF training step.  Not product code.
//   build: scripts/gemm_lab/build.sh        run: scripts/gemm_lab/lab [reps]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../idiaptts_amd/csrc/gemm_ring.h"
#include "../../include/idiaptts_amd.h"

using namespace itts::ring;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// C[m][n] = sum_k A(m,k) B(n,k); element (o,k) of a row-form operand is P[o*ld+k], of a col-form
// operand P[k*ld+o].  fp64 accumulation.
__global__ void ref_gemm(const float* A, int64_t lda, int arow, const float* B, int64_t ldb, int brow,
                         double* C, int64_t M, int N, int64_t K) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= M * N) return;
  const int64_t m = idx / N;
  const int n = (int)(idx % N);
  double s = 0.0;
  for (int64_t k = 0; k < K; ++k) {
    const float a = arow ? A[m * lda + k] : A[k * lda + m];
    const float b = brow ? B[n * ldb + k] : B[k * ldb + n];
    s += (double)a * (double)b;
  }
  C[idx] = s;
}

__global__ void fill_rand(float* p, int64_t rows, int64_t ld, int64_t cols, uint32_t seed, float scale) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= rows * ld) return;
  const int64_t c = i % ld;
  uint32_t x = (uint32_t)i * 2654435761u ^ seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  p[i] = c < cols ? ((float)(x >> 8) / 16777216.f - 0.5f) * scale : 0.f;
}

__global__ void reduce_slabs_ref(const float* slabs, int S, int64_t n, float* out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int z = 0; z < S; ++z) s += slabs[(int64_t)z * n + i];
  out[i] = s;
}

static float* dalloc(int64_t rows, int64_t ld, int64_t cols, uint32_t seed, float scale = 1.f) {
  float* p;
  CK(hipMalloc(&p, (size_t)(rows * ld) * 4 + 4096));
  hipLaunchKernelGGL(fill_rand, dim3((unsigned)((rows * ld + 255) / 256)), dim3(256), 0, 0, p, rows, ld, cols,
                     seed, scale);
  return p;
}

template <bool A_ROW, bool B_ROW, int EPI, int WM>
static void launch_ring(Args g, hipStream_t s, int maxwg = 512) {
  constexpr int BMT = 64 * WM, BNT = 32 * (4 / WM);
  g.tiles_m = (g.M + BMT - 1) / BMT;
  g.tiles_n = (g.N + BNT - 1) / BNT;
  const int64_t ntiles = (int64_t)g.tiles_m * g.tiles_n * g.splitk;
  const int G = (int)std::min<int64_t>(maxwg, (ntiles + 7) / 8 * 8);
  static bool once = false;
  if (!once) {
    once = true;
    int nb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_ring_kernel<A_ROW, B_ROW, EPI, WM>, THREADS, 0));
    printf("   occupancy API: %d workgroups of %d threads per CU (LDS %d B)\n", nb, THREADS, LDS_BYTES);
  }
  hipLaunchKernelGGL((gemm_ring_kernel<A_ROW, B_ROW, EPI, WM>), dim3(G), dim3(THREADS), 0, s, g);
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  template <class F> double us(F f, int reps) {
    for (int i = 0; i < reps; ++i) f();   // settle the clock on this kernel
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
  }
};

static double max_rel_err(const float* d_c, int64_t ldc, const double* d_ref, int64_t M, int N) {
  std::vector<float> c((size_t)(M * ldc));
  std::vector<double> r((size_t)(M * N));
  CK(hipMemcpy(c.data(), d_c, c.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r.data(), d_ref, r.size() * 8, hipMemcpyDeviceToHost));
  double num = 0, den = 0;
  for (int64_t m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      const double d = (double)c[m * ldc + n] - r[m * N + n];
      num = std::max(num, std::fabs(d));
      den = std::max(den, std::fabs(r[m * N + n]));
    }
  return num / den;
}
static double max_abs_diff(const float* d_a, const float* d_b, int64_t n) {
  std::vector<float> a((size_t)n), b((size_t)n);
  CK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
  double m = 0;
  for (int64_t i = 0; i < n; ++i) {
    const double d = std::fabs((double)a[i] - (double)b[i]);
    if (!(d <= m)) m = d;   // NaN propagates
  }
  return m;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const int64_t M = argc > 2 ? atoll(argv[2]) : 39129;
  const int maxwg = argc > 3 ? atoi(argv[3]) : 512;
  const char* only = argc > 4 ? argv[4] : "";
  Timer T;
  float* x = dalloc(M, 428, 425, 1);
  float* h1 = dalloc(M, 512, 512, 2, 1.8f);
  float* h2 = dalloc(M, 512, 512, 3, 1.8f);
  float* dz3 = dalloc(M, 188, 187, 4);
  float* dz2 = dalloc(M, 512, 512, 5);
  float* dz1 = dalloc(M, 512, 512, 6);
  float* w1 = dalloc(512, 428, 425, 7, 0.1f);
  float* w2 = dalloc(512, 512, 512, 8, 0.1f);
  float* w3 = dalloc(187, 512, 512, 9, 0.1f);
  float* b1 = dalloc(1, 512, 512, 10);
  float* b3 = dalloc(1, 188, 187, 11);
  float* o_old = dalloc(M, 512, 0, 0);
  float* o_new = dalloc(M, 512, 0, 0);
  float* slabs;
  CK(hipMalloc(&slabs, (size_t)64 * 512 * 512 * 4));
  float* dw_old = dalloc(512, 512, 0, 0);
  float* dw_new = dalloc(512, 512, 0, 0);
  void* ws;
  CK(hipMalloc(&ws, 256 << 20));
  double* ref;
  CK(hipMalloc(&ref, (size_t)M * 512 * 8));
  CK(hipDeviceSynchronize());

  uint64_t* stamps;
  CK(hipMalloc(&stamps, 4096 * 32));
  CK(hipMemset(stamps, 0, 4096 * 32));
  auto clock_report = [&](const char* name, int G) {
    std::vector<uint64_t> h(4 * G);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz, us;
    uint64_t t0 = ~0ull;
    for (int i = 0; i < G; ++i) if (h[4 * i + 1]) t0 = std::min(t0, h[4 * i + 2]);
    std::vector<double> st[2], en[2];
    for (int i = 0; i < G; ++i) {
      if (h[4 * i + 1] == 0) continue;
      ghz.push_back((double)h[4 * i] / (double)h[4 * i + 1] * 0.1);
      us.push_back((double)h[4 * i + 1] * 0.01);
      const int second = (i >> 3) >= (G >> 4);   // rank in the upper half
      st[second].push_back((double)(h[4 * i + 2] - t0) * 0.01);
      en[second].push_back((double)(h[4 * i + 2] - t0 + h[4 * i + 1]) * 0.01);
    }
    if (ghz.empty()) return;
    for (int b = 0; b < 2; ++b) {
      if (st[b].empty()) continue;
      std::sort(st[b].begin(), st[b].end());
      std::sort(en[b].begin(), en[b].end());
      printf("   ranks %s: start min %.1f med %.1f max %.1f us; end min %.1f med %.1f max %.1f us\n", b ? "upper" : "lower",
             st[b].front(), st[b][st[b].size() / 2], st[b].back(), en[b].front(), en[b][en[b].size() / 2], en[b].back());
    }
    // which CU: HW_ID bits: cu_id [11:8], sh_id [12], se_id [15:13] (gfx9 layout), xcc via blockIdx
    std::sort(ghz.begin(), ghz.end());
    std::sort(us.begin(), us.end());
    printf("   %-20s in-kernel clock median %.3f GHz (min %.3f max %.3f); workgroup lifetime median %.1f us, max %.1f us\n",
           name, ghz[ghz.size() / 2], ghz.front(), ghz.back(), us[us.size() / 2], us.back());
  };
  std::vector<std::function<void()>> step_old, step_new;
  struct Row { std::string name; double us_old, us_new, flops; double err, bitdiff; };
  std::vector<Row> rows;

  auto run_rowA = [&](const char* name, const float* A, int64_t lda, const float* B, int64_t ldb, bool brow,
                      int N, int64_t K, int epi, const float* bias, const float* aux, int64_t ldaux, int act,
                      int64_t ldc, auto old_fn) {
    if (*only && !strstr(name, only)) return;
    Args g{};
    g.A = A; g.lda = (int)lda; g.B = B; g.ldb = (int)ldb; g.C = o_new; g.ldc = (int)ldc; g.M = (int)M; g.N = N; g.K = (int)K;
    g.bias = bias; g.aux = aux; g.ldaux = (int)ldaux; g.act = act;
    g.kchunk = (int)((K + 31) / 32 * 32); g.splitk = 1; g.slab_stride = 0;
    auto new_fn = [&]() {
      if (brow) launch_ring<true, true, EPI_BIAS_ACT, 2>(g, 0, maxwg);
      else launch_ring<true, false, EPI_DACT, 2>(g, 0, maxwg);
    };
    CK(hipMemset(o_old, 0, (size_t)M * 512 * 4));
    CK(hipMemset(o_new, 0, (size_t)M * 512 * 4));
    const double t_old = T.us(old_fn, reps);
    const double t_new = T.us(new_fn, reps);
    { Args gc = g; step_old.push_back(old_fn);
      step_new.push_back([=]() { if (brow) launch_ring<true, true, EPI_BIAS_ACT, 2>(gc, 0, maxwg); else launch_ring<true, false, EPI_DACT, 2>(gc, 0, maxwg); }); }
    g.stamps = stamps;
    CK(hipGetLastError());
    clock_report("(timed loop)", 512);
    g.stamps = stamps;
    CK(hipMemset(stamps, 0, 1024 * 32));
    for (int i = 0; i < 50; ++i) new_fn();
    CK(hipDeviceSynchronize());
    clock_report(name, 512);
    {   // boundary between two consecutive launches: last end of one, first start of the next
      for (int i = 0; i < 10; ++i) { g.stamps = stamps + (i & 1 ? 4 * 512 : 0); new_fn(); }
      CK(hipDeviceSynchronize());
      std::vector<uint64_t> h(8 * 512);
      CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
      uint64_t end0 = 0, start1 = ~0ull, end1 = 0, start0 = ~0ull;
      for (int i = 0; i < 512; ++i) {
        if (!h[4 * i + 1]) continue;
        start0 = std::min(start0, h[4 * i + 2]);
        end0 = std::max(end0, h[4 * i + 2] + h[4 * i + 1]);
        start1 = std::min(start1, h[4 * (512 + i) + 2]);
        end1 = std::max(end1, h[4 * (512 + i) + 2] + h[4 * (512 + i) + 1]);
      }
      printf("   launch A: first start -> last end %.1f us; gap to launch B's first start %.1f us; launch B %.1f us\n",
             (end0 - start0) * 0.01, ((double)start1 - (double)end0) * 0.01, (end1 - start1) * 0.01);
    }
    g.stamps = nullptr;
    const double bd = max_abs_diff(o_old, o_new, M * ldc);
    if (bd > 1e-3) {
      std::vector<float> a((size_t)(M * ldc)), b((size_t)(M * ldc));
      CK(hipMemcpy(a.data(), o_old, a.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), o_new, b.size() * 4, hipMemcpyDeviceToHost));
      std::vector<int> hr(128, 0), hc(128, 0);
      std::vector<int64_t> tl;
      int64_t bad = 0; int shown = 0;
      for (int64_t m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
        if (std::fabs(a[m * ldc + n] - b[m * ldc + n]) > 1e-3) {
          ++bad; hr[m % 128]++; hc[n % 128]++; tl.push_back((m / 128) * 4 + n / 128);
          if (shown++ < 4) printf("     bad (%lld,%d): old %g new %g\n", (long long)m, n, a[m * ldc + n], b[m * ldc + n]);
        }
      }
      std::sort(tl.begin(), tl.end()); tl.erase(std::unique(tl.begin(), tl.end()), tl.end());
      printf("     %lld bad in %zu tiles (first:", (long long)bad, tl.size());
      for (size_t i = 0; i < tl.size() && i < 12; ++i) printf(" %lld", (long long)tl[i]);
      printf(")\n     rows:"); for (int i = 0; i < 128; ++i) if (hr[i]) printf(" %d:%d", i, hr[i]);
      printf("\n     cols:"); for (int i = 0; i < 128; ++i) if (hc[i]) printf(" %d:%d", i, hc[i]);
      printf("\n");
    }
    rows.push_back({name, t_old, t_new, 2.0 * M * N * (double)K, 0.0, bd});
  };

  run_rowA("fwd1 425->512", x, 428, w1, 428, true, 512, 428, EPI_BIAS_ACT, b1, nullptr, 0, ACT_TANH, 512,
           [&]() { itts_linear_fwd(x, 428, w1, b1, o_old, 512, M, 512, 428, 1, nullptr); });
  run_rowA("fwd2 512->512", h1, 512, w2, 512, true, 512, 512, EPI_BIAS_ACT, b1, nullptr, 0, ACT_TANH, 512,
           [&]() { itts_linear_fwd(h1, 512, w2, b1, o_old, 512, M, 512, 512, 1, nullptr); });
  run_rowA("fwd3 512->187", h2, 512, w3, 512, true, 187, 512, EPI_BIAS_ACT, b3, nullptr, 0, ACT_NONE, 188,
           [&]() { itts_linear_fwd(h2, 512, w3, b3, o_old, 188, M, 187, 512, 0, nullptr); });
  run_rowA("dX2 187->512", dz3, 188, w3, 512, false, 512, 187, EPI_DACT, nullptr, h2, 512, ACT_TANH, 512,
           [&]() { itts_linear_bwd_input(dz3, 188, w3, o_old, 512, h2, 512, 1, M, 187, 512, nullptr); });
  run_rowA("dX1 512->512", dz2, 512, w2, 512, false, 512, 512, EPI_DACT, nullptr, h1, 512, ACT_TANH, 512,
           [&]() { itts_linear_bwd_input(dz2, 512, w2, o_old, 512, h1, 512, 1, M, 512, 512, nullptr); });

  // weight gradients: dw[N][K] = dz^T x
  auto run_dw = [&](const char* name, const float* dz, int64_t lddz, int N, const float* xin, int64_t ldx, int Kc,
                    int wm) {
    if (*only && !strstr(name, only)) return;
    Args g{};
    g.A = dz; g.lda = (int)lddz; g.B = xin; g.ldb = (int)ldx; g.C = slabs; g.ldc = Kc; g.M = N; g.N = Kc; g.K = (int)M;
    const int bmt = 64 * wm, bnt = 32 * (4 / wm);
    const int tiles = ((N + bmt - 1) / bmt) * ((Kc + bnt - 1) / bnt);
    int S = std::max(1, maxwg / tiles);
    int64_t kchunk = ((M + S - 1) / S + 31) / 32 * 32;
    S = (int)((M + kchunk - 1) / kchunk);
    g.kchunk = (int)kchunk; g.splitk = S; g.slab_stride = (int64_t)N * Kc;
    auto new_fn = [&]() {
      if (wm == 2) launch_ring<false, false, EPI_STORE, 2>(g, 0, maxwg);
      else launch_ring<false, false, EPI_STORE, 1>(g, 0, maxwg);
      hipLaunchKernelGGL(reduce_slabs_ref, dim3((unsigned)(((int64_t)N * Kc + 255) / 256)), dim3(256), 0, 0, slabs,
                         S, (int64_t)N * Kc, dw_new);
    };
    auto old_fn = [&]() { itts_linear_bwd_weight(dz, lddz, xin, ldx, dw_old, nullptr, M, N, Kc, ws, 0, nullptr); };
    const double t_old = T.us(old_fn, reps);
    const double t_new = T.us(new_fn, reps);
    if (!strstr(name, "wm1")) { step_old.push_back(old_fn); Args gc = g; int Sc = S;
      step_new.push_back([=]() { if (wm == 2) launch_ring<false, false, EPI_STORE, 2>(gc, 0, maxwg); else launch_ring<false, false, EPI_STORE, 1>(gc, 0, maxwg);
        hipLaunchKernelGGL(reduce_slabs_ref, dim3((unsigned)(((int64_t)N * Kc + 255) / 256)), dim3(256), 0, 0, slabs, Sc, (int64_t)N * Kc, dw_new); }); }
    CK(hipGetLastError());
    hipLaunchKernelGGL(ref_gemm, dim3((unsigned)(((int64_t)N * Kc + 255) / 256)), dim3(256), 0, 0, dz, lddz, 0, xin,
                       ldx, 0, ref, (int64_t)N, Kc, M);
    const double e_new = max_rel_err(dw_new, Kc, ref, N, Kc);
    const double e_old = max_rel_err(dw_old, Kc, ref, N, Kc);
    rows.push_back({std::string(name) + " S=" + std::to_string(S), t_old, t_new, 2.0 * M * N * (double)Kc, e_new,
                    e_old});
  };
  run_dw("dW3 187x512 wm1", dz3, 188, 187, h2, 512, 512, 1);
  run_dw("dW3 187x512 wm2", dz3, 188, 187, h2, 512, 512, 2);
  run_dw("dW2 512x512 wm2", dz2, 512, 512, h1, 512, 512, 2);
  run_dw("dW1 512x428 wm2", dz1, 512, 512, x, 428, 428, 2);

  // fp64 check of row-form cases with a K tail
  for (int Kt : {425, 64, 33, 417}) {
    Args g{};
    g.A = x; g.lda = 428; g.B = w1; g.ldb = 428; g.C = o_new; g.ldc = 512; g.M = (int)M; g.N = 512; g.K = Kt;
    g.kchunk = (Kt + 31) / 32 * 32; g.splitk = 1;
    CK(hipMemset(o_new, 0, (size_t)M * 512 * 4));
    launch_ring<true, true, EPI_STORE, 2>(g, 0, maxwg);
    hipLaunchKernelGGL(ref_gemm, dim3((unsigned)((M * 512 + 255) / 256)), dim3(256), 0, 0, x, (int64_t)428, 1, w1,
                       (int64_t)428, 1, ref, M, 512, (int64_t)Kt);
    CK(hipDeviceSynchronize());
    std::vector<float> c((size_t)(M * 512));
    std::vector<double> r((size_t)(M * 512));
    CK(hipMemcpy(c.data(), o_new, c.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r.data(), ref, r.size() * 8, hipMemcpyDeviceToHost));
    int64_t bad = 0; int shown = 0;
    std::vector<int> hist_r(128, 0), hist_c(64, 0);
    std::vector<int64_t> tiles;
    for (int64_t m = 0; m < M; ++m)
      for (int n = 0; n < 512; ++n) {
        const double d = std::fabs((double)c[m * 512 + n] - r[m * 512 + n]);
        if (d > 1e-4) {
          ++bad;
          hist_r[m % 128]++; hist_c[n % 64]++;
          tiles.push_back((m / 128) * 8 + n / 64);
          if (shown < 2) { printf("     bad (%lld,%d): got %g want %g\n", (long long)m, n, c[m * 512 + n], r[m * 512 + n]); ++shown; }
        }
      }
    printf("row-form K=%d vs fp64: %lld bad of %lld\n", Kt, (long long)bad, (long long)(M * 512));
    if (bad && Kt != 448) {
      std::sort(tiles.begin(), tiles.end());
      tiles.erase(std::unique(tiles.begin(), tiles.end()), tiles.end());
      printf("     %zu tiles:", tiles.size());
      for (size_t i = 0; i < tiles.size() && i < 24; ++i) printf(" %lld", (long long)tiles[i]);
      printf("\n     rows in tile:");
      for (int i = 0; i < 128; ++i) if (hist_r[i]) printf(" %d:%d", i, hist_r[i]);
      printf("\n     cols in tile:");
      for (int i = 0; i < 64; ++i) if (hist_c[i]) printf(" %d:%d", i, hist_c[i]);
      printf("\n");
    }
  }

  if (!*only) {
    const double so = T.us([&]() { for (auto& f : step_old) f(); }, std::max(reps / 2, 10));
    const double sn = T.us([&]() { for (auto& f : step_new) f(); }, std::max(reps / 2, 10));
    double flt = 0;
    for (auto& r : rows) if (r.name.find("wm1") == std::string::npos) flt += r.flops;
    printf("the eight GEMMs in step order, back to back: old %.1f us (%.1f TF, %.3f)   new %.1f us (%.1f TF, %.3f)\n", so,
           flt / so * 1e-6, flt / so * 1e-6 / 157.3, sn, flt / sn * 1e-6, flt / sn * 1e-6 / 157.3);
  }
  double to = 0, tn = 0, fl = 0;
  printf("%-24s %10s %10s %8s %8s  %s\n", "case", "old us", "new us", "old TF", "new TF", "check");
  for (auto& r : rows) {
    printf("%-24s %10.1f %10.1f %8.1f %8.1f  err %.2e / %.2e\n", r.name.c_str(), r.us_old, r.us_new,
           r.flops / r.us_old * 1e-6, r.flops / r.us_new * 1e-6, r.err, r.bitdiff);
    if (r.name.find("wm1") == std::string::npos) { to += r.us_old; tn += r.us_new; fl += r.flops; }
  }
  printf("total (wm2 rows): old %.1f us (%.1f TF, %.3f)   new %.1f us (%.1f TF, %.3f of 157.3)\n", to,
         fl / to * 1e-6, fl / to * 1e-6 / 157.3, tn, fl / tn * 1e-6, fl / tn * 1e-6 / 157.3);
  return 0;
}
