// Lab for csrc/wave_fft.h (run on the GPU box: bash scripts/wave_fft_lab/run.sh):
//  1. bits: cfft512 / rfft1024 / irfft1024 of a wave against wd::fft_lds / rfft_lds / irfft_lds of a
//     256-thread workgroup on the same random inputs (must be identical bit for bit, up to the sign of zeros);
//  2. time: R transforms in a loop per wave (8 waves per CU, persistent) against R transforms per
//     workgroup (the structure of the frame kernels), same total number of transforms.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../idiaptts_amd/csrc/wave_fft.h"
#include "../../idiaptts_amd/csrc/world_dev.h"

using namespace itts;

namespace itts { void set_error(const std::string&) {} }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// mode 0: complex forward, 1: complex inverse, 2: rfft, 3: irfft (input X[0 .. N]); N = 64 R complex points
template <int R>
__global__ __launch_bounds__(256) void old_kernel(const double2* in, double2* out, const double2* g_tw, int mode, int reps) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int N = 64 * R, LOGN = R == 8 ? 9 : 10, P = N + 1;
  double2* tw = reinterpret_cast<double2*>(sm);
  double2* z = tw + N;
  wd::load_twiddles(tw, g_tw, 2 * N);    // g_tw: the full-size table, strided to the 2 N-point one
  const int nin = mode == 3 ? N + 1 : N;
  for (int i = threadIdx.x; i < P; i += 256) z[i] = i < nin ? in[(size_t)blockIdx.x * P + i] : make_double2(0, 0);
  __syncthreads();
  for (int r = 0; r < reps; ++r) {
    if (mode == 0) wd::fft_lds(z, N, LOGN, tw, 2 * N, -1);
    else if (mode == 1) wd::fft_lds(z, N, LOGN, tw, 2 * N, +1);
    else if (mode == 2) wd::rfft_lds(z, 2 * N, LOGN + 1, tw, 2 * N);
    else wd::irfft_lds(z, 2 * N, LOGN + 1, tw, 2 * N);
  }
  for (int i = threadIdx.x; i < P; i += 256) out[(size_t)blockIdx.x * P + i] = z[i];
}

template <int R>
__global__ __launch_bounds__(256, R == 8 ? 2 : 1) void new_kernel(const double2* in, double2* out, const double2* g_tw, int mode,
                                                                   int reps, int n_items) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int N = 64 * R, P = N + 1;
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  typename wf::PlanOf<R>::type p;
  wf::table_init<R>(sm, g_tw);
  wf::plan_init(p, g_tw, sm + wf::table_bytes<R>() + wv * wf::lds_bytes<R>(), sm);
  const int nw = gridDim.x * 4;
  for (int item = blockIdx.x * 4 + wv; item < n_items; item += nw) {
    double2 z[R], xh = make_double2(0, 0);
#pragma unroll
    for (int q = 0; q < R; ++q) z[q] = in[(size_t)item * P + l + 64 * q];
    if (mode == 3) xh = in[(size_t)item * P + N];
    for (int r = 0; r < reps; ++r) {
      if (mode == 0) wf::cfft(z, p, -1.0);
      else if (mode == 1) wf::cfft(z, p, +1.0);
      else if (mode == 2) wf::rfft<R>(z, xh, p);
      else wf::irfft<R>(z, xh, p);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) out[(size_t)item * P + l + 64 * q] = z[q];
    if (l == 0) out[(size_t)item * P + N] = xh;
  }
}

template <int R>
int run(const std::vector<double2>& tw_full) {
  constexpr int N = 64 * R, P = N + 1;
  const int B = 8192 * 8 / R;
  std::vector<double2> tw(N), in((size_t)B * P);
  std::mt19937_64 rng(1 + R);
  std::normal_distribution<double> nd(0.0, 1.0);
  for (auto& v : in) v = make_double2(nd(rng), nd(rng));
  for (int k = 0; k < N; ++k) tw[k] = tw_full[(size_t)k * (wd::TW_N / (2 * N))];
  double2 *d_tw, *d_twf, *d_in, *d_o1, *d_o2;
  CK(hipMalloc(&d_tw, N * 16)); CK(hipMalloc(&d_twf, tw_full.size() * 16));
  CK(hipMalloc(&d_in, in.size() * 16)); CK(hipMalloc(&d_o1, in.size() * 16)); CK(hipMalloc(&d_o2, in.size() * 16));
  CK(hipMemcpy(d_tw, tw.data(), N * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_twf, tw_full.data(), tw_full.size() * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_in, in.data(), in.size() * 16, hipMemcpyHostToDevice));
  const size_t lds_old = (N + N + 2) * 16, lds_new = wf::table_bytes<R>() + 4 * wf::lds_bytes<R>();
  CK(hipFuncSetAttribute((const void*)new_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
  CK(hipFuncSetAttribute((const void*)old_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_old));
  std::vector<double2> o1(in.size()), o2(in.size());
  const char* names[4] = {"cfft fwd", "cfft inv", "rfft", "irfft"};
  int bad_total = 0;
  for (int mode = 0; mode < 4; ++mode) {
    CK(hipMemset(d_o1, 0, in.size() * 16)); CK(hipMemset(d_o2, 0, in.size() * 16));
    hipLaunchKernelGGL(old_kernel<R>, dim3(B), dim3(256), lds_old, 0, d_in, d_o1, d_twf, mode, 1);
    hipLaunchKernelGGL(new_kernel<R>, dim3(512), dim3(256), lds_new, 0, d_in, d_o2, d_tw, mode, 1, B);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o1.data(), d_o1, in.size() * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o2.data(), d_o2, in.size() * 16, hipMemcpyDeviceToHost));
    const int nout = (mode == 2) ? N + 1 : N;
    long bad = 0, zsign = 0;
    double worst = 0;
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < nout; ++i) {
        const double2 a = o1[(size_t)b * P + i], c = o2[(size_t)b * P + i];
        for (int h = 0; h < 2; ++h) {
          const double x = h ? a.y : a.x, y = h ? c.y : c.x;
          if (memcmp(&x, &y, 8) != 0) {
            if (x == 0.0 && y == 0.0) { ++zsign; continue; }
            ++bad;
            worst = fmax(worst, fabs(x - y) / fmax(1e-300, fabs(x)));
            if (bad <= 3) printf("  mismatch N %d mode %d item %d bin %d %s: old %.17g new %.17g\n", N, mode, b, i, h ? "im" : "re", x, y);
          }
        }
      }
    printf("%-8s %4d : %ld of %ld values differ (worst rel %.2e), %ld zeros of opposite sign\n", names[mode], N, bad,
           (long)B * nout * 2, worst, zsign);
    bad_total += bad != 0;
  }
  // timing: 64 transforms back to back per input, B inputs
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 4; mode += 2) {
    const int reps = 64;
    float ms_old = 0, ms_new = 0, ms_one = 0;
    for (int it = 0; it < 3; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(old_kernel<R>, dim3(B), dim3(256), lds_old, 0, d_in, d_o1, d_twf, mode, reps);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_old, e0, e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(new_kernel<R>, dim3(512), dim3(256), lds_new, 0, d_in, d_o2, d_tw, mode, reps, B);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_new, e0, e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(new_kernel<R>, dim3(256), dim3(256), lds_new, 0, d_in, d_o2, d_tw, mode, reps, B);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_one, e0, e1));
    }
    printf("%-8s %4d x %d x %d: workgroup-per-transform %.3f ms (%.1f ns each), wave-per-transform %.3f ms (%.1f ns each): "
           "%.2fx; one workgroup per CU %.3f ms\n", names[mode], N, B, reps, ms_old, ms_old * 1e6 / ((double)B * reps), ms_new,
           ms_new * 1e6 / ((double)B * reps), ms_old / ms_new, ms_one);
  }
  CK(hipFree(d_tw)); CK(hipFree(d_twf)); CK(hipFree(d_in)); CK(hipFree(d_o1)); CK(hipFree(d_o2));
  return bad_total;
}

int main() {
  std::vector<double2> tw_full((size_t)wd::TW_N / 2);
  for (size_t k = 0; k < tw_full.size(); ++k)
    tw_full[k] = make_double2(std::cos(2.0 * M_PI * k / wd::TW_N), std::sin(2.0 * M_PI * k / wd::TW_N));
  const int bad = run<8>(tw_full) + run<16>(tw_full);
  return bad;
}
