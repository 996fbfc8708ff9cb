// Checks csrc/select_largest.h (the sum of a set without its R largest, as d4c_kernel uses it) against a
// sort on the host: random sets with many repeated values, sets of one value, fewer distinct values than R,
// R = 1 .. 66, 1025 and 2049 values.  The kernel side sums in another order than the host, so the
// comparison is relative (1e-13); the SET that is left out must be right, which values from a coarse
// grid (sums exact in either order) check exactly.
// build + run: scripts/select_lab/run.sh
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "select_largest.h"

using namespace itts;

template <int MPER>
__global__ __launch_bounds__(256) void lab_kernel(const double* __restrict__ v, int n, int R, int sets, double* __restrict__ out) {
  __shared__ double lists[4 * 72];
  __shared__ double red[8];
  for (int s = blockIdx.x; s < sets; s += gridDim.x) {
    const double* x = v + (size_t)s * n;
    double mine[MPER];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < MPER; ++i) {
      const int k = (int)threadIdx.x + i * 256;
      mine[i] = -1.0;
      if (k < n) { mine[i] = x[k]; ++cnt; }
    }
#pragma unroll
    for (int i = 1; i < MPER; ++i)
#pragma unroll
      for (int j = i; j > 0; --j)
        if (mine[j] > mine[j - 1]) { const double t = mine[j]; mine[j] = mine[j - 1]; mine[j - 1] = t; }
    const double rest = d4c_rest_without_largest<MPER>(mine, cnt, R, lists, red);
    if (threadIdx.x == 0) out[s] = rest;
    __syncthreads();
  }
}

static int run(int n, int sets, unsigned seed) {
  std::mt19937_64 rng(seed);
  std::vector<double> v((size_t)sets * n), ref(sets);
  std::vector<int> Rs(sets);
  int bad = 0;
  double worst = 0.0;
  for (int kind = 0; kind < 4; ++kind) {
    for (int R = 1; R <= 66; R += (R < 8 ? 1 : 7)) {
      for (int s = 0; s < sets; ++s) {
        double* x = v.data() + (size_t)s * n;
        const int mode = (s + kind) % 4;
        for (int i = 0; i < n; ++i) {
          if (mode == 0) x[i] = std::ldexp((double)(rng() % 4096), -12 + (int)(rng() % 30));     // wide range
          else if (mode == 1) x[i] = (double)(rng() % 8);                                          // eight distinct values
          else if (mode == 2) x[i] = (rng() % 64 == 0) ? 1024.0 : 0.25;                            // a few large ones, ties at the cut
          else x[i] = 3.0;                                                                         // one value
        }
        std::vector<double> t(x, x + n);
        std::sort(t.begin(), t.end());
        double sum = 0.0;
        for (int i = 0; i < n - R; ++i) sum += t[i];
        ref[s] = sum;
      }
      double *dv = nullptr, *dout = nullptr;
      hipMalloc(&dv, v.size() * 8);
      hipMalloc(&dout, sets * 8);
      hipMemcpy(dv, v.data(), v.size() * 8, hipMemcpyHostToDevice);
      if (n <= 5 * 256) hipLaunchKernelGGL(lab_kernel<5>, dim3(256), dim3(256), 0, 0, dv, n, R, sets, dout);
      else hipLaunchKernelGGL(lab_kernel<9>, dim3(256), dim3(256), 0, 0, dv, n, R, sets, dout);
      std::vector<double> got(sets);
      if (hipMemcpy(got.data(), dout, sets * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 1; }
      hipFree(dv);
      hipFree(dout);
      for (int s = 0; s < sets; ++s) {
        const double e = std::fabs(got[s] - ref[s]) / std::max(1e-300, std::fabs(ref[s]));
        const bool exact_kind = ((s + kind) % 4) != 0;
        if ((exact_kind && got[s] != ref[s]) || (!exact_kind && e > 1e-13)) {
          if (bad < 5) printf("n %d R %d set %d mode %d: got %.17g want %.17g\n", n, R, s, (s + kind) % 4, got[s], ref[s]);
          ++bad;
        }
        if (!exact_kind) worst = std::max(worst, e);
      }
    }
  }
  printf("n %4d: %d sets x 15 values of R x 4 mixes, %d wrong, worst relative difference of the wide-range sets %.1e\n",
         n, sets, bad, worst);
  return bad;
}

int main() {
  int bad = run(1025, 512, 7) + run(2049, 512, 11) + run(300, 256, 3);
  printf(bad ? "FAILED\n" : "select lab: all sums agree\n");
  return bad ? 1 : 0;
}
