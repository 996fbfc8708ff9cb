// Per-device read-only tables shared by the WORLD / SPTK kernels (twiddles, frequency-warping
// matrices).  Created lazily on the first GPU call of a device; never touched before that, so
// forked DataLoader workers do not inherit an initialised HIP runtime.
#pragma once
#include <map>
#include <tuple>

#include "common.h"

namespace itts {

struct FreqtTables {
  // SPTK freqt / frqtr as matrices, stored input-major ("transposed") so that consecutive
  // lanes read consecutive outputs:
  double* fwdT = nullptr;  // [f2+1][m+1]   mc[j]  = sum_i fwdT[i][j] * c[i]      freqt(c, f2 -> m, +a)
  double* invT = nullptr;  // [m+1][f2+1]   c'[i]  = sum_j invT[j][i] * mc[j]    freqt(mc, m -> f2, -a)
  double* frqT = nullptr;  // [f2+1][2m+1]  cr[j]  = sum_i frqT[i][j] * r[i]     frqtr(r, f2 -> 2m, +a)
  int m = 0, f2 = 0;
  double alpha = 0;
};

struct DeviceContext {
  int device = -1;
  double2* twiddles = nullptr;  // [TW_N/2] exp(+2 pi i k / TW_N)
  std::map<std::tuple<int, int, long long>, FreqtTables> freqt;
};

// Returns the context of the current device (creating it on first use); nullptr + error set on
// failure.
DeviceContext* get_context();
// Returns warping tables for (order m, f2 = fftlen/2, alpha); nullptr + error on failure.
// need_fwd_frq = false builds only invT (enough for mgc2sp).
const FreqtTables* get_freqt(DeviceContext* ctx, int m, int f2, double alpha, bool need_fwd_frq);

// Small stream-ordered device copy of a host int64 array (offsets). Caller frees with
// hipFreeAsync on the same stream.
int upload_i64(const int64_t* h, int n, int64_t** d_out, hipStream_t s);

}  // namespace itts
