// Mel-generalized cepstral analysis and its inverse for gamma != 0:
//   pysptk.mgcep(amp_sp, order, alpha, gamma, eps=1e-8, min_det=0, etype=1, itype=3)
//       AudioProcessing.extract_mgc (src/data_preparation/audio/AudioProcessing.py:123-140), gamma = -1/3
//   pysptk.mgc2sp(mgc, alpha, gamma, fftlen).real
//       AudioProcessing.mgc_to_amp_sp (:259-275)
// Algorithm (Tokuda, Kobayashi, Masuko, Imai, "Mel-generalized cepstral analysis", ICSLP 1994; SPTK
// mgcep.c): the model spectrum is D(w) = (1 + g C(w~))^(1/g), the criterion eps = 1/2pi int x / |D|^2
// is minimised by Newton steps [Toeplitz(p~) + (1+g) Hankel(q~)] dc = r~, where per frequency bin,
// with s = |1 + g C|^2:  p = x s^(-1/g) / s,  r = p (1 + g C),  q = p (1 + g C)^2 / s.  The first
// step is the g = -1 step from zero (LPC), converted to the target gamma by gc2gc.
//
// One workgroup per frame, everything in LDS; the two frequency-warping maps (SPTK's b2c with -a:
// m -> n, and with +a: n -> 2m) are matrix-vector products against tables built once per
// (order, fft size, alpha) -- the same structure as the per-frame mcep kernel (world_frame.hip).
// This feature type is in no benchmarked configuration: the kernel is complete, not tuned (the
// lockstep fp64-MFMA formulation of mcep_lockstep.hip would apply unchanged).
#include <algorithm>

#include "context.h"
#include "world_dev.h"

namespace itts {
using namespace wd;

struct MgLds {
  const double2* tw;
  double2* zp;     // [f2+1] p spectrum / sequence (flng + 2 doubles)
  double2* zq;     // [f2+1] q
  double2* zr;     // [f2+1] r
  double* xp;      // [f2+1] periodogram
  double* b;       // [m+1]  coefficients (b[0] = gain)
  double* pt;      // [2m+1] p~
  double* qt;      // [2m+1] q~
  double* rt;      // [m+1]  r~
  double* part;    // [4][3][2m+1]
  double* A;       // [m][m+2] augmented system
  double* fcol;    // [m+1]
  double* misc;    // [8]
};

struct MgArgs {
  const double* amp;      // [T, K] amplitude spectrum (or power spectrum: in_is_power)
  int in_is_power;
  int64_t T;
  int flng, logflng, m;
  double alpha, gamma, eps, dd;
  int itr1, itr2;
  const double* b1T;      // [m+1][f2+1]
  const double* p2T;      // [f2+1][2m+1]
  float* out_f32;
  int64_t ld_out;
  double* out_f64;
  int* iters;
  const double2* g_tw;
};

// One Newton step (SPTK mgcep.c newton()) on L.b with exponent g; returns log(eps) in every thread.
__device__ inline double mg_newton(const MgLds& L, const MgArgs& a, double g) {
  const int flng = a.flng, f2 = flng / 2, m = a.m, m2 = 2 * m, n_out = m2 + 1;
  const int t = tid();
  double* pr = reinterpret_cast<double*>(L.zp);
  double* qr = reinterpret_cast<double*>(L.zq);
  double* rr = reinterpret_cast<double*>(L.zr);
  // c = b2c(b[1..m], m -> f2, -alpha), zero padded to flng (thread t owns outputs t + 256 q)
  {
    constexpr int NQ = 5;          // f2 + 1 <= 1280
    double acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
    const double* row = a.b1T + (f2 + 1);          // row 0 multiplies cr[0] = 0
    for (int j = 1; j <= m; ++j, row += (f2 + 1)) {
      const double bj = L.b[j];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int i = t + NT * q;
        acc[q] += row[i <= f2 ? i : 0] * bj;
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = t + NT * q;
      if (i < flng + 2) pr[i] = (i <= f2) ? acc[q] : 0.0;
    }
    for (int i = t + NT * NQ; i < flng + 2; i += NT) pr[i] = 0.0;
  }
  __syncthreads();
  rfft_lds(L.zp, flng, a.logflng, L.tw, flng);       // zp[k] = C(w_k)
  const bool general = (g != 0.0 && g != -1.0);
  for (int k = t; k <= f2; k += NT) {
    const double cr = L.zp[k].x, ci = L.zp[k].y, x = L.xp[k];
    if (g == -1.0) {
      L.zp[k] = make_double2(x, 0.0);
    } else if (g == 0.0) {
      L.zp[k] = make_double2(x / exp(cr + cr), 0.0);
    } else {
      const double tr = 1.0 + g * cr, ti = g * ci;
      const double s = tr * tr + ti * ti;
      double v = x * pow(s, -1.0 / g);
      v /= s;
      L.zp[k] = make_double2(v, 0.0);
      L.zr[k] = make_double2(tr * v, ti * v);
      v /= s;
      L.zq[k] = make_double2((tr * tr - ti * ti) * v, 2.0 * tr * ti * v);
    }
  }
  __syncthreads();
  irfft_lds(L.zp, flng, a.logflng, L.tw, flng);
  if (general) {
    irfft_lds(L.zq, flng, a.logflng, L.tw, flng);
    irfft_lds(L.zr, flng, a.logflng, L.tw, flng);
  }
  // p~, q~ = b2c(., f2 -> 2m, +alpha), r~ = its first m + 1 outputs: wave w sums input rows
  // [w*per, (w+1)*per), lane l owns outputs l and l + 64
  {
    const int wv = t >> 6, ln = t & 63;
    const int per = (f2 + 1 + 3) / 4;
    const int i0 = wv * per, i1 = min(f2 + 1, i0 + per);
    double ap0 = 0.0, ap1 = 0.0, aq0 = 0.0, aq1 = 0.0, ar0 = 0.0, ar1 = 0.0;
    const bool h0 = ln < n_out, h1 = ln + 64 < n_out;
    const double* row = a.p2T + (size_t)i0 * n_out;
    for (int i = i0; i < i1; ++i, row += n_out) {
      const double w0 = row[h0 ? ln : 0], w1 = row[h1 ? ln + 64 : 0];
      const double pv = pr[i];
      ap0 += w0 * pv;
      ap1 += w1 * pv;
      if (general) {
        const double qv = qr[i], rv = rr[i];
        aq0 += w0 * qv;
        aq1 += w1 * qv;
        ar0 += w0 * rv;
        ar1 += w1 * rv;
      }
    }
    double* P = L.part + (size_t)wv * 3 * n_out;
    if (h0) { P[ln] = ap0; P[n_out + ln] = aq0; P[2 * n_out + ln] = ar0; }
    if (h1) { P[ln + 64] = ap1; P[n_out + ln + 64] = aq1; P[2 * n_out + ln + 64] = ar1; }
    __syncthreads();
    for (int j = t; j < n_out; j += NT) {
      double sp = 0.0, sq = 0.0, sr = 0.0;
      for (int w = 0; w < 4; ++w) {
        const double* Pw = L.part + (size_t)w * 3 * n_out;
        sp += Pw[j];
        sq += Pw[n_out + j];
        sr += Pw[2 * n_out + j];
      }
      L.pt[j] = sp;
      L.qt[j] = general ? sq : sp;
      if (j <= m) L.rt[j] = general ? sr : sp;
    }
    __syncthreads();
  }
  // ptrans / qtrans (sequential recursions of length m / 2m), gain, (1 + g) on the Hankel part
  if (t == 0) {
    const double al = a.alpha;
    if (al != 0.0) {
      int mm = m;
      double d = L.pt[mm], o;
      for (mm--; mm > 0; mm--) {
        o = L.pt[mm] + al * d;
        d = L.pt[mm];
        L.pt[mm] = o;
      }
      o = al * d;
      L.pt[0] = (1.0 - al * al) * L.pt[0] + o + o;
      d = L.qt[1];
      for (int i = 2; i <= m2; ++i) {
        o = L.qt[i] + al * d;
        d = L.qt[i];
        L.qt[i] = o;
      }
    }
    double eps = L.rt[0];
    if (g != 0.0)
      for (int i = 1; i <= m; ++i) eps += g * L.rt[i] * L.b[i];
    L.b[0] = sqrt(eps);
    L.misc[1] = eps;
  }
  __syncthreads();
  const double hank = (g == -1.0) ? 0.0 : 1.0 + g;
  const int ld = m + 2;
  for (int idx = t; idx < m * (m + 1); idx += NT) {
    const int i = idx / (m + 1), k = idx - i * (m + 1);
    double v;
    if (k == m) {
      v = L.rt[1 + i];
    } else {
      const int df = i > k ? i - k : k - i;
      v = L.pt[df] + hank * L.qt[2 + i + k];
    }
    L.A[i * ld + k] = v;
  }
  __syncthreads();
  // Gaussian elimination without pivoting (what SPTK's theq does for this symmetric system)
  for (int c = 0; c < m; ++c) {
    const double piv = L.A[c * ld + c];
    for (int r = c + 1 + t; r < m; r += NT) L.fcol[r] = L.A[r * ld + c] / piv;
    __syncthreads();
    const int w = m + 1 - (c + 1);           // columns c+1 .. m (rhs in column m)
    for (int idx = t; idx < (m - 1 - c) * w; idx += NT) {
      const int r = c + 1 + idx / w, k = c + 1 + idx % w;
      L.A[r * ld + k] -= L.fcol[r] * L.A[c * ld + k];
    }
    __syncthreads();
  }
  if (t < 64) {
    for (int r = m - 1; r >= 0; --r) {
      double s = 0.0;
      for (int k = r + 1 + t; k < m; k += 64) s += L.A[r * ld + k] * L.fcol[k];
      s = wave_sum(s);
      if (t == 0) L.fcol[r] = (L.A[r * ld + m] - s) / L.A[r * ld + r];
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int j = t; j < m; j += NT) L.b[1 + j] += L.fcol[j];
  const double ep = log(L.misc[1]);
  __syncthreads();
  return ep;
}

// SPTK gnorm / ignorm / b2mc / mc2b / gc2gc on a short vector in LDS (thread 0 only)
__device__ inline void mg_gnorm(double* c, int m, double g) {
  if (g != 0.0) {
    const double k = 1.0 + g * c[0];
    for (int i = m; i >= 1; --i) c[i] /= k;
    c[0] = pow(k, 1.0 / g);
  } else {
    c[0] = exp(c[0]);
  }
}
__device__ inline void mg_ignorm(double* c, int m, double g) {
  if (g != 0.0) {
    const double k = pow(c[0], g);
    for (int i = m; i >= 1; --i) c[i] *= k;
    c[0] = (k - 1.0) / g;
  } else {
    c[0] = log(c[0]);
  }
}
__device__ inline void mg_b2mc(double* b, int m, double a) {
  double d = b[m], o;
  for (m--; m >= 0; m--) {
    o = b[m] + a * d;
    d = b[m];
    b[m] = o;
  }
}
__device__ inline void mg_mc2b(double* b, int m, double a) {
  for (m--; m >= 0; m--) b[m] = b[m] - a * b[m + 1];
}

__global__ __launch_bounds__(NT) void mgcep_kernel(MgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int flng = a.flng, f2 = flng / 2, m = a.m, m2 = 2 * m;
  char* p = smem;
  MgLds L;
  L.zp = reinterpret_cast<double2*>(p); p += (size_t)(f2 + 1) * 16;
  L.zq = reinterpret_cast<double2*>(p); p += (size_t)(f2 + 1) * 16;
  L.zr = reinterpret_cast<double2*>(p); p += (size_t)(f2 + 1) * 16;
  L.xp = reinterpret_cast<double*>(p); p += (size_t)(f2 + 2) * 8;
  L.b = reinterpret_cast<double*>(p); p += (size_t)(m + 2) * 8;
  L.pt = reinterpret_cast<double*>(p); p += (size_t)(m2 + 2) * 8;
  L.qt = reinterpret_cast<double*>(p); p += (size_t)(m2 + 4) * 8;
  L.rt = reinterpret_cast<double*>(p); p += (size_t)(m + 2) * 8;
  L.part = reinterpret_cast<double*>(p); p += (size_t)12 * (m2 + 1) * 8;
  L.A = reinterpret_cast<double*>(p); p += (size_t)m * (m + 2) * 8;
  L.fcol = reinterpret_cast<double*>(p); p += (size_t)(m + 2) * 8;
  L.misc = reinterpret_cast<double*>(p);
  L.tw = a.g_tw;                  // compact twiddles of an flng-point transform, read through the cache
  const int64_t fr = blockIdx.x;
  const int t = tid();
  const double* sp = a.amp + fr * (f2 + 1);
  for (int k = t; k <= f2; k += NT) {
    const double v = sp[k];
    L.xp[k] = (a.in_is_power ? v : v * v) + a.eps;
  }
  for (int j = t; j <= m; j += NT) L.b[j] = 0.0;
  if (t == 0) L.qt[m2 + 2] = 0.0;            // q~[2 + i + k] reaches index 2m at most
  __syncthreads();
  double ep = mg_newton(L, a, -1.0);
  const double g = a.gamma, al = a.alpha;
  int it = 0;
  if (g != -1.0) {
    if (t == 0) {
      double* b = L.b;
      double* d = L.fcol;
      if (al != 0.0) {
        mg_ignorm(b, m, -1.0);
        mg_b2mc(b, m, al);
        mg_gnorm(b, m, -1.0);
      }
      for (int i = 0; i <= m; ++i) d[i] = b[i];
      // gc2gc(d, m, -1 -> g)
      for (int i = 1; i <= m; ++i) {
        double ss1 = 0.0, ss2 = 0.0;
        for (int k = 1; k <= i - 1; ++k) {
          const int mk = i - k;
          const double cc = d[k] * b[mk];
          ss2 += k * cc;
          ss1 += mk * cc;
        }
        b[i] = d[i] + (g * ss2 + ss1) / i;          // g1 = -1
      }
      if (al != 0.0) {
        mg_ignorm(b, m, g);
        mg_mc2b(b, m, al);
        mg_gnorm(b, m, g);
      }
    }
    __syncthreads();
    for (it = 1; it <= a.itr2; ++it) {
      const double epo = ep;
      ep = mg_newton(L, a, g);
      if (it >= a.itr1 && fabs((epo - ep) / ep) < a.dd) break;     // uniform: same value everywhere
    }
  }
  if (t == 0) {
    mg_ignorm(L.b, m, g);
    if (al != 0.0) mg_b2mc(L.b, m, al);
  }
  __syncthreads();
  for (int j = t; j <= m; j += NT) {
    if (a.out_f32) a.out_f32[fr * a.ld_out + j] = (float)L.b[j];
    if (a.out_f64) a.out_f64[fr * (m + 1) + j] = L.b[j];
  }
  if (a.iters && t == 0) a.iters[fr] = it > a.itr2 ? a.itr2 : it;
}

// ---- mgc2sp for gamma != 0 ---------------------------------------------------------------------
// rows of the de-warped generalized cepstrum (freqt(mgc, -alpha) to order f2, by GEMM) ->
// gnorm(gamma), gc2gc(gamma -> 0), ignorm(0): one WAVE per frame; c2[i] needs every c2[k < i], the
// lanes share the sum over k.
__global__ __launch_bounds__(256) void mg_gc2gc_rows_kernel(double* __restrict__ cep, int64_t ld, int64_t T,
                                                            int f2, double g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int64_t fr = (int64_t)blockIdx.x * 4 + wv;
  if (fr >= T) return;
  double* cin = reinterpret_cast<double*>(smem) + (size_t)wv * 2 * (f2 + 1);
  double* c2 = cin + (f2 + 1);
  double* row = cep + fr * ld;
  const double k0 = 1.0 + g * row[0];
  for (int i = ln; i <= f2; i += 64) cin[i] = i == 0 ? pow(k0, 1.0 / g) : row[i] / k0;   // gnorm
  __builtin_amdgcn_wave_barrier();
  if (ln == 0) c2[0] = cin[0];
  for (int i = 1; i <= f2; ++i) {
    __builtin_amdgcn_wave_barrier();
    double ss1 = 0.0, ss2 = 0.0;
    for (int k = 1 + ln; k <= i - 1; k += 64) {
      const int mk = i - k;
      const double cc = cin[k] * c2[mk];
      ss2 += k * cc;
      ss1 += mk * cc;
    }
    ss1 = wave_sum(ss1);
    ss2 = wave_sum(ss2);
    if (ln == 0) c2[i] = cin[i] + (0.0 * ss2 - g * ss1) / i;      // g2 = 0, g1 = gamma
  }
  __builtin_amdgcn_wave_barrier();
  for (int i = ln; i <= f2; i += 64) row[i] = i == 0 ? log(c2[0]) : c2[i];                 // ignorm(0)
}

struct MgSpArgs {
  const double* cep;
  int64_t ld_cep, T;
  int fftlen, logfft;
  float* out_f32;
  double* out_f64;
  double* out_pow;
  const double2* g_tw;
};

__global__ __launch_bounds__(NT) void mg_c2sp_kernel(MgSpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double2* z = reinterpret_cast<double2*>(smem);
  double* zr = reinterpret_cast<double*>(z);
  const int64_t fr = blockIdx.x;
  const int f2 = a.fftlen / 2;
  const double* c = a.cep + fr * a.ld_cep;
  for (int i = tid(); i < a.fftlen + 2; i += NT) zr[i] = i <= f2 ? c[i] : 0.0;
  __syncthreads();
  rfft_lds(z, a.fftlen, a.logfft, a.g_tw, a.fftlen);
  for (int k = tid(); k <= f2; k += NT) {
    const double re = z[k].x;
    if (a.out_f64) a.out_f64[fr * (f2 + 1) + k] = re;
    const float amp = expf((float)re);
    if (a.out_f32) a.out_f32[fr * (f2 + 1) + k] = amp;
    if (a.out_pow) a.out_pow[fr * (f2 + 1) + k] = (double)amp * (double)amp;
  }
}

static int ilog2_h(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}

}  // namespace itts

using namespace itts;

extern "C" int itts_mgcep(const double* d_amp_sp, int input_is_power, int64_t T, int K, int order,
                          double alpha, double gamma, double eps, int miniter, int maxiter,
                          double threshold, float* d_mgc_f32, int64_t ld_mgc, double* d_mgc_f64,
                          int* d_iters, void* stream) {
  ITTS_REQUIRE(d_amp_sp && (d_mgc_f32 || d_mgc_f64), "null pointer");
  const int flng = (K - 1) * 2;
  ITTS_REQUIRE(T >= 0 && flng >= 64 && flng <= 2048 && (flng & (flng - 1)) == 0, "K must be 2^k/2+1, K <= 1025");
  ITTS_REQUIRE(order >= 1 && order < flng / 2 && order <= 63, "mgc order must be in [1, 63]");
  ITTS_REQUIRE(gamma <= 0.0 && gamma >= -1.0, "gamma must be in [-1, 0]");
  ITTS_REQUIRE(!d_mgc_f32 || ld_mgc >= order + 1, "ld_mgc too small");
  ITTS_REQUIRE(fabs(alpha) < 1.0 && maxiter >= 1, "bad alpha / maxiter");
  if (T == 0) return ITTS_OK;
  DeviceContext* ctx = get_context();
  if (!ctx) return ITTS_E_HIP;
  const FreqtTables* ft = get_freqt(ctx, order, flng / 2, alpha, false, true);
  if (!ft) return ITTS_E_HIP;
  MgArgs a{};
  a.amp = d_amp_sp; a.in_is_power = input_is_power; a.T = T; a.flng = flng; a.logflng = ilog2_h(flng); a.m = order; a.alpha = alpha;
  a.gamma = gamma; a.eps = eps; a.dd = threshold; a.itr1 = miniter; a.itr2 = maxiter;
  a.b1T = ft->b1T; a.p2T = ft->p2T; a.out_f32 = d_mgc_f32; a.ld_out = ld_mgc; a.out_f64 = d_mgc_f64;
  a.iters = d_iters; a.g_tw = ctx->tw_compact[a.logflng];
  const int f2 = flng / 2, m = order, m2 = 2 * m;
  const size_t lds = (size_t)3 * (f2 + 1) * 16 + (size_t)(f2 + 2) * 8 + (size_t)(m + 2) * 8 * 3 +
                     (size_t)(m2 + 2) * 8 + (size_t)(m2 + 4) * 8 + (size_t)12 * (m2 + 1) * 8 +
                     (size_t)m * (m + 2) * 8 + 64;
  ITTS_REQUIRE(lds <= 160 * 1024, "LDS budget exceeded");
  ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)mgcep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
  hipLaunchKernelGGL(mgcep_kernel, dim3((unsigned)T), dim3(NT), lds, as_stream(stream), a);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_mgc2sp_gamma(const double* d_mgc, int64_t T, int order, double alpha, double gamma,
                                 int fftlen, float* d_amp_f32, double* d_logamp_f64, double* d_pow_f64,
                                 void* stream) {
  ITTS_REQUIRE(d_mgc && (d_amp_f32 || d_logamp_f64 || d_pow_f64), "null pointer");
  ITTS_REQUIRE(T >= 0 && fftlen >= 64 && fftlen <= 4096 && (fftlen & (fftlen - 1)) == 0, "bad fftlen");
  ITTS_REQUIRE(order >= 0 && order <= fftlen / 2 && order <= 1023, "bad order");
  ITTS_REQUIRE(gamma <= 0.0 && gamma >= -1.0, "gamma must be in [-1, 0]");
  if (gamma == 0.0)
    return itts_mgc2sp(d_mgc, T, order, alpha, fftlen, d_amp_f32, d_logamp_f64, d_pow_f64, stream);
  if (T == 0) return ITTS_OK;
  DeviceContext* ctx = get_context();
  if (!ctx) return ITTS_E_HIP;
  const FreqtTables* ft = get_freqt(ctx, order, fftlen / 2, alpha, false);
  if (!ft) return ITTS_E_HIP;
  hipStream_t s = as_stream(stream);
  itts::ScratchScope scratch_scope(s);
  const int K = fftlen / 2 + 1;
  const int64_t ld_cep = (K + 1) & ~1;
  double* d_cep = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_cep, (size_t)T * ld_cep * 8, s));
  // freqt(mgc, m -> f2, -alpha) of all frames: one fp64-MFMA GEMM against the cached warping matrix
  int rc = launch_gemm_f64(d_mgc, order + 1, ft->invT, K, d_cep, ld_cep, T, K, order + 1, nullptr, s,
                           /*a_has_slack=*/false);
  if (rc) return rc;
  const size_t lds1 = (size_t)4 * 2 * K * 8;
  ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)mg_gc2gc_rows_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  hipLaunchKernelGGL(mg_gc2gc_rows_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), lds1, s, d_cep, ld_cep,
                     T, fftlen / 2, gamma);
  ITTS_LAUNCH_CHECK();
  const int logfft = ilog2_h(fftlen);
  MgSpArgs a{d_cep, ld_cep, T, fftlen, logfft, d_amp_f32, d_logamp_f64, d_pow_f64, ctx->tw_compact[logfft]};
  const size_t lds2 = (size_t)(fftlen / 2 + 1) * 16;
  hipLaunchKernelGGL(mg_c2sp_kernel, dim3((unsigned)T), dim3(NT), lds2, s, a);
  ITTS_LAUNCH_CHECK();
  ITTS_HIP_CHECK(itts::scratch_free(d_cep, s));
  return ITTS_OK;
}
