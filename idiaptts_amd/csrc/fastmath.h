// fp64 exp / log / sincos / cos for the per-bin loops of the WORLD and SPTK frame kernels.
//
// Why: these kernels are bound by VALU issue, and the library versions spend
// most of their instructions on cases the per-bin loops never see (subnormals, infinities, NaNs,
// arguments beyond 1e5 radians): exp 42, log 98, sincos 157 VALU instructions in this toolchain
// against 21 / 38 / 50 here.  Same algorithms as the classic libm (Cody-Waite reduction + the
// fdlibm kernels), so the error stays below 1 ulp on the domains stated per function
// (tests/test_fastmath.py checks them on the host against long double; the header compiles for
// both sides).  Anything outside a stated domain is the caller's job (the kernels keep the library
// call where a loop can see such values).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define ITTS_FM __host__ __device__ __forceinline__
#else
#define ITTS_FM static inline
#endif

namespace itts {
namespace fm {

// exp(x), |x| <= 700 (no overflow / underflow handling).
ITTS_FM double fexp(double x) {
  const double k = rint(x * 1.4426950408889634074);                 // x / ln 2
  double r = fma(-k, 6.93147180559945286227e-01, x);                // ln 2, high and low part
  r = fma(-k, 2.31904681384629955842e-17, r);
  // Taylor polynomial of degree 13 on |r| <= 0.347: remainder 4e-18
  double p = 1.6059043836821613e-10;                                // 1 / 13!
  p = fma(p, r, 2.08767569878680989792e-09);
  p = fma(p, r, 2.50521083854417187751e-08);
  p = fma(p, r, 2.75573192239858906526e-07);
  p = fma(p, r, 2.75573192239858906526e-06);
  p = fma(p, r, 2.48015873015873015873e-05);
  p = fma(p, r, 1.98412698412698412698e-04);
  p = fma(p, r, 1.38888888888888888889e-03);
  p = fma(p, r, 8.33333333333333333333e-03);
  p = fma(p, r, 4.16666666666666666667e-02);
  p = fma(p, r, 1.66666666666666666667e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}

// log(x), x positive and normal (2.3e-308 <= x < inf).
ITTS_FM double flog(double x) {
  int e;
  double m = frexp(x, &e);                       // [0.5, 1)
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;                           // [sqrt(1/2), sqrt(2))
  e = low ? e - 1 : e;
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double dk = (double)e;
  const double z = s * s, w = z * z;
  double t1 = fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01);
  t1 = fma(w, t1, 2.857142874366239149e-01);
  t1 = fma(w, t1, 6.666666666666735130e-01);      // Lg1 + w (Lg3 + w (Lg5 + w Lg7))
  double t2 = fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01);
  t2 = fma(w, t2, 3.999999999940941908e-01);      // Lg2 + w (Lg4 + w Lg6)
  const double R = fma(z, t1, w * t2);
  const double hfsq = 0.5 * f * f;
  return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}

// reduction to [-pi/4, pi/4]: x = n pi/2 + (y0 + y1), |x| <= 1e5 (two Cody-Waite steps of fdlibm's
// medium path, always taken: 118 bits of pi/2)
ITTS_FM int rem_pio2(double x, double& y0, double& y1) {
  const double fn = rint(x * 6.36619772367581382433e-01);
  double r = fma(-fn, 1.57079632673412561417e+00, x);               // pio2_1 (33 bits): exact
  const double t = r;
  double w = fn * 6.07710050630396597660e-11;                       // pio2_2 (33 bits)
  r = t - w;
  w = fma(fn, 2.02226624879595063154e-21, -((t - r) - w));          // pio2_2t
  y0 = r - w;
  y1 = (r - y0) - w;
  return (int)fn;
}

ITTS_FM double kernel_sin(double x, double y) {
  const double z = x * x;
  const double v = z * x;
  double r = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  r = fma(z, r, 2.75573137070700676789e-06);
  r = fma(z, r, -1.98412698298579493134e-04);
  r = fma(z, r, 8.33333333332248946124e-03);
  return x - ((z * (0.5 * y - v * r) - y) - v * -1.66666666666666324348e-01);
}

ITTS_FM double kernel_cos(double x, double y) {
  const double z = x * x;
  double r = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  r = fma(z, r, -2.75573143513906633035e-07);
  r = fma(z, r, 2.48015872894767294178e-05);
  r = fma(z, r, -1.38888888888741095749e-03);
  r = fma(z, r, 4.16666666666666019037e-02);
  r = z * r;
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}

// sin and cos of x, |x| <= 1e5
ITTS_FM void fsincos(double x, double* sn, double* cs) {
  double y0, y1;
  const int n = rem_pio2(x, y0, y1);
  const double s = kernel_sin(y0, y1), c = kernel_cos(y0, y1);
  const double a = (n & 1) ? c : s, b = (n & 1) ? s : c;
  *sn = (n & 2) ? -a : a;
  *cs = ((n + 1) & 2) ? -b : b;
}

ITTS_FM double fcos(double x) {
  double y0, y1;
  const int n = rem_pio2(x, y0, y1);
  const double v = (n & 1) ? kernel_sin(y0, y1) : kernel_cos(y0, y1);
  return ((n + 1) & 2) ? -v : v;
}

ITTS_FM double fsin(double x) {
  double y0, y1;
  const int n = rem_pio2(x, y0, y1);
  const double v = (n & 1) ? kernel_cos(y0, y1) : kernel_sin(y0, y1);
  return (n & 2) ? -v : v;
}

}  // namespace fm
}  // namespace itts
