// pt_portable_math.h — deterministic sin/cos/acos used by the HIP shade kernel.
//
// Why this exists: the reference's shading (src/pathtrace.cu:225-238, 404-415)
// calls sin/cos/acos from whatever libm the compiler provides (CUDA libdevice on
// the GPU, glibc on the host).  Those differ in the last ulp between vendors, so
// no two builds of the reference agree bit-for-bit.  To make "GPU == CPU oracle"
// a bit-exact statement, both sides evaluate these three functions with the SAME
// sequence of IEEE-754 operations written out here: double-precision argument
// reduction + fixed polynomials (the classic fdlibm kernel coefficients), only
// +,-,*,/,sqrt,rint and explicit fma().  Every one of those is correctly rounded
// on gfx950 (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt, f64 IEEE)
// and on x86-64, so the results are identical on both.
//
// Accuracy: the double result is within ~1 ulp(double) of the true value, so the
// float-returning wrappers are correctly rounded except for ~1e-8 of inputs —
// i.e. at least as close to CUDA's / glibc's sinf/cosf/acosf as those are to each
// other.  tests/test_portable_math.py measures this against libm.
//
// The file is plain C++ with no dependencies; `PT_HD` expands to
// `__host__ __device__` under hipcc.
#pragma once

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ inline
#else
#define PT_HD inline
#endif

namespace ptmath {

// fused multiply-add, exact on both sides (v_fma_f64 / libm or vfmadd).
PT_HD double fma64(double a, double b, double c) { return __builtin_fma(a, b, c); }

// sin(r), cos(r) for |r| <= pi/4 (+ a hair).  Horner in z = r*r.
PT_HD double ksin(double r) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = r * r;
  double p = fma64(z, S6, S5);
  p = fma64(z, p, S4);
  p = fma64(z, p, S3);
  p = fma64(z, p, S2);
  p = fma64(z, p, S1);
  return fma64(r * z, p, r);
}

PT_HD double kcos(double r) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = r * r;
  double p = fma64(z, C6, C5);
  p = fma64(z, p, C4);
  p = fma64(z, p, C3);
  p = fma64(z, p, C2);
  p = fma64(z, p, C1);
  // 1 - z/2 + z*z*p
  return fma64(z * z, p, fma64(z, -0.5, 1.0));
}

// Reduce x to r in [-pi/4, pi/4] and quadrant q (0..3).  Valid for |x| < ~1e6
// (the renderer only passes [0, 2*pi]).
PT_HD double reduce_pio2(double x, int* q) {
  const double INV_PIO2 = 6.36619772367581382433e-01;
  const double PIO2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
  const double PIO2_1T = 6.07710050650619224932e-11;  // pi/2 - PIO2_1
  double k = __builtin_rint(x * INV_PIO2);
  double r = fma64(k, -PIO2_1, x);  // exact: k*PIO2_1 has <= 53 bits
  r = fma64(k, -PIO2_1T, r);
  *q = ((int)k) & 3;
  return r;
}

PT_HD double sin64(double x) {
  int q;
  double r = reduce_pio2(x, &q);
  double s = ksin(r), c = kcos(r);
  double v = (q & 1) ? c : s;
  return (q & 2) ? -v : v;
}

PT_HD double cos64(double x) {
  int q;
  double r = reduce_pio2(x, &q);
  double s = ksin(r), c = kcos(r);
  double v = (q & 1) ? s : c;
  return ((q + 1) & 2) ? -v : v;
}

// sin and cos of the same argument with one reduction and one pair of kernels; bit-identical to
// calling sin64 and cos64 separately (same operations on the same values).
PT_HD void sincos64(double x, double* s_out, double* c_out) {
  int q;
  double r = reduce_pio2(x, &q);
  double s = ksin(r), c = kcos(r);
  double vs = (q & 1) ? c : s;
  double vc = (q & 1) ? s : c;
  *s_out = (q & 2) ? -vs : vs;
  *c_out = ((q + 1) & 2) ? -vc : vc;
}

// acos(x) for x in [-1, 1]; NaN outside.  fdlibm rational approximation of
// asin on [0, 0.5] plus the half-angle identities.
PT_HD double acos_rational(double z) {
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
               pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
               qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  double p = fma64(z, pS5, pS4);
  p = fma64(z, p, pS3);
  p = fma64(z, p, pS2);
  p = fma64(z, p, pS1);
  p = fma64(z, p, pS0);
  p = p * z;
  double q = fma64(z, qS4, qS3);
  q = fma64(z, q, qS2);
  q = fma64(z, q, qS1);
  q = fma64(z, q, 1.0);
  return p / q;
}

PT_HD double acos64(double x) {
  const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
  const double PI = 3.14159265358979311600e+00;
  double ax = x < 0.0 ? -x : x;
  if (!(ax <= 1.0)) return __builtin_nan("");
  if (ax < 0.5) {
    double r = acos_rational(x * x);
    return PIO2_HI - (x - (PIO2_LO - x * r));
  }
  double z = (1.0 - ax) * 0.5;
  double s = __builtin_sqrt(z);
  double r = acos_rational(z);
  double w = fma64(r, s, s);  // s + s*r  = asin-ish half angle
  if (x < 0.0) return PI - 2.0 * (w - PIO2_LO);
  return 2.0 * w;
}

// float wrappers (what the renderer calls where the reference calls the float
// overloads of sin/cos/acos).
PT_HD float sinf32(float x) { return (float)sin64((double)x); }
PT_HD float cosf32(float x) { return (float)cos64((double)x); }
PT_HD float acosf32(float x) { return (float)acos64((double)x); }
PT_HD void sincosf32(float x, float* s, float* c) {
  double ds, dc;
  sincos64((double)x, &ds, &dc);
  *s = (float)ds;
  *c = (float)dc;
}

}  // namespace ptmath
