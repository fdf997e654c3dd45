// pt_portable_math.h — deterministic sin/cos/acos used by the HIP shade kernel.
//
// Why this exists: the reference's shading (src/pathtrace.cu:225-238, 404-415)
// calls sin/cos/acos from whatever libm the compiler provides (CUDA libdevice on
// the GPU, glibc on the host).  Those differ in the last ulp between vendors, so
// no two builds of the reference agree bit-for-bit.  To make "GPU == CPU oracle"
// a bit-exact statement, both sides evaluate these three functions with the SAME
// sequence of IEEE-754 operations written out here: for sin/cos double-precision argument
// reduction + fixed polynomials (the classic fdlibm kernel coefficients), for acos the
// fdlibm float algorithm; only +,-,*,/,sqrt,rint and explicit fma().  Every one of those is correctly rounded
// on gfx950 (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt, f64 IEEE)
// and on x86-64, so the results are identical on both.
//
// Accuracy: sin/cos are evaluated in double (within ~1 ulp(double) of the true value), so
// their float-returning wrappers are correctly rounded except for ~1e-8 of inputs; acosf is
// the fdlibm float algorithm (< 1 ulp, bit-identical to glibc's acosf).  Either way the
// results are at least as close to CUDA's / glibc's sinf/cosf/acosf as those are to each
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

// acosf: the classic fdlibm single-precision algorithm (rational approximation of asin on
// [0, 0.5], half-angle identities, one float sqrt and one float division), written out with plain
// IEEE float operations.  glibc's acosf is this same algorithm, and the sequence below reproduces
// glibc 2.35's results bit-for-bit for every float in [0, 1] (1,065,353,217 inputs checked) and for
// sampled negatives, so PORTABLE and LIBM modes of the oracle agree exactly on acos; it is also ~4x
// cheaper on the GPU than a double-precision evaluation (f64 sqrt + f64 divide).  Max error < 1 ulp.
PT_HD float acosf32(float x) {
  const float one = 1.0000000000e+00f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f,
              pio2_lo = 7.5497894159e-08f, pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f,
              pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
              pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
              qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
  const int hx = __builtin_bit_cast(int, x);
  const int ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;  // |x| == 1
  if (ix > 0x3f800000) return (x - x) / (x - x);                     // |x| > 1: NaN
  if (ix < 0x3f000000) {                                             // |x| < 0.5
    if (ix <= 0x32800000) return pio2_hi + pio2_lo;                  // |x| <= 2^-26
    const float z = x * x;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (hx < 0) {  // x < -0.5
    const float z = (one + x) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float s = __builtin_sqrtf(z);
    const float r = p / q;
    const float w = r * s - pio2_lo;
    return pi - 2.0f * (s + w);
  }
  // x > 0.5
  const float z = (one - x) * 0.5f;
  const float s = __builtin_sqrtf(z);
  const float df = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s) & 0xfffff000u);
  const float c = (z - df * df) / (s + df);
  const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const float r = p / q;
  const float w = r * s + c;
  return 2.0f * (df + w);
}

// float wrappers (what the renderer calls where the reference calls the float
// overloads of sin/cos/acos).
PT_HD float sinf32(float x) { return (float)sin64((double)x); }
PT_HD float cosf32(float x) { return (float)cos64((double)x); }
PT_HD void sincosf32(float x, float* s, float* c) {
  double ds, dc;
  sincos64((double)x, &ds, &dc);
  *s = (float)ds;
  *c = (float)dc;
}

}  // namespace ptmath
