// pt_portable_math.h — deterministic sin/cos/acos used by the HIP shade kernel.
//
// Why this exists: the reference's shading (src/pathtrace.cu:225-238, 404-415)
// calls sin/cos/acos from whatever libm the compiler provides (CUDA libdevice on
// the GPU, glibc on the host).  Those differ in the last ulp between vendors, so
// no two builds of the reference agree bit-for-bit.  To make "GPU == CPU oracle"
// a bit-exact statement, both sides evaluate these three functions with the SAME
// sequence of IEEE-754 operations written out here: for sin/cos a double-precision argument
// reduction + fixed float polynomials, for acos the fdlibm float algorithm; only
// +,-,*,/,sqrt,rint and explicit fma().  Every one of those is correctly rounded
// on gfx950 (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt, f64 IEEE)
// and on x86-64, so the results are identical on both.
//
// Accuracy: sin/cos use a double-precision argument reduction and float polynomial kernels
// (<= 1 ulp, 2 ulp for 1.5e-5 of cos arguments); acosf is the fdlibm float algorithm (< 1 ulp,
// bit-identical to glibc's acosf).  That is the accuracy class of the functions the reference
// itself calls (CUDA documents 2 ulp for sinf/cosf, glibc's are < 1 ulp), so the results are as
// close to a CUDA or a glibc build of the reference as those two are to each other.
// tests/test_portable_math.py measures this against libm.
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

// fused multiply-adds, exact on both sides (v_fma_f64 / v_fma_f32; libm or vfmadd on the host).
PT_HD double fma64(double a, double b, double c) { return __builtin_fma(a, b, c); }
PT_HD float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// sin and cos of x (|x| < ~1e6; the renderer passes [0, 2*pi]), x given in double because the
// reference's specular branch forms its arguments in double (2.0f*M_PI*u, pathtrace.cu:411-413).
//   1. argument reduction in double: k = rint(x*2/pi), r = x - k*pi/2 (two-term Cody-Waite, the
//      product k*PIO2_1 is exact) — four f64 operations, so r is accurate to ~1e-17 even for the
//      double arguments and no precision is lost near multiples of pi/2;
//   2. r is rounded to float once, and sin(r), cos(r) on [-pi/4, pi/4] are evaluated in float with
//      fma Horner chains (near-minimax degree-9 / degree-8 polynomials, Chebyshev fit of the Taylor
//      series): float polynomials cost a quarter of the double-precision ones on gfx950's VALU.
// Accuracy against the correctly rounded float value, measured over all 2^24 arguments 2*pi*i/2^24 and
// over [0, pi/2]: sin <= 1 ulp everywhere (78.7 % exact), cos <= 1 ulp except 248 arguments with 2 ulp
// (73.2 % exact).  CUDA documents 2 ulp for its sinf/cosf and glibc's are within 1 ulp, so this is the
// same accuracy class as what the reference itself runs on.
PT_HD void sincos_r(double x, float* s_out, float* c_out) {
  const double INV_PIO2 = 6.36619772367581382433e-01;
  const double PIO2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
  const double PIO2_1T = 6.07710050650619224932e-11;  // pi/2 - PIO2_1
  const double k = __builtin_rint(x * INV_PIO2);
  double rd = fma64(k, -PIO2_1, x);  // exact: k*PIO2_1 has <= 53 bits
  rd = fma64(k, -PIO2_1T, rd);
  const int q = ((int)k) & 3;
  const float r = (float)rd, z = r * r;
  const float S1 = -1.666666666e-01f, S2 = 8.333331871e-03f, S3 = -1.984008473e-04f, S4 = 2.724965793e-06f;
  const float C1 = 4.166666666e-02f, C2 = -1.388888767e-03f, C3 = 2.480059866e-05f, C4 = -2.730073108e-07f;
  float ps = fma32(z, S4, S3);
  ps = fma32(z, ps, S2);
  ps = fma32(z, ps, S1);
  const float s = fma32(r * z, ps, r);                          // r + r*z*(S1 + z*(S2 + z*(S3 + z*S4)))
  float pc = fma32(z, C4, C3);
  pc = fma32(z, pc, C2);
  pc = fma32(z, pc, C1);
  const float c = fma32(z * z, pc, fma32(z, -0.5f, 1.0f));      // 1 - z/2 + z*z*(C1 + z*(C2 + z*(C3 + z*C4)))
  const float vs = (q & 1) ? c : s;
  const float vc = (q & 1) ? s : c;
  *s_out = (q & 2) ? -vs : vs;
  *c_out = ((q + 1) & 2) ? -vc : vc;
}
PT_HD float sin_r(double x) {
  float s, c;
  sincos_r(x, &s, &c);
  return s;
}
PT_HD float cos_r(double x) {
  float s, c;
  sincos_r(x, &s, &c);
  return c;
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

// sin(2 pi u) and cos(2 pi u) for u in [0, 1], float operations only — the fma arithmetic mode's direction sampling
// (pt_kernels.hip shade_bounce_float).  The argument is taken in REVOLUTIONS, so the reduction is exact: k = rint(4u),
// f = u - k/4 (one FMA, exact), |f| <= 1/8; r = 2 pi f is formed as a head + tail product (float(2 pi) and its remainder,
// the head product's rounding recovered with an FMA), rounded once, and the tail of that sum corrects the results to first
// order.  Same polynomials as sincos_r.  Measured against double-precision sin / cos over all 2^24 + 1 arguments i / 2^24
// and 10^7 random ones (tests/test_portable_math.py): <= 1.5 ulp, i.e. inside what CUDA documents for the sinf / cosf
// the reference itself runs on (2 ulp) and tighter than evaluating sinf on a 2 pi u that was rounded to float first.
PT_HD void sincos_rev(float u, float* s_out, float* c_out) {
  const float TWO_PI_HI = 6.2831854820251465f;   // float(2 pi)
  const float TWO_PI_LO = -1.7484555e-07f;       // 2 pi - float(2 pi)
  const float k = __builtin_rintf(4.0f * u);
  const float f = fma32(k, -0.25f, u);
  const float rh = f * TWO_PI_HI;
  const float rl = fma32(f, TWO_PI_LO, fma32(f, TWO_PI_HI, -rh));  // what rh lost + the constant's tail
  const float r = rh + rl;
  const float rt = rl - (r - rh);  // tail of the sum (|rt| <= ulp(r) / 2)
  const int q = ((int)k) & 3;
  const float z = r * r;
  const float S1 = -1.666666666e-01f, S2 = 8.333331871e-03f, S3 = -1.984008473e-04f, S4 = 2.724965793e-06f;
  const float C1 = 4.166666666e-02f, C2 = -1.388888767e-03f, C3 = 2.480059866e-05f, C4 = -2.730073108e-07f;
  float ps = fma32(z, S4, S3);
  ps = fma32(z, ps, S2);
  ps = fma32(z, ps, S1);
  float pc = fma32(z, C4, C3);
  pc = fma32(z, pc, C2);
  pc = fma32(z, pc, C1);
  const float c0 = fma32(z * z, pc, fma32(z, -0.5f, 1.0f));
  const float s0 = fma32(r * z, ps, r);
  const float s = fma32(rt, c0, s0);  // sin(r + rt) = sin r + rt cos r
  const float c = fma32(-rt, s0, c0);
  const float vs = (q & 1) ? c : s;
  const float vc = (q & 1) ? s : c;
  *s_out = (q & 2) ? -vs : vs;
  *c_out = ((q + 1) & 2) ? -vc : vc;
}

// float wrappers (what the renderer calls where the reference calls the float overloads)
PT_HD float sinf32(float x) { return sin_r((double)x); }
PT_HD float cosf32(float x) { return cos_r((double)x); }
PT_HD void sincosf32(float x, float* s, float* c) { sincos_r((double)x, s, c); }

}  // namespace ptmath
