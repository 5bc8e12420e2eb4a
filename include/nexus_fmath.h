/*
 * nexus_fmath.h — the transcendental functions of the shading path, ONE text compiled on both sides.
 *
 * Why this file exists.  The reference's shading code calls the platform's libm: double sin / cos in RandomCosineHemisphere
 * (/root/reference/Nexus/src/Cuda/Random.cuh:119-121), expf / logf / sinf / cosf in the Beckmann helpers
 * (Cuda/BSDF/Microfacet.cuh:18, 75), atan2f / asinf in SampleBackground (Cuda/PathTracer/PathTracer.cu:65-83) and a double pow in
 * LinearToGamma (Utils/Utils.h:51-54).  On the device those come from ROCm's ocml, in the CPU oracle from glibc; the two agree to
 * an ulp or two, one flipped Russian-roulette / lobe decision changes a pixel, and frames could only be compared as "x % of the
 * pixels within 1e-3".  The functions below are used by BOTH the HIP kernels (nx_rng.h, nx_bsdf.h, nx_wavefront.hip) and the CPU
 * oracle (orc_shade.c, orc_wavefront.c): the same sequence of IEEE-754 operations on either side, so a frame of the device
 * equals the oracle's bit for bit.
 *
 * Rules of this text (what makes it bit-reproducible across gcc / x86-64 and hipcc / gfx950):
 *   - only + - * / sqrt, fma, rint, floor, fabs, comparisons, conversions and bit casts — each correctly rounded and therefore
 *     uniquely defined by IEEE 754; no libm call, no table;
 *   - the double functions (sin / cos of RandomCosineHemisphere, pow of LinearToGamma) are within a few ulp of binary64; the
 *     float functions (second half of this file, round 5) are binary32 arithmetic within 2 ulp of the true value over every
 *     binary32 argument — what CUDA's own sinf / cosf / expf / atan2f / asinf promise the reference (tests/test_fmath.py
 *     measures both against a 50-digit reference, tools/fmath_exhaustive.c the float ones exhaustively);
 *   - both translation units are compiled with floating-point contraction off (Makefile, oracle/Makefile): an fma happens where
 *     nxf_fma is written and nowhere else;
 *   - the double functions: series in nested form with small exact divisors; the float functions: minimax polynomials whose
 *     coefficients tools/fmath_coeffs.py derives — no table transcribed from anywhere.
 *
 * Domain notes: the trigonometric reduction is two-term Cody-Waite, exact to ~1e-16 * |x| (the path's arguments lie in
 * [-2 pi, 2 pi]); beyond 2^30 the result is still the same on both sides, and within [-1, 1], but no longer accurate.
 */
#ifndef NEXUS_FMATH_H
#define NEXUS_FMATH_H

#include <stdint.h>

#if defined(__HIP__) /* clang in HIP mode (the .hip translation units): callable from kernels and from host code */
#define NXF_FN static __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define NXF_FN static inline
#endif

#define NXF_PI 3.14159265358979323846
#define NXF_PIO2_HI 1.57079632679489655800e+00 /* the binary64 nearest to pi / 2 */
#define NXF_PIO2_LO 6.12323399573676603587e-17 /* pi / 2 - NXF_PIO2_HI */
#define NXF_PIO4 7.85398163397448309616e-01
#define NXF_LN2_HI 6.93147180369123816490e-01  /* ln 2 with the low 21 bits of the significand cleared: k * LN2_HI is exact for |k| < 2^20 */
#define NXF_LN2_LO 1.90821492927058770002e-10  /* ln 2 - NXF_LN2_HI */
#define NXF_LOG2E 1.44269504088896338700e+00
#define NXF_SQRT2 1.41421356237309514547e+00
#define NXF_TAN_PIO8 4.14213562373095034e-01

NXF_FN double nxf_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
NXF_FN double nxf_rint(double x) { return __builtin_rint(x); }
NXF_FN double nxf_floor(double x) { return __builtin_floor(x); }
NXF_FN double nxf_sqrt(double x) { return __builtin_sqrt(x); }
NXF_FN double nxf_abs(double x) { return __builtin_fabs(x); }
NXF_FN uint64_t nxf_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
NXF_FN double nxf_from_bits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
NXF_FN uint32_t nxf_bitsf(float x) { uint32_t u; __builtin_memcpy(&u, &x, 4); return u; }
NXF_FN float nxf_from_bitsf(uint32_t u) { float x; __builtin_memcpy(&x, &u, 4); return x; }
NXF_FN double nxf_nan(void) { return nxf_from_bits(0x7ff8000000000000ull); }
NXF_FN double nxf_inf(void) { return nxf_from_bits(0x7ff0000000000000ull); }
NXF_FN int nxf_isnan(double x) { return x != x; }
NXF_FN int nxf_isinf(double x) { return nxf_abs(x) == nxf_inf(); }
NXF_FN int nxf_signbit(double x) { return (int)(nxf_bits(x) >> 63); }
NXF_FN double nxf_copysign(double mag, double sgn) { return nxf_from_bits((nxf_bits(mag) & 0x7fffffffffffffffull) | (nxf_bits(sgn) & 0x8000000000000000ull)); }
/* 2^k as a double, k in [-1022, 1023] */
NXF_FN double nxf_pow2i(int k) { return nxf_from_bits((uint64_t)(k + 1023) << 52); }

/* ---- sin / cos ------------------------------------------------------------------------------------------------------- */

/* sin r and cos r for |r| <= pi / 4 (+ a rounding error of the reduction): Taylor series in nested form,
 *   sin r = r (1 - z/(2*3) (1 - z/(4*5) (1 - ... ))),   cos r = 1 - z/(1*2) (1 - z/(3*4) (1 - ... )),   z = r^2;
 * truncated after z^9 / z^10: the first dropped term is below 1e-19 / 3e-21 at pi / 4. */
NXF_FN double nxf_sin_kernel(double r)
{
    const double z = r * r;
    double t = 1.0;
    t = 1.0 - (z * (1.0 / (18.0 * 19.0))) * t;
    t = 1.0 - (z * (1.0 / (16.0 * 17.0))) * t;
    t = 1.0 - (z * (1.0 / (14.0 * 15.0))) * t;
    t = 1.0 - (z * (1.0 / (12.0 * 13.0))) * t;
    t = 1.0 - (z * (1.0 / (10.0 * 11.0))) * t;
    t = 1.0 - (z * (1.0 / (8.0 * 9.0))) * t;
    t = 1.0 - (z * (1.0 / (6.0 * 7.0))) * t;
    t = 1.0 - (z * (1.0 / (4.0 * 5.0))) * t;
    /* the last step as r - r * (z / 6 * t): the correction is added to r itself, which keeps sin r accurate relative to r */
    return r - r * ((z * (1.0 / (2.0 * 3.0))) * t);
}
NXF_FN double nxf_cos_kernel(double r)
{
    const double z = r * r;
    double t = 1.0;
    t = 1.0 - (z * (1.0 / (19.0 * 20.0))) * t;
    t = 1.0 - (z * (1.0 / (17.0 * 18.0))) * t;
    t = 1.0 - (z * (1.0 / (15.0 * 16.0))) * t;
    t = 1.0 - (z * (1.0 / (13.0 * 14.0))) * t;
    t = 1.0 - (z * (1.0 / (11.0 * 12.0))) * t;
    t = 1.0 - (z * (1.0 / (9.0 * 10.0))) * t;
    t = 1.0 - (z * (1.0 / (7.0 * 8.0))) * t;
    t = 1.0 - (z * (1.0 / (5.0 * 6.0))) * t;
    t = 1.0 - (z * (1.0 / (3.0 * 4.0))) * t;
    return 1.0 - (z * 0.5) * t;
}

/* x = k * pi/2 + r, |r| <= pi/4: *quadrant = k mod 4 */
NXF_FN double nxf_reduce_pio2(double x, int *quadrant)
{
    const double k = nxf_rint(x * (2.0 / NXF_PI));
    double r = nxf_fma(-k, NXF_PIO2_HI, x);
    r = nxf_fma(-k, NXF_PIO2_LO, r);
    if (!(nxf_abs(r) <= 0.7854)) r = 0.0;  /* |x| beyond ~2^50: k no longer resolves quarter turns; keep the result bounded */
    /* k mod 4 in double arithmetic (exact for every finite k), so no integer conversion of a huge value */
    *quadrant = (int)(k - 4.0 * nxf_floor(k * 0.25));
    return r;
}

NXF_FN void nxf_sincos(double x, double *s, double *c)
{
    if (nxf_isnan(x) || nxf_isinf(x)) { *s = nxf_nan(); *c = nxf_nan(); return; }
    if (x == 0.0) { *s = x; *c = 1.0; return; }  /* sin(-0) = -0 */
    int q;
    const double r = nxf_reduce_pio2(x, &q);
    const double sr = nxf_sin_kernel(r), cr = nxf_cos_kernel(r);
    *s = (q & 1) ? cr : sr;
    *c = (q & 1) ? sr : cr;
    if (q & 2) *s = -*s;
    if ((q + 1) & 2) *c = -*c;
}
NXF_FN double nxf_sin(double x) { double s, c; nxf_sincos(x, &s, &c); return s; }
NXF_FN double nxf_cos(double x) { double s, c; nxf_sincos(x, &s, &c); return c; }

/* ---- exp / log / pow ------------------------------------------------------------------------------------------------- */

/* e^x in double.  x = k ln 2 + r, |r| <= ln 2 / 2; e^r = 1 + r (1 + r/2 (1 + r/3 ( ... ))) through r^14 / 14!
 * (first dropped term: 0.35^15 / 15! = 1e-19); scaled by 2^k in two exact steps so that results in the subnormal range are
 * reached by one final rounding multiplication. */
NXF_FN double nxf_exp(double x)
{
    if (nxf_isnan(x)) return nxf_nan();
    if (x > 709.79) return nxf_inf();
    if (x < -745.14) return 0.0;
    const double k = nxf_rint(x * NXF_LOG2E);
    double r = nxf_fma(-k, NXF_LN2_HI, x);
    r = nxf_fma(-k, NXF_LN2_LO, r);
    double t = 1.0;
    t = 1.0 + (r * (1.0 / 14.0)) * t;
    t = 1.0 + (r * (1.0 / 13.0)) * t;
    t = 1.0 + (r * (1.0 / 12.0)) * t;
    t = 1.0 + (r * (1.0 / 11.0)) * t;
    t = 1.0 + (r * (1.0 / 10.0)) * t;
    t = 1.0 + (r * (1.0 / 9.0)) * t;
    t = 1.0 + (r * (1.0 / 8.0)) * t;
    t = 1.0 + (r * (1.0 / 7.0)) * t;
    t = 1.0 + (r * (1.0 / 6.0)) * t;
    t = 1.0 + (r * (1.0 / 5.0)) * t;
    t = 1.0 + (r * (1.0 / 4.0)) * t;
    t = 1.0 + (r * (1.0 / 3.0)) * t;
    t = 1.0 + (r * (1.0 / 2.0)) * t;
    t = 1.0 + r * t;
    const int ki = (int)k;           /* |k| <= 1075 */
    const int k1 = ki / 2, k2 = ki - k1;
    return (t * nxf_pow2i(k1)) * nxf_pow2i(k2);
}

/* ln x in double.  x = m 2^e with m in [sqrt(1/2), sqrt 2); ln m = 2 atanh s = 2 s (1 + z/3 + z^2/5 + ...), s = (m-1)/(m+1),
 * z = s^2 <= 0.0295, through z^11 / 23 (first dropped term 1e-20 relative). */
NXF_FN double nxf_log(double x)
{
    if (nxf_isnan(x) || x < 0.0) return nxf_nan();
    if (x == 0.0) return -nxf_inf();
    if (nxf_isinf(x)) return x;
    int e = 0;
    if (x < 2.2250738585072014e-308) { x *= 18014398509481984.0; e = -54; }  /* subnormal: scale by 2^54 */
    const uint64_t b = nxf_bits(x);
    e += (int)(b >> 52) - 1023;
    double m = nxf_from_bits((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > NXF_SQRT2) { m *= 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double t = 1.0 / 23.0;
    t = 1.0 / 21.0 + z * t;
    t = 1.0 / 19.0 + z * t;
    t = 1.0 / 17.0 + z * t;
    t = 1.0 / 15.0 + z * t;
    t = 1.0 / 13.0 + z * t;
    t = 1.0 / 11.0 + z * t;
    t = 1.0 / 9.0 + z * t;
    t = 1.0 / 7.0 + z * t;
    t = 1.0 / 5.0 + z * t;
    t = 1.0 / 3.0 + z * t;
    const double lnm = 2.0 * s + (2.0 * s) * (z * t);
    const double de = (double)e;
    return de * NXF_LN2_HI + (lnm + de * NXF_LN2_LO);
}

/* x^y for the cases the path has (a base >= 0, any finite exponent) as exp(y ln x); about 1e-14 relative for |y ln x| < 100 —
 * the callers round the result to float (LinearToGamma, Utils/Utils.h:51-54).  A negative base gives NaN (no integer-exponent
 * special cases: the path never raises a negative number). */
NXF_FN double nxf_pow(double x, double y)
{
    if (y == 0.0 || x == 1.0) return 1.0;  /* (also for a NaN in the other argument: C Annex F) */
    if (nxf_isnan(x) || nxf_isnan(y)) return nxf_nan();
    if (x < 0.0) return nxf_nan();
    if (x == 0.0) return y > 0.0 ? 0.0 : nxf_inf();
    if (nxf_isinf(x)) return y > 0.0 ? x : 0.0;
    return nxf_exp(y * nxf_log(x));
}

/* ---- atan2 / asin ---------------------------------------------------------------------------------------------------- */

/* atan a for a in [0, 1].  Above tan(pi/8): atan a = pi/4 + atan((a - 1) / (a + 1)), which leaves |t| <= tan(pi/8) = 0.4142;
 * atan t = t (1 - z/3 + z^2/5 - ...), z = t^2 <= 0.1716, through z^21 / 43 (first dropped term 3e-18 relative). */
NXF_FN double nxf_atan01(double a)
{
    double base = 0.0, t = a;
    if (a > NXF_TAN_PIO8) { t = (a - 1.0) / (a + 1.0); base = NXF_PIO4; }
    const double z = t * t;
    double p = 1.0 / 43.0;
    p = 1.0 / 41.0 - z * p;
    p = 1.0 / 39.0 - z * p;
    p = 1.0 / 37.0 - z * p;
    p = 1.0 / 35.0 - z * p;
    p = 1.0 / 33.0 - z * p;
    p = 1.0 / 31.0 - z * p;
    p = 1.0 / 29.0 - z * p;
    p = 1.0 / 27.0 - z * p;
    p = 1.0 / 25.0 - z * p;
    p = 1.0 / 23.0 - z * p;
    p = 1.0 / 21.0 - z * p;
    p = 1.0 / 19.0 - z * p;
    p = 1.0 / 17.0 - z * p;
    p = 1.0 / 15.0 - z * p;
    p = 1.0 / 13.0 - z * p;
    p = 1.0 / 11.0 - z * p;
    p = 1.0 / 9.0 - z * p;
    p = 1.0 / 7.0 - z * p;
    p = 1.0 / 5.0 - z * p;
    p = 1.0 / 3.0 - z * p;
    return base + (t - t * (z * p));
}

/* atan2(y, x) in double with the C standard's (Annex F.10.1.4) results for zeros and infinities */
NXF_FN double nxf_atan2(double y, double x)
{
    if (nxf_isnan(x) || nxf_isnan(y)) return nxf_nan();
    const double ax = nxf_abs(x), ay = nxf_abs(y);
    double r;
    if (ay == 0.0) r = 0.0;                                         /* +-0 or +-pi */
    else if (ax == 0.0) r = NXF_PIO2_HI;
    else if (nxf_isinf(ax) && nxf_isinf(ay)) r = NXF_PIO4;
    else if (nxf_isinf(ax)) r = 0.0;
    else if (nxf_isinf(ay)) r = NXF_PIO2_HI;
    else if (ay <= ax) r = nxf_atan01(ay / ax);
    else r = NXF_PIO2_HI - nxf_atan01(ax / ay);
    if (nxf_signbit(x)) r = NXF_PI - r;
    return nxf_copysign(r, y);
}

/* asin x = atan2(x, sqrt((1 - x)(1 + x))); NaN outside [-1, 1] */
NXF_FN double nxf_asin(double x)
{
    if (nxf_isnan(x) || nxf_abs(x) > 1.0) return nxf_nan();
    return nxf_atan2(x, nxf_sqrt((1.0 - x) * (1.0 + x)));
}

/* ---- the binary32 functions: sinf cosf expf logf atan2f asinf ---------------------------------------------------------- */

/* The reference calls these as float functions (Microfacet.cuh:18, 75; PathTracer.cu:65-83), whose CUDA implementations are good
 * to 1-3 ulp.  Until round 5 this text evaluated them through the binary64 series above and rounded once (0.5 ulp, at the price
 * of 20-term series and two divisions in binary64 per call: 3 % of configs[3], whose every miss and light sample looks the
 * environment up).  Now: binary32 arithmetic throughout — a reduction, one minimax polynomial of the series remainder evaluated with
 * fmaf, at most one division or square root — each within 2 ulp over EVERY binary32 argument of its domain
 * (tools/fmath_exhaustive.c: sinf / cosf 1.53 ulp for |x| <= 1e5, expf 1.01, logf 0.84, asinf 1.89, atan2f 1.53).  The coefficients are fitted, not transcribed:
 * tools/fmath_coeffs.py prints them.  The only binary64 left is the trigonometric reduction (four operations), because a float
 * pi / 2 in three pieces loses the small remainders near multiples of pi / 2.  Same rule as above for reproducibility: only
 * correctly rounded IEEE operations, fmaf where written and nowhere else. */
NXF_FN float nxf_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
NXF_FN float nxf_absf(float x) { return __builtin_fabsf(x); }
NXF_FN float nxf_nanf(void) { return nxf_from_bitsf(0x7fc00000u); }
NXF_FN float nxf_inff(void) { return nxf_from_bitsf(0x7f800000u); }
NXF_FN float nxf_copysignf(float mag, float sgn) { return nxf_from_bitsf((nxf_bitsf(mag) & 0x7fffffffu) | (nxf_bitsf(sgn) & 0x80000000u)); }
#define NXF_PIO2_HI_F 1.57079637f      /* the binary32 nearest to pi / 2 */
#define NXF_PIO2_LO_F -4.37113883e-08f /* pi / 2 - NXF_PIO2_HI_F */
#define NXF_PI_HI_F 3.14159274f
#define NXF_PI_LO_F -8.74227766e-08f
#define NXF_PIO4_F 0.785398185f
#define NXF_LN2_HI_F 0.693115234f      /* ln 2 with the low 12 bits of the significand cleared: k * hi is exact for |k| < 2^12 */
#define NXF_LN2_LO_F 3.19461833e-05f   /* ln 2 - NXF_LN2_HI_F */
#define NXF_LOG2E_F 1.44269502f

/* sin x and cos x.  x = k pi/2 + r in binary64 (nxf_reduce_pio2: exact to 1e-16 |x|), r rounded once to binary32, |r| <= pi/4;
 *   sin r = r + r z (S0 + S1 z + S2 z^2),   cos r = 1 - z/2 + z^2 (C0 + C1 z + C2 z^2),   z = r^2
 * (fit errors 3.8e-9 / 1.2e-10 relative); 1 - z/2 is rounded to w and what the rounding dropped, (1 - w) - z/2 — exact —, is
 * added back with the polynomial. */
NXF_FN void nxf_sincosf(float x, float *s, float *c)
{
    if (x != x || nxf_absf(x) == nxf_inff()) { *s = nxf_nanf(); *c = nxf_nanf(); return; }
    if (x == 0.0f) { *s = x; *c = 1.0f; return; }  /* sin(-0) = -0 */
    int q;
    const float r = (float)nxf_reduce_pio2((double)x, &q);
    const float z = r * r;
    float ps = nxf_fmaf(z, -0.000195152184f, 0.0083321603f);
    ps = nxf_fmaf(z, ps, -0.166666552f);
    const float sr = nxf_fmaf(r * z, ps, r);
    float pc = nxf_fmaf(z, 2.44330822e-05f, -0.00138873153f);
    pc = nxf_fmaf(z, pc, 0.0416666456f);
    const float hz = 0.5f * z, w = 1.0f - hz;
    const float cr = w + (((1.0f - w) - hz) + (z * z) * pc);
    *s = (q & 1) ? cr : sr;
    *c = (q & 1) ? sr : cr;
    if (q & 2) *s = -*s;
    if ((q + 1) & 2) *c = -*c;
}
NXF_FN float nxf_sinf(float x) { float s, c; nxf_sincosf(x, &s, &c); return s; }
NXF_FN float nxf_cosf(float x) { float s, c; nxf_sincosf(x, &s, &c); return c; }

/* e^x.  x = k ln 2 + r, |r| <= ln 2 / 2 (k * NXF_LN2_HI_F is exact); e^r = 1 + (r + r^2 (E0 + ... + E4 r^4)) (fit error 3.1e-9);
 * scaled by 2^k in two exact-or-final steps, so that a subnormal result is rounded once. */
NXF_FN float nxf_pow2if(int k) { return nxf_from_bitsf((uint32_t)(k + 127) << 23); } /* 2^k, k in [-126, 127] */
NXF_FN float nxf_expf(float x)
{
    if (x != x) return nxf_nanf();
    if (x > 89.0f) return nxf_inff();  /* (e^88.73 already overflows: the last multiplication below says so) */
    if (x < -104.0f) return 0.0f;      /* below 2^-150: rounds to zero */
    const float k = __builtin_rintf(x * NXF_LOG2E_F);
    float r = nxf_fmaf(-k, NXF_LN2_HI_F, x);
    r = nxf_fmaf(-k, NXF_LN2_LO_F, r);
    float p = nxf_fmaf(r, 0.00138145988f, 0.00836871658f);
    p = nxf_fmaf(r, p, 0.041668389f);
    p = nxf_fmaf(r, p, 0.166665211f);
    p = nxf_fmaf(r, p, 0.49999994f);
    const float t = 1.0f + nxf_fmaf(r * r, p, r);
    const int ki = (int)k;  /* |k| <= 150 */
    const int k1 = ki / 2, k2 = ki - k1;
    return (t * nxf_pow2if(k1)) * nxf_pow2if(k2);
}

/* ln x.  x = m 2^e, m in [sqrt(1/2), sqrt 2), f = m - 1; ln m = 2 atanh s = 2 s + 2 s z (L0 + L1 z + L2 z^2), s = f / (2 + f),
 * z = s^2 (fit error 8e-10); e * NXF_LN2_HI_F is exact. */
NXF_FN float nxf_logf(float x)
{
    if (x != x || x < 0.0f) return nxf_nanf();
    if (x == 0.0f) return -nxf_inff();
    if (x == nxf_inff()) return x;
    int e = 0;
    if (x < 1.17549435e-38f) { x *= 8388608.0f; e = -23; }  /* subnormal: scaled by 2^23 */
    const uint32_t b = nxf_bitsf(x);
    e += (int)(b >> 23) - 127;
    float m = nxf_from_bitsf((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m *= 0.5f; e += 1; }
    const float f = m - 1.0f;  /* exact */
    const float s = f / (2.0f + f);
    const float z = s * s;
    float p = nxf_fmaf(z, 0.149356037f, 0.199887827f);
    p = nxf_fmaf(z, p, 0.33333388f);
    /* 2 s = f - s f and s f = f^2/2 - s f^2/2, so ln m = f - (f^2/2 - s (f^2/2 + 2 z P)): the leading term is f itself, exact, and
     * the rounding of s only touches the small terms */
    const float hfsq = 0.5f * f * f;
    const float fe = (float)e;
    return fe * NXF_LN2_HI_F - ((hfsq - nxf_fmaf(s, hfsq + (z + z) * p, fe * NXF_LN2_LO_F)) - f);
}

/* atan a for a in [0, 1]: a + a z (A0 + ... + A8 z^8), z = a^2 (fit error 2.6e-9) */
NXF_FN float nxf_atan01f(float a)
{
    const float z = a * a;
    float p = nxf_fmaf(z, -0.00179362029f, 0.0109145995f);
    p = nxf_fmaf(z, p, -0.0311778337f);
    p = nxf_fmaf(z, p, 0.0579575822f);
    p = nxf_fmaf(z, p, -0.0840344951f);
    p = nxf_fmaf(z, p, 0.109521858f);
    p = nxf_fmaf(z, p, -0.142642424f);
    p = nxf_fmaf(z, p, 0.199985489f);
    p = nxf_fmaf(z, p, -0.333332986f);
    return nxf_fmaf(a * z, p, a);
}

/* atan2(y, x): ONE division, with the C standard's (Annex F.10.1.4) results for zeros and infinities */
NXF_FN float nxf_atan2f(float y, float x)
{
    if (x != x || y != y) return nxf_nanf();
    const float ax = nxf_absf(x), ay = nxf_absf(y), inf = nxf_inff();
    float r;
    if (ay == 0.0f) r = 0.0f;                                       /* +-0 or +-pi */
    else if (ax == 0.0f) r = NXF_PIO2_HI_F;
    else if (ax == inf && ay == inf) r = NXF_PIO4_F;
    else if (ax == inf) r = 0.0f;
    else if (ay == inf) r = NXF_PIO2_HI_F;
    else if (ay <= ax) r = nxf_atan01f(ay / ax);
    else r = (NXF_PIO2_HI_F - nxf_atan01f(ax / ay)) + NXF_PIO2_LO_F;
    if (nxf_bitsf(x) >> 31) r = (NXF_PI_HI_F - r) + NXF_PI_LO_F;
    return nxf_copysignf(r, y);
}

/* asin x.  |x| <= 1/2: x + x z (R0 + ... + R4 z^4), z = x^2 (fit error 4.9e-9); above: pi/2 - 2 asin(sqrt((1 - |x|) / 2)), one
 * square root; NaN outside [-1, 1] */
NXF_FN float nxf_asin_poly(float z)
{
    float p = nxf_fmaf(z, 0.0421663076f, 0.0241795164f);
    p = nxf_fmaf(z, p, 0.0454703756f);
    p = nxf_fmaf(z, p, 0.0749529749f);
    return nxf_fmaf(z, p, 0.166667521f);
}
NXF_FN float nxf_asinf(float x)
{
    if (x != x || nxf_absf(x) > 1.0f) return nxf_nanf();
    const float ax = nxf_absf(x);
    float r;
    if (ax <= 0.5f) {
        const float z = ax * ax;
        r = nxf_fmaf(ax * z, nxf_asin_poly(z), ax);
    } else {
        const float z = (1.0f - ax) * 0.5f;  /* exact */
        const float s = __builtin_sqrtf(z);
        const float t = nxf_fmaf(s * z, nxf_asin_poly(z), s);
        r = (NXF_PIO2_HI_F - (t + t)) + NXF_PIO2_LO_F;
    }
    return nxf_copysignf(r, x);
}

/* ---- batch entry for the tests (oracle: orc_fmath_batch; device: nxhip_fmath_batch) ------------------------------------ */

enum {
    NXF_OP_SIN = 0,    /* double in, double out */
    NXF_OP_COS = 1,
    NXF_OP_EXP = 2,
    NXF_OP_LOG = 3,
    NXF_OP_POW = 4,    /* a^b */
    NXF_OP_ATAN2 = 5,  /* atan2(a, b) */
    NXF_OP_ASIN = 6,
    NXF_OP_SINF = 7,   /* the float functions: arguments and result are floats carried in doubles */
    NXF_OP_COSF = 8,
    NXF_OP_EXPF = 9,
    NXF_OP_LOGF = 10,
    NXF_OP_ATAN2F = 11,
    NXF_OP_ASINF = 12,
    NXF_OP_COUNT = 13
};
NXF_FN double nxf_apply(int op, double a, double b)
{
    switch (op) {
    case NXF_OP_SIN: return nxf_sin(a);
    case NXF_OP_COS: return nxf_cos(a);
    case NXF_OP_EXP: return nxf_exp(a);
    case NXF_OP_LOG: return nxf_log(a);
    case NXF_OP_POW: return nxf_pow(a, b);
    case NXF_OP_ATAN2: return nxf_atan2(a, b);
    case NXF_OP_ASIN: return nxf_asin(a);
    case NXF_OP_SINF: return (double)nxf_sinf((float)a);
    case NXF_OP_COSF: return (double)nxf_cosf((float)a);
    case NXF_OP_EXPF: return (double)nxf_expf((float)a);
    case NXF_OP_LOGF: return (double)nxf_logf((float)a);
    case NXF_OP_ATAN2F: return (double)nxf_atan2f((float)a, (float)b);
    case NXF_OP_ASINF: return (double)nxf_asinf((float)a);
    default: return nxf_nan();
    }
}

#endif /* NEXUS_FMATH_H */
