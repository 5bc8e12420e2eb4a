/* fmath_exhaustive.c — the binary32 functions of include/nexus_fmath.h over EVERY binary32 argument of their domains, against
 * glibc's binary64 functions (correct to < 1 ulp of binary64, i.e. exact for this purpose): worst error in ulp of the result.
 *
 *   gcc -O2 -std=c11 -ffp-contract=off -mfma -fopenmp -Iinclude tools/fmath_exhaustive.c -o /tmp/fmath_exhaustive -lm && /tmp/fmath_exhaustive
 *
 * (the oracle's flags: contraction off, fmaf a single instruction).  A measurement tool, not part of the test suite — the suite
 * samples (tests/test_fmath.py); the figures this prints are quoted in nexus_fmath.h and DESIGN.md section 2. */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "nexus_fmath.h"

static float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static double ulp_error(float got, double want)
{
    if (isnan(want)) return isnan(got) ? 0.0 : 1e30;
    if (isinf(want)) return (got == (float)want) ? 0.0 : 1e30;
    const float wf = (float)want;
    if (isinf(wf)) return isinf(got) && (got > 0) == (want > 0) ? 0.0 : (fabs((double)got) >= 3.4028234e38 ? 1.0 : 1e30);  /* rounds to infinity: either neighbour is within an ulp */
    int e;
    frexp(want, &e);
    double ulp = ldexp(1.0, e - 24);
    if (ulp < ldexp(1.0, -149)) ulp = ldexp(1.0, -149);  /* gradual underflow */
    return fabs((double)got - want) / ulp;
}

typedef struct { double worst; uint32_t at; } Worst;

#define SWEEP(name, lo, hi, expr_got, expr_want)                                                     \
    do {                                                                                               \
        if (only && !strstr(name, only)) break;                                                        \
        Worst w = {0.0, 0};                                                                            \
        _Pragma("omp parallel")                                                                        \
        {                                                                                              \
            Worst mine = {0.0, 0};                                                                     \
            _Pragma("omp for schedule(static, 1 << 16)")                                               \
            for (uint64_t u = 0; u < (1ull << 32); u++) {                                              \
                const float x = from_bits((uint32_t)u);                                                \
                if (!(x >= (lo) && x <= (hi))) continue;                                               \
                const double err = ulp_error((expr_got), (expr_want));                                 \
                if (err > mine.worst) { mine.worst = err; mine.at = (uint32_t)u; }                     \
            }                                                                                          \
            _Pragma("omp critical")                                                                    \
            if (mine.worst > w.worst) w = mine;                                                        \
        }                                                                                              \
        printf("%-28s worst %.4f ulp at %.9g (0x%08x)\n", name, w.worst, (double)from_bits(w.at), w.at); \
        fflush(stdout);                                                                                \
    } while (0)

static const char *only;

int main(int argc, char **argv)
{
    only = argc > 1 ? argv[1] : NULL;  /* a substring of the sweeps to run */
    SWEEP("sinf  |x| <= 2 pi", -6.2831855f, 6.2831855f, nxf_sinf(x), sin((double)x));
    SWEEP("cosf  |x| <= 2 pi", -6.2831855f, 6.2831855f, nxf_cosf(x), cos((double)x));
    SWEEP("sinf  |x| <= 1e5", -1.0e5f, 1.0e5f, nxf_sinf(x), sin((double)x));
    SWEEP("cosf  |x| <= 1e5", -1.0e5f, 1.0e5f, nxf_cosf(x), cos((double)x));
    SWEEP("expf  every float", -INFINITY, INFINITY, nxf_expf(x), exp((double)x));
    SWEEP("logf  every float >= 0", 0.0f, INFINITY, nxf_logf(x), log((double)x));
    SWEEP("asinf [-1, 1]", -1.0f, 1.0f, nxf_asinf(x), asin((double)x));
    SWEEP("atan2f(x, 1) every float", -INFINITY, INFINITY, nxf_atan2f(x, 1.0f), atan2((double)x, 1.0));
    SWEEP("atan2f(1, x) every float", -INFINITY, INFINITY, nxf_atan2f(1.0f, x), atan2(1.0, (double)x));
    SWEEP("atan2f(x, -0.37)", -INFINITY, INFINITY, nxf_atan2f(x, -0.37f), atan2((double)x, (double)-0.37f));
    /* two-argument sample: 2^28 pseudo-random pairs over 40 binades each way */
    {
        Worst w = {0.0, 0};
#pragma omp parallel
        {
            Worst mine = {0.0, 0};
#pragma omp for schedule(static, 1 << 16)
            for (uint64_t i = 0; i < (1ull << 28); i++) {
                uint64_t h = i * 0x9e3779b97f4a7c15ull;
                h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
                const uint32_t a = (uint32_t)h, b = (uint32_t)(h >> 32);
                const float y = from_bits((a & 0x807fffffu) | ((107u + (a >> 23) % 40u) << 23));
                const float x = from_bits((b & 0x807fffffu) | ((107u + (b >> 23) % 40u) << 23));
                const double err = ulp_error(nxf_atan2f(y, x), atan2((double)y, (double)x));
                if (err > mine.worst) { mine.worst = err; mine.at = (uint32_t)i; }
            }
#pragma omp critical
            if (mine.worst > w.worst) w = mine;
        }
        printf("%-28s worst %.4f ulp (pair %u)\n", "atan2f 2^28 random pairs", w.worst, w.at);
    }
    return 0;
}
