/*
 * orc_shade.c — CPU oracle: RNG, samplers, BSDFs and the software texture fetch.
 *
 * TEST INFRASTRUCTURE ONLY (see nexus_oracle.h).  Restates
 *   /root/reference/Nexus/src/Cuda/Random.cuh:24-134      jenkinsHash, xorShift, InitRNG, Rand, samplers
 *   /root/reference/Nexus/src/Cuda/Sampler.cuh:7-62       PowerHeuristic, Uniform, UniformSampleTriangle, IsPdfValid
 *   /root/reference/Nexus/src/Cuda/BSDF/LambertianBSDF.cuh:9-39
 *   /root/reference/Nexus/src/Cuda/BSDF/DielectricBSDF.cuh:17-119
 *   /root/reference/Nexus/src/Cuda/BSDF/PlasticBSDF.cuh:18-106
 *   /root/reference/Nexus/src/Cuda/BSDF/ConductorBSDF.cuh:10-49 (Sample only in the reference)
 *   /root/reference/Nexus/src/Cuda/BSDF/Fresnel.cuh:5-76, Microfacet.cuh:9-81
 * Where the reference promotes to double through `#define PI 3.14159265358979323846`
 * (Utils/Utils.h:7) or double literals, the same promotion is written out explicitly here.
 * C++ overload resolution picks the float overload for float arguments (sqrt, exp, log, cos, sin, pow,
 * fabs, atan2, asin); this C file therefore calls the f-suffixed functions in those places — the
 * transcendental ones through include/nexus_fmath.h (nxf_*), the text the device code calls too.
 * Function-argument evaluation order for make_float2(Rand(), Rand()) is taken left to right.
 */
#include "nexus_oracle.h"
#include "orc_math.h"
#include "orc_shade.h"

/* ------------------------------------------------------------------------------------------------ */
/* RNG — Random.cuh                                                                                 */

uint32_t orc_jenkins(uint32_t x)
{
    x += x << 10;
    x ^= x >> 6;
    x += x << 3;
    x ^= x >> 11;
    x += x << 15;
    return x;
}

/* Random.cuh:71-77 */
uint32_t orc_rng_init_pixel(uint32_t px, uint32_t py, uint32_t resX, uint32_t frame)
{
    uint32_t s = (px * 1u + py * resX) ^ orc_jenkins(frame);
    if (s == 0) s = 1;
    return orc_jenkins(s);
}

/* Random.cuh:79-82: InitRNG(index) == InitRNG(uint2(1, index)) */
uint32_t orc_rng_init_index(uint32_t index, uint32_t resX, uint32_t frame) { return orc_rng_init_pixel(1u, index, resX, frame); }

/* PIXEL_KEYED extension (not in the reference): stream depends on the global pixel, the bounce and the
 * stage (0 = logic, 1 = shade) instead of the queue slot. */
uint32_t orc_rng_init_keyed(uint32_t globalPixel, uint32_t bounce, uint32_t frame, uint32_t stage)
{
    uint32_t h = orc_jenkins(globalPixel + 0x9e3779b9u * (bounce * 2u + stage + 1u));
    h ^= orc_jenkins(frame);
    if (h == 0) h = 1;
    return orc_jenkins(h);
}

/* Random.cuh:56-69,84-87 */
float orc_rand(uint32_t *s)
{
    uint32_t x = *s;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    *s = x;
    return u2f(0x3f800000u | (x >> 9)) - 1.0f;
}

/* Random.cuh:114-125 */
f3 orc_cosine_hemisphere(uint32_t *rng)
{
    const float r1 = orc_rand(rng);
    const float r2 = orc_rand(rng);
    const float B = sqrtf(r2);
    const double phi = 2 * ORC_PI * r1;
    double sinPhi, cosPhi;
    nxf_sincos(phi, &sinPhi, &cosPhi);
    const float x = (float)(cosPhi * B);
    const float y = (float)(sinPhi * B);
    const float z = sqrtf(1 - r2);
    return mk3(x, y, z);
}

/* Random.cuh:127-134 */
f2 orc_unit_disk(uint32_t *rng)
{
    f2 p;
    do {
        const float a = orc_rand(rng);
        const float b = orc_rand(rng);
        p.x = 2.0f * (a - 0.5f);
        p.y = 2.0f * (b - 0.5f);
    } while (sqrtf(p.x * p.x + p.y * p.y) >= 1.0f);
    return p;
}

/* Sampler.cuh */
int orc_pdf_valid(float pdf) { return isfinite(pdf) && pdf > 1.0e-4f; }
float orc_power_heuristic(float a, float b) { return a * a / (a * a + b * b); }
/* Sampler.cuh UniformSample*: floor(rand * max).  rand < 1, but rand * max can round up to max when max > 2^23: the
 * reference then reads one element past the end; here (and in the device code) the index is clamped to max - 1. */
uint32_t orc_uniform(uint32_t max, uint32_t *rng)
{
    const uint32_t i = (uint32_t)floorf(orc_rand(rng) * (float)max);
    return (max != 0u && i >= max) ? max - 1u : i;
}
f2 orc_uniform_triangle(uint32_t *rng)
{
    const float a = orc_rand(rng);
    const float b = orc_rand(rng);
    const float su0 = sqrtf(a);
    f2 r = {1 - su0, b * su0};
    return r;
}

/* ------------------------------------------------------------------------------------------------ */
/* Fresnel.cuh                                                                                      */

static float dielectric_reflectance(float eta, float cosThetaI, float *cosThetaT)
{
    if (cosThetaI < 0.0f) { eta = 1.0f / eta; cosThetaI = -cosThetaI; }
    const float sinThetaTSq = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSq > 1.0f) { *cosThetaT = 0.0f; return 1.0f; }
    *cosThetaT = sqrtf(fmaxf(0.0f, 1.0f - sinThetaTSq));
    const float Rparl = (eta * cosThetaI - *cosThetaT) / (eta * cosThetaI + *cosThetaT);
    /* the denominator uses cosThetaT twice in the reference (Fresnel.cuh:28); mirrored */
    const float Rperp = (eta * *cosThetaT - cosThetaI) / (eta * *cosThetaT + *cosThetaT);
    return (Rparl * Rparl + Rperp * Rperp) * 0.5f;
}

static float complex_reflectance1(float cosThetaI, float eta, float k)
{
    cosThetaI = clampf(cosThetaI, 0.0f, 1.0f);
    const float cosThetaISq = cosThetaI * cosThetaI;
    const float sinThetaISq = fmaxf(1.0f - cosThetaISq, 0.0f);
    const float sinThetaIQu = sinThetaISq * sinThetaISq;
    const float innerTerm = eta * eta - k * k - sinThetaISq;
    const float aSqPlusBSq = sqrtf(fmaxf(innerTerm * innerTerm + 4.0f * eta * eta * k * k, 0.0f));
    const float a = sqrtf(fmaxf((aSqPlusBSq + innerTerm) * 0.5f, 0.0f));
    const float Rs = ((aSqPlusBSq + cosThetaISq) - (2.0f * a * cosThetaI)) / ((aSqPlusBSq + cosThetaISq) + (2.0f * a * cosThetaI));
    const float Rp = ((cosThetaISq * aSqPlusBSq + sinThetaIQu) - (2.0f * a * cosThetaI * sinThetaISq)) /
                     ((cosThetaISq * aSqPlusBSq + sinThetaIQu) + (2.0f * a * cosThetaI * sinThetaISq));
    return 0.5f * (Rs + Rs * Rp);
}

/* ------------------------------------------------------------------------------------------------ */
/* Microfacet.cuh                                                                                   */

static float beckmann_d(float alpha, float mDotN)
{
    const float alphaSq = alpha * alpha;
    const float cosThetaSq = mDotN * mDotN;
    const float numerator = nxf_expf((cosThetaSq - 1.0f) / (alphaSq * cosThetaSq));
    const double denominator = ORC_PI * alphaSq * cosThetaSq * cosThetaSq;
    return (float)(numerator / denominator);
}
static float smith_g_a(float alpha, float sDotN) { return sDotN / (alpha * sqrtf(1.0f - fminf(0.99999f, sDotN * sDotN))); }
static float smith_g1(float a)
{
    if (a < 1.6f) return ((3.535f + 2.181f * a) * a) / (1.0f + (2.276f + 2.577f * a) * a);
    return 1.0f;
}
static float smith_g2(float alpha, float woDotN, float wiDotN)
{
    const float aL = smith_g_a(alpha, woDotN);
    const float aV = smith_g_a(alpha, wiDotN);
    return smith_g1(aL) * smith_g1(aV);
}
static float weight_beckmann_walter(float alpha, float wiDotM, float woDotN, float wiDotN, float mDotN)
{
    return (wiDotM * smith_g2(alpha, woDotN, wiDotN)) / (wiDotN * mDotN);
}
static float walter_reflection_pdf(float alpha, float mDotN, float wiDotM) { return beckmann_d(alpha, mDotN) * mDotN / (4.0f * wiDotM); }
static float walter_refraction_pdf(float alpha, float mDotN, float wiDotM, float woDotM, float eta)
{
    return beckmann_d(alpha, mDotN) * mDotN * woDotM / squaref(eta * wiDotM + woDotM);
}
static f3 sample_half_beckmann(float alpha, uint32_t *rng)
{
    const float a = alpha * 0.5f + alpha * 0.5f; /* dot(make_float2(alpha), make_float2(0.5f, 0.5f)) */
    const float ux = orc_rand(rng);
    const float uy = orc_rand(rng);
    const float tanThetaSquared = -(a * a) * nxf_logf(1.0f - ux);
    const float phi = ORC_TWO_PI * uy;
    const float cosTheta = (float)(1.0 / sqrtf(1.0f + tanThetaSquared));
    const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    return normalize3(mk3(sinTheta * nxf_cosf(phi), sinTheta * nxf_sinf(phi), cosTheta));
}

static f3 reflect3(f3 i, f3 n) { return sub3(i, scale3(scale3(n, 2.0f), dot3(n, i))); } /* cuda_math.h:1472-1475 */

static float rough_alpha(float wiz, float roughness)
{
    return clampf((1.2f - 0.2f * sqrtf(fabsf(wiz))) * roughness * roughness, 1.0e-4f, 1.0f);
}

/* ------------------------------------------------------------------------------------------------ */
/* BSDFs                                                                                            */

static int lambert_eval(const nx_material *m, f3 wi, f3 wo, f3 *thr, float *pdf)
{
    if (!(wi.z * wo.z > 0.0f)) return 0;
    *thr = scale3(scale3(ld3(m->diffuse.albedo), ORC_INV_PI), wo.z);
    *pdf = ORC_INV_PI * wo.z;
    return orc_pdf_valid(*pdf);
}
static int lambert_sample(const nx_material *m, f3 wi, uint32_t *rng, f3 *wo, f3 *thr, float *pdf)
{
    (void)wi;
    *wo = orc_cosine_hemisphere(rng);
    *thr = ld3(m->diffuse.albedo);
    *pdf = ORC_INV_PI * wo->z;
    return orc_pdf_valid(*pdf);
}

static int dielectric_eval(const nx_material *mat, f3 wi, f3 wo, f3 *thr, float *pdf)
{
    const float alpha = rough_alpha(wi.z, mat->dielectric.roughness);
    const float eta = wi.z < 0.0f ? mat->dielectric.ior : 1 / mat->dielectric.ior;
    const float wiDotN = wi.z, woDotN = wo.z;
    const int reflected = wiDotN * woDotN > 0.0f;
    f3 m;
    if (reflected) m = scale3(normalize3(add3(wo, wi)), sgnE(wiDotN));
    else m = neg3(normalize3(add3(scale3(wi, eta), wo)));
    float cosThetaT;
    const float wiDotM = dot3(wi, m), woDotM = dot3(wo, m);
    const float F = dielectric_reflectance(1.0f / mat->dielectric.ior, wiDotM, &cosThetaT);
    const float G = smith_g2(alpha, fabsf(woDotN), fabsf(wiDotN));
    const float D = beckmann_d(alpha, m.z);
    if (reflected) {
        *thr = mk3s(F * G * D / (4.0f * fabsf(wiDotN)));
        *pdf = F * D * m.z / (4.0f * fabsf(wiDotM));
    } else {
        const float s = fabsf(wiDotM * woDotM) * (1.0f - F) * G * D / (fabsf(wiDotN) * squaref(eta * wiDotM + woDotM));
        *thr = scale3(ld3(mat->dielectric.albedo), s);
        *pdf = (1.0f - F) * D * m.z * fabsf(woDotM) / squaref(eta * wiDotM + woDotM);
    }
    return orc_pdf_valid(*pdf);
}
static int dielectric_sample(const nx_material *mat, f3 wi, uint32_t *rng, f3 *wo, f3 *thr, float *pdf)
{
    const float alpha = rough_alpha(wi.z, mat->dielectric.roughness);
    const float eta = wi.z < 0.0f ? mat->dielectric.ior : 1 / mat->dielectric.ior;
    const f3 m = sample_half_beckmann(alpha, rng);
    const float wiDotM = dot3(wi, m);
    float cosThetaT;
    const float fr = dielectric_reflectance(1.0f / mat->dielectric.ior, wiDotM, &cosThetaT);
    if (orc_rand(rng) < fr) {
        *wo = reflect3(neg3(wi), m);
        if (wo->z * wi.z < 0.0f) return 0;
        const float weight = weight_beckmann_walter(alpha, fabsf(wiDotM), fabsf(wo->z), fabsf(wi.z), m.z);
        *thr = mk3s(weight);
        *pdf = fr * walter_reflection_pdf(alpha, m.z, fabsf(wiDotM));
    } else {
        *wo = sub3(scale3(m, eta * wiDotM - sgnE(wiDotM) * cosThetaT), scale3(wi, eta));
        const float weight = weight_beckmann_walter(alpha, fabsf(wiDotM), fabsf(wo->z), fabsf(wi.z), m.z);
        if (weight > 1.0e10) return 0;
        if (wo->z * wi.z > 0.0f) return 0;
        *thr = scale3(ld3(mat->dielectric.albedo), weight);
        const float woDotM = dot3(*wo, m);
        *pdf = (1.0f - fr) * walter_refraction_pdf(alpha, m.z, fabsf(wiDotM), fabsf(woDotM), eta);
    }
    return orc_pdf_valid(*pdf);
}

static int plastic_eval(const nx_material *mat, f3 wi, f3 wo, f3 *thr, float *pdf)
{
    const float alpha = rough_alpha(wi.z, mat->plastic.roughness);
    const float wiDotN = wi.z, woDotN = wo.z;
    if (!(wiDotN * woDotN > 0.0f)) return 0;
    const f3 m = normalize3(add3(wo, wi));
    float cosThetaT;
    const float wiDotM = dot3(wi, m);
    const float F = dielectric_reflectance(1.0f / mat->plastic.ior, wiDotM, &cosThetaT);
    const float G = smith_g2(alpha, fabsf(woDotN), fabsf(wiDotN));
    const float D = beckmann_d(alpha, m.z);
    const f3 brdf = mk3s(F * G * D / (4.0f * fabsf(wiDotN)));
    const f3 btdf = scale3(scale3(scale3(ld3(mat->plastic.albedo), (1.0f - F)), ORC_INV_PI), wo.z);
    *thr = add3(brdf, btdf);
    const float pdfSpecular = D * m.z / (4.0f * wiDotM);
    const float pdfDiffuse = wo.z * ORC_INV_PI;
    *pdf = F * pdfSpecular + (1.0f - F) * pdfDiffuse;
    return orc_pdf_valid(*pdf);
}
static int plastic_sample(const nx_material *mat, f3 wi, uint32_t *rng, f3 *wo, f3 *thr, float *pdf)
{
    const float alpha = rough_alpha(wi.z, mat->plastic.roughness);
    const f3 m = sample_half_beckmann(alpha, rng);
    const float wiDotM = dot3(wi, m);
    float cosThetaT;
    const float fr = dielectric_reflectance(1.0f / mat->plastic.ior, wiDotM, &cosThetaT);
    if (orc_rand(rng) < fr) {
        *wo = reflect3(neg3(wi), m);
        if (wo->z * wi.z < 0.0f) return 0;
        const float weight = weight_beckmann_walter(alpha, fabsf(wiDotM), fabsf(wo->z), fabsf(wi.z), m.z);
        *thr = mk3s(weight);
        *pdf = fr * walter_reflection_pdf(alpha, m.z, fabsf(wiDotM));
    } else {
        *wo = orc_cosine_hemisphere(rng);
        *thr = ld3(mat->plastic.albedo);
        *pdf = (1.0f - fr) * ORC_INV_PI * wo->z;
    }
    return orc_pdf_valid(*pdf);
}

/* ConductorBSDF.cuh:24-48 (Sample).  The reference never runs it (kernel body commented out). */
static int conductor_sample(const nx_material *mat, f3 wi, uint32_t *rng, f3 *wo, f3 *thr, float *pdf)
{
    const float alpha = rough_alpha(wi.z, mat->conductor.roughness);
    const f3 m = sample_half_beckmann(alpha, rng);
    const float wiDotM = dot3(wi, m);
    const f3 F = mk3(complex_reflectance1(wiDotM, mat->conductor.ior[0], mat->conductor.k[0]),
                     complex_reflectance1(wiDotM, mat->conductor.ior[1], mat->conductor.k[1]),
                     complex_reflectance1(wiDotM, mat->conductor.ior[2], mat->conductor.k[2]));
    *wo = reflect3(neg3(wi), m);
    const float weight = weight_beckmann_walter(alpha, fabsf(wiDotM), fabsf(wo->z), fabsf(wi.z), m.z);
    if (weight > 1.0e10) return 0;
    if (wo->z * wi.z < 0.0f) return 0;
    *thr = scale3(F, weight);
    *pdf = walter_reflection_pdf(alpha, m.z, fabsf(wiDotM));
    return 1;
}
/* EXTENSION (the reference has no conductor Eval): the reflection lobe matching conductor_sample. */
static int conductor_eval(const nx_material *mat, f3 wi, f3 wo, f3 *thr, float *pdf)
{
    const float alpha = rough_alpha(wi.z, mat->conductor.roughness);
    if (!(wi.z * wo.z > 0.0f)) return 0;
    const f3 m = scale3(normalize3(add3(wo, wi)), sgnE(wi.z));
    const float wiDotM = dot3(wi, m);
    const f3 F = mk3(complex_reflectance1(wiDotM, mat->conductor.ior[0], mat->conductor.k[0]),
                     complex_reflectance1(wiDotM, mat->conductor.ior[1], mat->conductor.k[1]),
                     complex_reflectance1(wiDotM, mat->conductor.ior[2], mat->conductor.k[2]));
    const float G = smith_g2(alpha, fabsf(wo.z), fabsf(wi.z));
    const float D = beckmann_d(alpha, m.z);
    *thr = scale3(F, G * D / (4.0f * fabsf(wi.z)));
    *pdf = D * m.z / (4.0f * fabsf(wiDotM));
    return orc_pdf_valid(*pdf);
}

int orc_bsdf_sample_f3(const nx_material *m, f3 wi, uint32_t *rng, f3 *wo, f3 *thr, float *pdf)
{
    switch (m->type) {
    case NX_MAT_DIFFUSE: return lambert_sample(m, wi, rng, wo, thr, pdf);
    case NX_MAT_DIELECTRIC: return dielectric_sample(m, wi, rng, wo, thr, pdf);
    case NX_MAT_PLASTIC: return plastic_sample(m, wi, rng, wo, thr, pdf);
    case NX_MAT_CONDUCTOR: return conductor_sample(m, wi, rng, wo, thr, pdf);
    default: return 0;
    }
}
int orc_bsdf_eval_f3(const nx_material *m, f3 wi, f3 wo, f3 *thr, float *pdf)
{
    switch (m->type) {
    case NX_MAT_DIFFUSE: return lambert_eval(m, wi, wo, thr, pdf);
    case NX_MAT_DIELECTRIC: return dielectric_eval(m, wi, wo, thr, pdf);
    case NX_MAT_PLASTIC: return plastic_eval(m, wi, wo, thr, pdf);
    case NX_MAT_CONDUCTOR: return conductor_eval(m, wi, wo, thr, pdf);
    default: return 0;
    }
}

int orc_bsdf_sample(const nx_material *m, const float wi[3], uint32_t *rng, float wo[3], float throughput[3], float *pdf)
{
    f3 o = mk3s(0.0f), t = mk3s(0.0f);
    *pdf = 0.0f;
    const int ok = orc_bsdf_sample_f3(m, ld3(wi), rng, &o, &t, pdf);
    st3(wo, o);
    st3(throughput, t);
    return ok;
}
int orc_bsdf_eval(const nx_material *m, const float wi[3], const float wo[3], float throughput[3], float *pdf)
{
    f3 t = mk3s(0.0f);
    *pdf = 0.0f;
    const int ok = orc_bsdf_eval_f3(m, ld3(wi), ld3(wo), &t, pdf);
    st3(throughput, t);
    return ok;
}

/* The same two calls over arrays of the C-ABI hooks' query / result records (nexus_pod.h): what nxhip_bsdf_sample_batch /
 * nxhip_bsdf_eval_batch do on the device, so that the statistical pins of tests/test_bsdf_pins.py can run a few hundred thousand
 * queries on either side. */
void orc_bsdf_sample_batch(const nx_material *m, const nx_bsdf_query *q, uint32_t n, nx_bsdf_result *r)
{
    for (uint32_t i = 0; i < n; i++) {
        uint32_t s = q[i].rng;
        r[i].pdf = 0.0f;
        r[i].ok = (uint32_t)orc_bsdf_sample(m, q[i].wi, &s, r[i].wo, r[i].throughput, &r[i].pdf);
        r[i].rngOut = s;
    }
}
void orc_bsdf_eval_batch(const nx_material *m, const nx_bsdf_query *q, uint32_t n, nx_bsdf_result *r)
{
    for (uint32_t i = 0; i < n; i++) {
        r[i].pdf = 0.0f;
        r[i].ok = (uint32_t)orc_bsdf_eval(m, q[i].wi, q[i].wo, r[i].throughput, &r[i].pdf);
        r[i].wo[0] = q[i].wo[0]; r[i].wo[1] = q[i].wo[1]; r[i].wo[2] = q[i].wo[2];
        r[i].rngOut = q[i].rng;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* Software tex2D<float4>: normalised coordinates, wrap addressing, bilinear filter with 8-bit        */
/* fractional weights, sRGB decode of RGB before filtering (texture descriptor: Assets/Texture.cpp:  */
/* 26-33).  The filtering itself lives in CUDA hardware, outside /root/reference: PARITY UNPINNED.  */

float orc_srgb_to_linear(uint8_t c)
{
    const float x = (float)c / 255.0f;
    return x <= 0.04045f ? x / 12.92f : powf((x + 0.055f) / 1.055f, 2.4f);
}

static int wrapi(int i, int n) { i %= n; return i < 0 ? i + n : i; }

f4 orc_tex2d_f4(const nx_texture_desc *t, float u, float v)
{
    const int W = (int)t->width, H = (int)t->height;
    const float xb = u * (float)W - 0.5f, yb = v * (float)H - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    const float ax = floorf((xb - fx) * 256.0f + 0.5f) * (1.0f / 256.0f);
    const float ay = floorf((yb - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
    const int i0 = wrapi((int)fx, W), i1 = wrapi((int)fx + 1, W);
    const int j0 = wrapi((int)fy, H), j1 = wrapi((int)fy + 1, H);
    const uint8_t *p00 = t->rgba8 + 4 * ((size_t)j0 * W + i0), *p10 = t->rgba8 + 4 * ((size_t)j0 * W + i1);
    const uint8_t *p01 = t->rgba8 + 4 * ((size_t)j1 * W + i0), *p11 = t->rgba8 + 4 * ((size_t)j1 * W + i1);
    float out[4];
    for (int c = 0; c < 4; c++) {
        float t00, t10, t01, t11;
        if (c < 3) { t00 = orc_srgb_to_linear(p00[c]); t10 = orc_srgb_to_linear(p10[c]); t01 = orc_srgb_to_linear(p01[c]); t11 = orc_srgb_to_linear(p11[c]); }
        else { t00 = (float)p00[c] / 255.0f; t10 = (float)p10[c] / 255.0f; t01 = (float)p01[c] / 255.0f; t11 = (float)p11[c] / 255.0f; }
        const float top = t00 + ax * (t10 - t00);
        const float bot = t01 + ax * (t11 - t01);
        out[c] = top + ay * (bot - top);
    }
    f4 r = {out[0], out[1], out[2], out[3]};
    return r;
}

void orc_tex2d(const nx_texture_desc *t, float u, float v, float out[4])
{
    const f4 r = orc_tex2d_f4(t, u, v);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

/* include/nexus_fmath.h on arrays (see nexus_oracle.h) */
void orc_fmath_batch(int op, const double *a, const double *b, uint32_t n, double *out)
{
    for (uint32_t i = 0; i < n; i++) out[i] = nxf_apply(op, a[i], b ? b[i] : 0.0);
}
