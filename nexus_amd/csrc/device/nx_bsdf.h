// nx_bsdf.h — BSDF sample / eval in the local shading frame (z = normal).
// Restated from the reference's device BSDFs (paths relative to /root/reference/Nexus/src/Cuda/BSDF):
//   LambertianBSDF.cuh:9-39, DielectricBSDF.cuh:17-119 (Walter 2007 rough dielectric, Beckmann),
//   PlasticBSDF.cuh:18-106, ConductorBSDF.cuh:10-49 (Sample only), Fresnel.cuh:5-76, Microfacet.cuh:9-81.
// Reference quirks are mirrored: the Rperp denominator uses cosThetaT twice (Fresnel.cuh:28), `eta` handed
// to Fresnel is always 1/ior (DielectricBSDF.cuh:44,76), BeckmannD divides by a double (Microfacet.cuh:18).
// conductor_eval is an extension: the reference has no conductor Eval and its conductor kernel is empty
// (PathTracer.cu:475-478); it is only reachable in NX_CONDUCTOR_EXTENDED mode.
#pragma once

#include "nexus_pod.h"
#include "nx_rng.h"

namespace nxd {

NXD float dielectric_reflectance(float eta, float cosThetaI, float& cosThetaT)
{
    if (cosThetaI < 0.0f) { eta = 1.0f / eta; cosThetaI = -cosThetaI; }
    const float sinThetaTSq = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSq > 1.0f) { cosThetaT = 0.0f; return 1.0f; }
    cosThetaT = sqrtf(fmaxf(0.0f, 1.0f - sinThetaTSq));
    const float Rparl = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    const float Rperp = (eta * cosThetaT - cosThetaI) / (eta * cosThetaT + cosThetaT);
    return (Rparl * Rparl + Rperp * Rperp) * 0.5f;
}

NXD float complex_reflectance1(float cosThetaI, float eta, float k)
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

NXD float beckmann_d(float alpha, float mDotN)
{
    const float alphaSq = alpha * alpha;
    const float cosThetaSq = mDotN * mDotN;
    const float numerator = nxf_expf((cosThetaSq - 1.0f) / (alphaSq * cosThetaSq));
    const double denominator = kPiD * alphaSq * cosThetaSq * cosThetaSq;
    return (float)(numerator / denominator);
}
NXD float smith_g_a(float alpha, float sDotN) { return sDotN / (alpha * sqrtf(1.0f - fminf(0.99999f, sDotN * sDotN))); }
NXD float smith_g1(float a)
{
    if (a < 1.6f) return ((3.535f + 2.181f * a) * a) / (1.0f + (2.276f + 2.577f * a) * a);
    return 1.0f;
}
NXD float smith_g2(float alpha, float woDotN, float wiDotN) { return smith_g1(smith_g_a(alpha, woDotN)) * smith_g1(smith_g_a(alpha, wiDotN)); }
NXD float weight_beckmann_walter(float alpha, float wiDotM, float woDotN, float wiDotN, float mDotN)
{
    return (wiDotM * smith_g2(alpha, woDotN, wiDotN)) / (wiDotN * mDotN);
}
NXD float walter_reflection_pdf(float alpha, float mDotN, float wiDotM) { return beckmann_d(alpha, mDotN) * mDotN / (4.0f * wiDotM); }
NXD float walter_refraction_pdf(float alpha, float mDotN, float wiDotM, float woDotM, float eta)
{
    return beckmann_d(alpha, mDotN) * mDotN * woDotM / squaref(eta * wiDotM + woDotM);
}
NXD f3 sample_half_beckmann(float alpha, uint32_t& rng)
{
    const float a = alpha * 0.5f + alpha * 0.5f;
    const float ux = rng_next(rng);
    const float uy = rng_next(rng);
    const float tanThetaSquared = -(a * a) * nxf_logf(1.0f - ux);
    const float phi = kTwoPi * uy;
    const float cosTheta = (float)(1.0 / sqrtf(1.0f + tanThetaSquared));
    const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    return normalize3(mk3(sinTheta * nxf_cosf(phi), sinTheta * nxf_sinf(phi), cosTheta));
}
NXD f3 reflect3(f3 i, f3 n) { return i - (n * 2.0f) * dot3(n, i); }
NXD float rough_alpha(float wiz, float roughness) { return clampf((1.2f - 0.2f * sqrtf(fabsf(wiz))) * roughness * roughness, 1.0e-4f, 1.0f); }

// material views: the 28-byte union of nx_material
struct MatParams {
    f3 albedo;       // diffuse / dielectric / plastic
    float roughness; // dielectric / plastic / conductor
    float ior;       // dielectric / plastic
    f3 cIor, cK;     // conductor
};

template <int TYPE> struct Bsdf;

template <> struct Bsdf<NX_MAT_DIFFUSE> {
    static NXD bool eval(const MatParams& m, f3 wi, f3 wo, f3& thr, float& pdf)
    {
        if (!(wi.z * wo.z > 0.0f)) return false;
        thr = (m.albedo * kInvPi) * wo.z;
        pdf = kInvPi * wo.z;
        return pdf_valid(pdf);
    }
    static NXD bool sample(const MatParams& m, f3 wi, uint32_t& rng, f3& wo, f3& thr, float& pdf)
    {
        wo = cosine_hemisphere(rng);
        thr = m.albedo;
        pdf = kInvPi * wo.z;
        return pdf_valid(pdf);
    }
};

template <> struct Bsdf<NX_MAT_DIELECTRIC> {
    static NXD bool eval(const MatParams& mat, f3 wi, f3 wo, f3& thr, float& pdf)
    {
        const float alpha = rough_alpha(wi.z, mat.roughness);
        const float eta = wi.z < 0.0f ? mat.ior : 1 / mat.ior;
        const float wiDotN = wi.z, woDotN = wo.z;
        const bool reflected = wiDotN * woDotN > 0.0f;
        f3 m;
        if (reflected) m = normalize3(wo + wi) * sgnE(wiDotN);
        else m = -normalize3(wi * eta + wo);
        float cosThetaT;
        const float wiDotM = dot3(wi, m), woDotM = dot3(wo, m);
        const float F = dielectric_reflectance(1.0f / mat.ior, wiDotM, cosThetaT);
        const float G = smith_g2(alpha, fabsf(woDotN), fabsf(wiDotN));
        const float D = beckmann_d(alpha, m.z);
        if (reflected) {
            thr = mk3(F * G * D / (4.0f * fabsf(wiDotN)));
            pdf = F * D * m.z / (4.0f * fabsf(wiDotM));
        } else {
            const float s = fabsf(wiDotM * woDotM) * (1.0f - F) * G * D / (fabsf(wiDotN) * squaref(eta * wiDotM + woDotM));
            thr = mat.albedo * s;
            pdf = (1.0f - F) * D * m.z * fabsf(woDotM) / squaref(eta * wiDotM + woDotM);
        }
        return pdf_valid(pdf);
    }
    static NXD bool sample(const MatParams& mat, f3 wi, uint32_t& rng, f3& wo, f3& thr, float& pdf)
    {
        const float alpha = rough_alpha(wi.z, mat.roughness);
        const float eta = wi.z < 0.0f ? mat.ior : 1 / mat.ior;
        const f3 m = sample_half_beckmann(alpha, rng);
        const float wiDotM = dot3(wi, m);
        float cosThetaT;
        const float fr = dielectric_reflectance(1.0f / mat.ior, wiDotM, cosThetaT);
        if (rng_next(rng) < fr) {
            wo = reflect3(-wi, m);
            if (wo.z * wi.z < 0.0f) return false;
            const float weight = weight_beckmann_walter(alpha, fabsf(wiDotM), fabsf(wo.z), fabsf(wi.z), m.z);
            thr = mk3(weight);
            pdf = fr * walter_reflection_pdf(alpha, m.z, fabsf(wiDotM));
        } else {
            wo = m * (eta * wiDotM - sgnE(wiDotM) * cosThetaT) - wi * eta;
            const float weight = weight_beckmann_walter(alpha, fabsf(wiDotM), fabsf(wo.z), fabsf(wi.z), m.z);
            if (weight > 1.0e10) return false;
            if (wo.z * wi.z > 0.0f) return false;
            thr = mat.albedo * weight;
            const float woDotM = dot3(wo, m);
            pdf = (1.0f - fr) * walter_refraction_pdf(alpha, m.z, fabsf(wiDotM), fabsf(woDotM), eta);
        }
        return pdf_valid(pdf);
    }
};

template <> struct Bsdf<NX_MAT_PLASTIC> {
    static NXD bool eval(const MatParams& mat, f3 wi, f3 wo, f3& thr, float& pdf)
    {
        const float alpha = rough_alpha(wi.z, mat.roughness);
        const float wiDotN = wi.z, woDotN = wo.z;
        if (!(wiDotN * woDotN > 0.0f)) return false;
        const f3 m = normalize3(wo + wi);
        float cosThetaT;
        const float wiDotM = dot3(wi, m);
        const float F = dielectric_reflectance(1.0f / mat.ior, wiDotM, cosThetaT);
        const float G = smith_g2(alpha, fabsf(woDotN), fabsf(wiDotN));
        const float D = beckmann_d(alpha, m.z);
        const f3 brdf = mk3(F * G * D / (4.0f * fabsf(wiDotN)));
        const f3 btdf = ((mat.albedo * (1.0f - F)) * kInvPi) * wo.z;
        thr = brdf + btdf;
        const float pdfSpecular = D * m.z / (4.0f * wiDotM);
        const float pdfDiffuse = wo.z * kInvPi;
        pdf = F * pdfSpecular + (1.0f - F) * pdfDiffuse;
        return pdf_valid(pdf);
    }
    static NXD bool sample(const MatParams& mat, f3 wi, uint32_t& rng, f3& wo, f3& thr, float& pdf)
    {
        const float alpha = rough_alpha(wi.z, mat.roughness);
        const f3 m = sample_half_beckmann(alpha, rng);
        const float wiDotM = dot3(wi, m);
        float cosThetaT;
        const float fr = dielectric_reflectance(1.0f / mat.ior, wiDotM, cosThetaT);
        if (rng_next(rng) < fr) {
            wo = reflect3(-wi, m);
            if (wo.z * wi.z < 0.0f) return false;
            const float weight = weight_beckmann_walter(alpha, fabsf(wiDotM), fabsf(wo.z), fabsf(wi.z), m.z);
            thr = mk3(weight);
            pdf = fr * walter_reflection_pdf(alpha, m.z, fabsf(wiDotM));
        } else {
            wo = cosine_hemisphere(rng);
            thr = mat.albedo;
            pdf = (1.0f - fr) * kInvPi * wo.z;
        }
        return pdf_valid(pdf);
    }
};

template <> struct Bsdf<NX_MAT_CONDUCTOR> {
    static NXD f3 fresnel(const MatParams& mat, float wiDotM)
    {
        return mk3(complex_reflectance1(wiDotM, mat.cIor.x, mat.cK.x), complex_reflectance1(wiDotM, mat.cIor.y, mat.cK.y),
                   complex_reflectance1(wiDotM, mat.cIor.z, mat.cK.z));
    }
    static NXD bool eval(const MatParams& mat, f3 wi, f3 wo, f3& thr, float& pdf)
    {
        const float alpha = rough_alpha(wi.z, mat.roughness);
        if (!(wi.z * wo.z > 0.0f)) return false;
        const f3 m = normalize3(wo + wi) * sgnE(wi.z);
        const float wiDotM = dot3(wi, m);
        const f3 F = fresnel(mat, wiDotM);
        const float G = smith_g2(alpha, fabsf(wo.z), fabsf(wi.z));
        const float D = beckmann_d(alpha, m.z);
        thr = F * (G * D / (4.0f * fabsf(wi.z)));
        pdf = D * m.z / (4.0f * fabsf(wiDotM));
        return pdf_valid(pdf);
    }
    static NXD bool sample(const MatParams& mat, f3 wi, uint32_t& rng, f3& wo, f3& thr, float& pdf)
    {
        const float alpha = rough_alpha(wi.z, mat.roughness);
        const f3 m = sample_half_beckmann(alpha, rng);
        const float wiDotM = dot3(wi, m);
        const f3 F = fresnel(mat, wiDotM);
        wo = reflect3(-wi, m);
        const float weight = weight_beckmann_walter(alpha, fabsf(wiDotM), fabsf(wo.z), fabsf(wi.z), m.z);
        if (weight > 1.0e10) return false;
        if (wo.z * wi.z < 0.0f) return false;
        thr = F * weight;
        pdf = walter_reflection_pdf(alpha, m.z, fabsf(wiDotM));
        return true;
    }
};

}  // namespace nxd
