// nx_rng.h — RNG and samplers of the shading kernels.
// Restated from /root/reference/Nexus/src/Cuda/Random.cuh:24-134 and Cuda/Sampler.cuh:7-62:
// Jenkins one-at-a-time hash seeding, xorshift32 13/17/5, mantissa-fill float conversion.  The double
// promotions through `#define PI 3.14159265358979323846` (Utils/Utils.h:7) are written out explicitly.
#pragma once

#include "nx_math.h"

namespace nxd {

constexpr double kPiD = 3.14159265358979323846;
constexpr float kInvPi = 0.31830988618f;
constexpr float kTwoPi = 6.28318530718f;

NXD uint32_t jenkins(uint32_t x)
{
    x += x << 10;
    x ^= x >> 6;
    x += x << 3;
    x ^= x >> 11;
    x += x << 15;
    return x;
}
// Random.cuh:71-77
NXD uint32_t rng_init_pixel(uint32_t px, uint32_t py, uint32_t resX, uint32_t frame)
{
    uint32_t s = (px + py * resX) ^ jenkins(frame);
    if (s == 0) s = 1;
    return jenkins(s);
}
// Random.cuh:79-82: InitRNG(index) == InitRNG(uint2(1, index))
NXD uint32_t rng_init_index(uint32_t index, uint32_t resX, uint32_t frame) { return rng_init_pixel(1u, index, resX, frame); }
// NX_RNG_PIXEL_KEYED extension: keyed by global pixel, bounce and stage (0 logic, 1 shade)
NXD uint32_t rng_init_keyed(uint32_t globalPixel, uint32_t bounce, uint32_t frame, uint32_t stage)
{
    uint32_t h = jenkins(globalPixel + 0x9e3779b9u * (bounce * 2u + stage + 1u));
    h ^= jenkins(frame);
    if (h == 0) h = 1;
    return jenkins(h);
}
NXD float rng_next(uint32_t& s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return __uint_as_float(0x3f800000u | (s >> 9)) - 1.0f;
}
// Random.cuh:114-125
NXD f3 cosine_hemisphere(uint32_t& rng)
{
    const float r1 = rng_next(rng);
    const float r2 = rng_next(rng);
    const float B = sqrtf(r2);
    const double phi = 2 * kPiD * r1;
    double sinPhi, cosPhi;
    nxf_sincos(phi, &sinPhi, &cosPhi);
    const float x = (float)(cosPhi * B);
    const float y = (float)(sinPhi * B);
    const float z = sqrtf(1 - r2);
    return mk3(x, y, z);
}
// Random.cuh:127-134
NXD f2 unit_disk(uint32_t& rng)
{
    f2 p;
    do {
        const float a = rng_next(rng);
        const float b = rng_next(rng);
        p.x = 2.0f * (a - 0.5f);
        p.y = 2.0f * (b - 0.5f);
    } while (sqrtf(p.x * p.x + p.y * p.y) >= 1.0f);
    return p;
}
NXD bool pdf_valid(float pdf) { return isfinite(pdf) && pdf > 1.0e-4f; }
NXD float power_heuristic(float a, float b) { return a * a / (a * a + b * b); }
// Sampler.cuh UniformSample*: floor(rand * max).  rand < 1, but rand * max can round up to max when max > 2^23: the
// reference then reads one element past the end; the index is clamped to max - 1 here (and in the CPU restatement used by the tests).
NXD uint32_t uniform_index(uint32_t max, uint32_t& rng)
{
    const uint32_t i = (uint32_t)floorf(rng_next(rng) * (float)max);
    return (max != 0u && i >= max) ? max - 1u : i;
}
NXD f2 uniform_triangle(uint32_t& rng)
{
    const float a = rng_next(rng);
    const float b = rng_next(rng);
    const float su0 = sqrtf(a);
    return f2{1 - su0, b * su0};
}

}  // namespace nxd
