/*
 * orc_math.h — scalar float3 / Mat4 / quaternion helpers of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by or executed from the
 * product (nexus_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Restates the arithmetic of the reference's helper_math derivative and Mat4
 * (/root/reference/Nexus/src/Utils/cuda_math.h:1143-1535, Math/Mat4.h:142-230).
 *
 * Arithmetic convention shared with the HIP device code (nexus_amd/csrc/device/nx_math.h), so that
 * integer/byte results are bit-exact and float results agree to the last bit wherever only
 * + - * / sqrt fma are involved:
 *   - both sides are compiled with FP contraction OFF;
 *   - dot3, cross3 and the Mat4 transforms use explicit fmaf in the order written below (the
 *     reference's nvcc build contracts these into FMAs too; the exact grouping is ours);
 *   - normalize(v) = v * (1 / sqrtf(dot3(v,v))) with correctly rounded divide and sqrt;
 *   - the transcendental functions of the shading path (sin, cos, exp, log, pow, atan2, asin) come from
 *     include/nexus_fmath.h — one text compiled on both sides, IEEE operations only — so they too agree
 *     bit for bit (round 4; before, libm here and ocml on the device differed by an ulp or two).
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>
#include "../include/nexus_pod.h"
#include "../include/nexus_fmath.h"

typedef struct { float x, y, z; } f3;
typedef struct { float x, y; } f2;
typedef struct { float x, y, z, w; } f4;

#define ORC_PI 3.14159265358979323846 /* double, as Utils/Utils.h:7 */
#define ORC_INV_PI 0.31830988618f     /* Utils/Utils.h:8 */
#define ORC_TWO_PI 6.28318530718f     /* Utils/Utils.h:9 */

static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 mk3s(float s) { f3 r = {s, s, s}; return r; }
static inline f3 ld3(const float *p) { f3 r = {p[0], p[1], p[2]}; return r; }
static inline void st3(float *p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 div3s(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
static inline f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
static inline float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline f3 cross3(f3 a, f3 b)
{
    return mk3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float length3(f3 a) { return sqrtf(dot3(a, a)); }
static inline f3 normalize3(f3 a) { return scale3(a, 1.0f / sqrtf(dot3(a, a))); }
static inline float maxcomp3(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); } /* cuda_math.h:1143 */
static inline f3 min3v(f3 a, f3 b) { return mk3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
static inline f3 max3v(f3 a, f3 b) { return mk3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
static inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
static inline float sgnE(float v) { return v < 0.0f ? -1.0f : 1.0f; } /* Utils/Utils.h:24-28 */
static inline float squaref(float x) { return x * x; }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int32_t f2i(float f) { int32_t u; memcpy(&u, &f, 4); return u; }
static inline float i2f(int32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Mat4::TransformVector / TransformPoint, Math/Mat4.h:216-230 */
static inline f3 mat_vec(const nx_mat4 *m, f3 v)
{
    const float *c = m->cell;
    return mk3(fmaf(c[2], v.z, fmaf(c[1], v.y, c[0] * v.x)),
               fmaf(c[6], v.z, fmaf(c[5], v.y, c[4] * v.x)),
               fmaf(c[10], v.z, fmaf(c[9], v.y, c[8] * v.x)));
}
static inline f3 mat_point(const nx_mat4 *m, f3 v)
{
    const float *c = m->cell;
    return mk3(fmaf(c[2], v.z, fmaf(c[1], v.y, c[0] * v.x)) + c[3],
               fmaf(c[6], v.z, fmaf(c[5], v.y, c[4] * v.x)) + c[7],
               fmaf(c[10], v.z, fmaf(c[9], v.y, c[8] * v.x)) + c[11]);
}
/* invTransform.Transposed().TransformVector(v): 3x3 transpose only, Math/Mat4.h:142-149 */
static inline f3 mat_vec_transposed(const nx_mat4 *m, f3 v)
{
    const float *c = m->cell;
    return mk3(fmaf(c[8], v.z, fmaf(c[4], v.y, c[0] * v.x)),
               fmaf(c[9], v.z, fmaf(c[5], v.y, c[1] * v.x)),
               fmaf(c[10], v.z, fmaf(c[6], v.y, c[2] * v.x)));
}

/* Quaternion helpers, Utils/cuda_math.h:1514-1535 */
static inline f4 rotation_to_z(f3 d)
{
    f4 q;
    if (d.z < -0.99999f) { q.x = 1.0f; q.y = 0.0f; q.z = 0.0f; q.w = 0.0f; return q; }
    {
        const float x = d.y, y = -d.x, z = 0.0f, w = 1.0f + d.z;
        const float inv = 1.0f / sqrtf(fmaf(w, w, fmaf(z, z, fmaf(y, y, x * x))));
        q.x = x * inv; q.y = y * inv; q.z = z * inv; q.w = w * inv;
    }
    return q;
}
static inline f4 invert_rotation(f4 q) { f4 r = {-q.x, -q.y, -q.z, q.w}; return r; }
static inline f3 rotate_point(f4 q, f3 v)
{
    const f3 a = mk3(q.x, q.y, q.z);
    const f3 t0 = scale3(a, 2.0f * dot3(a, v));
    const f3 t1 = scale3(v, q.w * q.w - dot3(a, a));
    const f3 t2 = scale3(cross3(a, v), 2.0f * q.w);
    return add3(add3(t0, t1), t2);
}

/* Barycentric, Cuda/Utils.cuh:47-51 */
static inline f3 bary3(f3 t0, f3 t1, f3 t2, float u, float v)
{
    const float w = 1.0f - u - v;
    return add3(add3(scale3(t1, u), scale3(t2, v)), scale3(t0, w));
}
static inline f2 bary2(const float *t0, const float *t1, const float *t2, float u, float v)
{
    const float w = 1.0f - u - v;
    f2 r;
    r.x = u * t1[0] + v * t2[0] + w * t0[0];
    r.y = u * t1[1] + v * t2[1] + w * t0[1];
    return r;
}

/* OffsetRay, Cuda/Utils.cuh:53-74 (Ray Tracing Gems ch. 6) */
static inline float offset_axis(float p, float n)
{
    const int32_t ofi = (int32_t)(256.0f * n);
    const float pi = i2f(f2i(p) + ((p < 0.0f) ? -ofi : ofi));
    return fabsf(p) < (1.0f / 32.0f) ? p + (1.0f / 65536.0f) * n : pi;
}
static inline f3 offset_ray(f3 p, f3 n) { return mk3(offset_axis(p.x, n.x), offset_axis(p.y, n.y), offset_axis(p.z, n.z)); }

#endif
