// nx_math.h — device-side float3 / matrix / quaternion helpers (HIP, gfx950).
//
// Arithmetic convention (shared with the CPU oracle's orc_math.h so that results agree to the bit):
// the device code is compiled with -ffp-contract=off; dot3, cross3 and
// the row-major matrix transforms use explicit fmaf in the order written here; normalize(v) is
// v * (1 / sqrtf(dot3(v,v))) with correctly rounded divide and sqrt (hipcc's default); the transcendental functions of
// the shading path come from include/nexus_fmath.h (nxf_*), one text of IEEE operations compiled here and in the oracle.
// Semantics follow the reference's helper_math derivative (/root/reference/Nexus/src/Utils/cuda_math.h:1143-1535),
// Mat4 (Math/Mat4.h:142-230) and Cuda/Utils.cuh:47-74; nvcc contracts the same expressions into FMAs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nexus_fmath.h"

namespace nxd {

struct f3 { float x, y, z; };
struct f2 { float x, y; };

#define NXD __device__ __forceinline__

NXD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
NXD f3 mk3(float s) { return f3{s, s, s}; }
template <class P>  // (generic or global-address-space pointer to three floats)
NXD f3 ld3(P p) { return f3{p[0], p[1], p[2]}; }
NXD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
NXD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
NXD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
NXD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
NXD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
NXD f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
NXD float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
NXD f3 cross3(f3 a, f3 b) { return f3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))}; }
NXD float length3(f3 a) { return sqrtf(dot3(a, a)); }
NXD f3 normalize3(f3 a) { return a * (1.0f / sqrtf(dot3(a, a))); }
NXD float maxcomp3(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
NXD float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
NXD float sgnE(float v) { return v < 0.0f ? -1.0f : 1.0f; }
NXD float squaref(float x) { return x * x; }

// rows r0..r2 of a row-major 3x4 transform held as float4s
NXD f3 mat_vec(float4 r0, float4 r1, float4 r2, f3 v)
{
    return f3{fmaf(r0.z, v.z, fmaf(r0.y, v.y, r0.x * v.x)), fmaf(r1.z, v.z, fmaf(r1.y, v.y, r1.x * v.x)), fmaf(r2.z, v.z, fmaf(r2.y, v.y, r2.x * v.x))};
}
NXD f3 mat_point(float4 r0, float4 r1, float4 r2, f3 v)
{
    return f3{fmaf(r0.z, v.z, fmaf(r0.y, v.y, r0.x * v.x)) + r0.w, fmaf(r1.z, v.z, fmaf(r1.y, v.y, r1.x * v.x)) + r1.w,
              fmaf(r2.z, v.z, fmaf(r2.y, v.y, r2.x * v.x)) + r2.w};
}
// 16-float row-major matrix in memory (P: a generic or a global-address-space pointer to its floats)
template <class P>
NXD f3 mat_vec(P c, f3 v)
{
    return f3{fmaf(c[2], v.z, fmaf(c[1], v.y, c[0] * v.x)), fmaf(c[6], v.z, fmaf(c[5], v.y, c[4] * v.x)), fmaf(c[10], v.z, fmaf(c[9], v.y, c[8] * v.x))};
}
template <class P>
NXD f3 mat_point(P c, f3 v)
{
    return f3{fmaf(c[2], v.z, fmaf(c[1], v.y, c[0] * v.x)) + c[3], fmaf(c[6], v.z, fmaf(c[5], v.y, c[4] * v.x)) + c[7],
              fmaf(c[10], v.z, fmaf(c[9], v.y, c[8] * v.x)) + c[11]};
}
// (3x3 block of M)^T * v — normals through invTransform.Transposed()
template <class P>
NXD f3 mat_vec_transposed(P c, f3 v)
{
    return f3{fmaf(c[8], v.z, fmaf(c[4], v.y, c[0] * v.x)), fmaf(c[9], v.z, fmaf(c[5], v.y, c[1] * v.x)), fmaf(c[10], v.z, fmaf(c[6], v.y, c[2] * v.x))};
}

// quaternion helpers — cuda_math.h:1514-1535
NXD float4 rotation_to_z(f3 d)
{
    if (d.z < -0.99999f) return make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    const float x = d.y, y = -d.x, z = 0.0f, w = 1.0f + d.z;
    const float inv = 1.0f / sqrtf(fmaf(w, w, fmaf(z, z, fmaf(y, y, x * x))));
    return make_float4(x * inv, y * inv, z * inv, w * inv);
}
NXD float4 invert_rotation(float4 q) { return make_float4(-q.x, -q.y, -q.z, q.w); }
NXD f3 rotate_point(float4 q, f3 v)
{
    const f3 a = mk3(q.x, q.y, q.z);
    const f3 t0 = a * (2.0f * dot3(a, v));
    const f3 t1 = v * (q.w * q.w - dot3(a, a));
    const f3 t2 = cross3(a, v) * (2.0f * q.w);
    return (t0 + t1) + t2;
}

NXD f3 bary3(f3 t0, f3 t1, f3 t2, float u, float v)
{
    const float w = 1.0f - u - v;
    return (t1 * u + t2 * v) + t0 * w;
}
template <class P>
NXD f2 bary2(P t0, P t1, P t2, float u, float v)
{
    const float w = 1.0f - u - v;
    return f2{u * t1[0] + v * t2[0] + w * t0[0], u * t1[1] + v * t2[1] + w * t0[1]};
}

// OffsetRay — Cuda/Utils.cuh:53-74 (Ray Tracing Gems ch. 6)
NXD float offset_axis(float p, float n)
{
    const int ofi = (int)(256.0f * n);
    const float pi = __int_as_float(__float_as_int(p) + ((p < 0.0f) ? -ofi : ofi));
    return fabsf(p) < (1.0f / 32.0f) ? p + (1.0f / 65536.0f) * n : pi;
}
NXD f3 offset_ray(f3 p, f3 n) { return f3{offset_axis(p.x, n.x), offset_axis(p.y, n.y), offset_axis(p.z, n.z)}; }

}  // namespace nxd
