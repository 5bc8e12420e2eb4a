// nx_traverse.h — the pieces of the BVH8 traversal shared by the persistent trace kernels (nx_trace.hip) and the tail kernel
// (nx_wavefront.hip): record fetch, the LDS / scratch stack, ChildTrace, and traverse_wave, the same traversal step run for
// the 64 rays of one wave without refill.
#pragma once
#include "nx_device.h"
#include "nx_math.h"

namespace nxd {

#ifndef NX_TRACE_BLOCK
#define NX_TRACE_BLOCK 256
#endif
constexpr int kTraceBlock = NX_TRACE_BLOCK;  // 4 waves (NX_TRACE_BLOCK: measurement knob, 64 / 128: DESIGN.md section 7)
#ifndef NX_LDS_DEPTH
#define NX_LDS_DEPTH 8
#endif
constexpr int kLdsDepth = NX_LDS_DEPTH;        // stack entries per lane held in LDS (2 KiB each per workgroup)
constexpr int kSpillDepth = 32 - kLdsDepth;  // further entries in scratch; 32 in total as the reference (BVH8Traversal.cuh:17)
using GU4 = const NX_G uint4*;   // global-memory pointers: global_load_dwordx4, never flat
using GF4 = const NX_G float4*;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;  // one stack entry (uint2) as a 64-bit scalar
typedef __attribute__((address_space(3))) float lds_f32;

// Record fetch, ONE per loop iteration for all record kinds: every lane issues the 16-byte global loads of its own record (3
// or 5 in flight) and the wave waits once.  Must be reached by all lanes of the wave.  (A cooperative LDS-staged form — lanes
// publish addresses, the wave streams the records into LDS with global_load ... lds of consecutive chunks — measured slower
// in round 1 and was removed: tools/micro/gather.hip shows the L1's cost is per lane-load, coalesced or not.)
NXD void fetch_record(bool kind5, bool kind3, unsigned long long addr, uint4 (&out)[5])
{
    if (kind5 || kind3) {
        GU4 p = (GU4)addr;
        out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
        if (kind5) { out[3] = p[3]; out[4] = p[4]; }
    }
}

// Traversal stack: entries [0, kLdsDepth) live in LDS (entry-major: lane stride 1, depth stride kTraceBlock), the rest in
// a scratch array.  The stack pointer and the LDS base stay in registers: they are deliberately NOT members of a struct
// together with the scratch array (a struct holding a dynamically indexed array is kept in scratch as a whole, which
// turned every push / pop into scratch loads of its own stack pointer).
// (STRIDE: the distance between two entries of a lane — the workgroup's 256 lanes side by side in the trace and tail kernels; the
//  thin kernel's replay keeps a wave's 64 in that wave's own pool)
template <int STRIDE = kTraceBlock>
NXD void stack_push(lds_u64* lds, uint2* spill, int& sp, uint2 e)
{
    if (sp < kLdsDepth) lds[sp * STRIDE] = ((unsigned long long)e.y << 32) | e.x;
    else if (sp < kLdsDepth + kSpillDepth) spill[sp - kLdsDepth] = e;
    sp++;
}
template <int STRIDE = kTraceBlock>
NXD uint2 stack_pop(lds_u64* lds, const uint2* spill, int& sp)
{
    sp--;
    if (sp < kLdsDepth) {
        const unsigned long long v = lds[sp * STRIDE];
        return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    }
    if (sp < kLdsDepth + kSpillDepth) return spill[sp - kLdsDepth];
    return make_uint2(0u, 0u);
}

// Entering an instance whose inverse transform is exactly the identity leaves a ray as it is, bit for bit — unless one of its
// components is a zero (the transform's `+ 0 * y` terms
// can flip the sign of a zero, and 1 / dir follows the sign), an infinity or a NaN (0 * inf), or a denormal.  Such rays are
// rare (axis-parallel ones) and take the general path, which computes what the reference computes (BVH8Traversal.cuh:259-264);
// all others skip the two matrix products, the six comparisons of the old "did anything change" test and the LDS parking.
NXD bool ray_is_ordinary(f3 o, f3 d)
{
    constexpr int kSpecial = 0x003 | 0x004 | 0x200 | 0x020 | 0x040 | 0x010 | 0x080;  // NaN, +-inf, +-0, +-denormal
    // (bitwise |: six v_cmp_class and five scalar ORs; the short-circuit form compiles to a chain of branches)
    return !((int)__builtin_amdgcn_classf(o.x, kSpecial) | (int)__builtin_amdgcn_classf(o.y, kSpecial) | (int)__builtin_amdgcn_classf(o.z, kSpecial) |
             (int)__builtin_amdgcn_classf(d.x, kSpecial) | (int)__builtin_amdgcn_classf(d.y, kSpecial) | (int)__builtin_amdgcn_classf(d.z, kSpecial));
}

NXD float ubyte_f(uint32_t x, int j) { return (float)((x >> (8 * j)) & 0xffu); }  // v_cvt_f32_ubyte{j}
NXD int imax3(int a, int b, int c) { return max(max(a, b), c); }                  // v_max3_i32
NXD int imin3(int a, int b, int c) { return min(min(a, b), c); }                  // v_min3_i32

// The part of an instance record in front of its BLAS root copy.  In a scene whose instances all carry the identity (a uniform
// flag of the device state, DeviceState::sceneFlags) the three transform rows are not even loaded.
struct InstFetch {
    uint4 rows[3];
    uint4 ptrs;
    uint32_t instIdx;
    NXD GU4 nodes() const { return (GU4)(((unsigned long long)ptrs.y << 32) | ptrs.x); }
    NXD GF4 isect() const { return (GF4)(((unsigned long long)ptrs.w << 32) | ptrs.z); }
};
NXD void fetch_instance(unsigned long long recAddr, bool sceneIdentity, InstFetch& f)
{
    GU4 p = (GU4)recAddr;
    if (!sceneIdentity) { f.rows[0] = p[0]; f.rows[1] = p[1]; f.rows[2] = p[2]; }
    f.ptrs = p[3];
    f.instIdx = *(const NX_G uint32_t*)(p + 4);
}
// The ray in the instance's frame (BVH8Traversal.cuh:259-264: origin and direction through the inverse transform, not
// renormalised).  Returns whether it differs from the incoming ray; false leaves the caller's ray, 1 / dir and (on exit) the
// restore untouched.  The comparison is on bit patterns: a zero whose sign the transform flips counts.
// Scene of identity instances only: an ordinary ray is its own image (nothing of the record is needed for that decision, the
// two matrix products and the six comparisons are skipped); the others go through the identity's rows as constants.
// Any other scene: the general path for every entry — an identity instance among transformed ones is recognised by the
// comparison as before.  (A per-record flag test in front of it was measured: it needs the record's flag word before the
// first instruction of the step, and on scenes without identity instances — configs[3] / configs[4] — cost 3 %.)
NXD bool enter_instance(const InstFetch& f, bool sceneIdentity, f3 org, f3 dir, f3& o2, f3& d2)
{
    float4 r0 = make_float4(1.0f, 0.0f, 0.0f, 0.0f), r1 = make_float4(0.0f, 1.0f, 0.0f, 0.0f), r2 = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
    if (sceneIdentity) {
#ifndef NX_NO_IDENTITY_FLAG
        if (ray_is_ordinary(org, dir)) return false;
#endif
    } else {
        r0 = make_float4(__uint_as_float(f.rows[0].x), __uint_as_float(f.rows[0].y), __uint_as_float(f.rows[0].z), __uint_as_float(f.rows[0].w));
        r1 = make_float4(__uint_as_float(f.rows[1].x), __uint_as_float(f.rows[1].y), __uint_as_float(f.rows[1].z), __uint_as_float(f.rows[1].w));
        r2 = make_float4(__uint_as_float(f.rows[2].x), __uint_as_float(f.rows[2].y), __uint_as_float(f.rows[2].z), __uint_as_float(f.rows[2].w));
    }
    o2 = mat_point(r0, r1, r2, org);
    d2 = mat_vec(r0, r1, r2, dir);
    return ((__float_as_uint(o2.x) ^ __float_as_uint(org.x)) | (__float_as_uint(o2.y) ^ __float_as_uint(org.y)) | (__float_as_uint(o2.z) ^ __float_as_uint(org.z)) |
            (__float_as_uint(d2.x) ^ __float_as_uint(dir.x)) | (__float_as_uint(d2.y) ^ __float_as_uint(dir.y)) | (__float_as_uint(d2.z) ^ __float_as_uint(dir.z))) != 0u;
}

// ChildTrace — BVH8Traversal.cuh:55-146
NXD void child_trace(const uint4 (&nd)[5], f3 org, f3 dir, f3 idir, uint32_t invOct4, float tmaxRay, uint2& ng, uint2& tg)
{
    const uint4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3], n4 = nd[4];

    const f3 p = mk3(__uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z));
    const uint32_t e_imask = n0.w;
    const f3 tdir = mk3(__uint_as_float((e_imask & 0xffu) << 23) * idir.x, __uint_as_float((e_imask << 15) & 0x7f800000u) * idir.y,
                        __uint_as_float((e_imask << 7) & 0x7f800000u) * idir.z);
    const f3 torg = (p - org) * idir;
    const bool nx = dir.x < 0.0f, ny = dir.y < 0.0f, nz = dir.z < 0.0f;

    uint32_t hitMask = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t meta4 = i ? n1.w : n1.z;
        const uint32_t isInner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
        const uint32_t innerMask4 = (isInner4 >> 4) * 0xffu;  // per byte 0xff where inner (reference: prmt sign extension)
        const uint32_t bitIndex4 = (meta4 ^ (invOct4 & innerMask4)) & 0x1f1f1f1fu;
        const uint32_t childBits4 = (meta4 >> 5) & 0x07070707u;

        const uint32_t qlox = i ? n2.y : n2.x, qloy = i ? n2.w : n2.z, qloz = i ? n3.y : n3.x;
        const uint32_t qhix = i ? n3.w : n3.z, qhiy = i ? n4.y : n4.x, qhiz = i ? n4.w : n4.z;
        const uint32_t xMin = nx ? qhix : qlox, xMax = nx ? qlox : qhix;
        const uint32_t yMin = ny ? qhiy : qloy, yMax = ny ? qloy : qhiy;
        const uint32_t zMin = nz ? qhiz : qloz, zMax = nz ? qloz : qhiz;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float tminx = fmaf(ubyte_f(xMin, j), tdir.x, torg.x);
            const float tminy = fmaf(ubyte_f(yMin, j), tdir.y, torg.y);
            const float tminz = fmaf(ubyte_f(zMin, j), tdir.z, torg.z);
            const float tmaxx = fmaf(ubyte_f(xMax, j), tdir.x, torg.x);
            const float tmaxy = fmaf(ubyte_f(yMax, j), tdir.y, torg.y);
            const float tmaxz = fmaf(ubyte_f(zMax, j), tdir.z, torg.z);
            const float tmin = __int_as_float(imax3(__float_as_int(tminx), __float_as_int(tminy), __float_as_int(fmaxf(tminz, 0.0f))));
            const float tmax = __int_as_float(imin3(__float_as_int(tmaxx), __float_as_int(tmaxy), __float_as_int(fminf(tmaxz, tmaxRay))));
            const uint32_t childBits = (childBits4 >> (8 * j)) & 0xffu;
            const uint32_t bitIndex = (bitIndex4 >> (8 * j)) & 0xffu;
            hitMask |= (tmin <= tmax) ? (childBits << bitIndex) : 0u;
        }
    }
    ng = make_uint2(n1.x, (hitMask & 0xff000000u) | (e_imask >> 24));
    tg = make_uint2(n1.y, hitMask & 0x00ffffffu);
}

// ChildTrace for the thin kernel's closest-hit search (nx_trace.hip thin_wave_search): the SAME slab arithmetic as child_trace
// above, child by child in storage order, but what comes out is which children the ray enters (bit c of `entered`: child_trace's
// `tmin <= tmax`) and every child's computed entry distance tminOut[c] — the very number that test compares — instead of the two
// hit masks.  The search carries the largest entry distance along a triangle's path from the root: its "gate".
NXD void child_trace_gate(const uint4 (&nd)[5], f3 org, f3 dir, f3 idir, float tmaxRay, uint32_t& entered, float (&tminOut)[8])
{
    const uint4 n0 = nd[0], n2 = nd[2], n3 = nd[3], n4 = nd[4];
    const f3 p = mk3(__uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z));
    const uint32_t e_imask = n0.w;
    const f3 tdir = mk3(__uint_as_float((e_imask & 0xffu) << 23) * idir.x, __uint_as_float((e_imask << 15) & 0x7f800000u) * idir.y,
                        __uint_as_float((e_imask << 7) & 0x7f800000u) * idir.z);
    const f3 torg = (p - org) * idir;
    const bool nx = dir.x < 0.0f, ny = dir.y < 0.0f, nz = dir.z < 0.0f;
    entered = 0u;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t qlox = i ? n2.y : n2.x, qloy = i ? n2.w : n2.z, qloz = i ? n3.y : n3.x;
        const uint32_t qhix = i ? n3.w : n3.z, qhiy = i ? n4.y : n4.x, qhiz = i ? n4.w : n4.z;
        const uint32_t xMin = nx ? qhix : qlox, xMax = nx ? qlox : qhix;
        const uint32_t yMin = ny ? qhiy : qloy, yMax = ny ? qloy : qhiy;
        const uint32_t zMin = nz ? qhiz : qloz, zMax = nz ? qloz : qhiz;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float tminx = fmaf(ubyte_f(xMin, j), tdir.x, torg.x);
            const float tminy = fmaf(ubyte_f(yMin, j), tdir.y, torg.y);
            const float tminz = fmaf(ubyte_f(zMin, j), tdir.z, torg.z);
            const float tmaxx = fmaf(ubyte_f(xMax, j), tdir.x, torg.x);
            const float tmaxy = fmaf(ubyte_f(yMax, j), tdir.y, torg.y);
            const float tmaxz = fmaf(ubyte_f(zMax, j), tdir.z, torg.z);
            const float tmin = __int_as_float(imax3(__float_as_int(tminx), __float_as_int(tminy), __float_as_int(fmaxf(tminz, 0.0f))));
            const float tmax = __int_as_float(imin3(__float_as_int(tmaxx), __float_as_int(tmaxy), __float_as_int(fminf(tmaxz, tmaxRay))));
            tminOut[4 * i + j] = tmin;
            entered |= (tmin <= tmax) ? (1u << (4 * i + j)) : 0u;
        }
    }
}

// The traversal of nx_trace.hip's trace_kernel for ONE ray per lane, run until every lane of the wave is done (no queue, no
// refill): the tail kernel traces its continuation and shadow rays with it.  The loop body is the same step — pop / retire,
// one record per iteration (node, instance + BLAS root, or triangle), same order, same arithmetic — so a ray's result is
// the one the persistent kernel would produce (tests/test_gpu_tail.py compares the two pipelines bit for bit).
// `valid`: this lane has a ray.  Closest hit: hitT (1e30 = miss), hitU, hitV, hitTri, hitInst.  Any hit: returns whether the
// ray is occluded within hitT.  Must be called by all 64 lanes.
template <bool ANY_HIT, int STRIDE = kTraceBlock>
NXD bool traverse_wave(const DeviceState* __restrict__ S, lds_u64* stackLds, bool valid, f3 org, f3 dir, float& hitT, float& hitU, float& hitV,
                       uint32_t& hitTri, uint32_t& hitInst)
{
    GU4 tlasNodes = S->tlasNodes;
    const NX_G InstTrav* instTrav = S->instTrav;
    const bool sceneIdentity = (S->sceneFlags & kSceneAllIdentity) != 0u;
    const f3 worldOrg = org, worldDir = dir;
    uint2 stackSpill[kSpillDepth];
    int sp = 0;
    bool active = valid;
    bool occluded = false;
    f3 idir = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    if (!ANY_HIT) hitT = 1e30f;
    hitU = 0.0f; hitV = 0.0f; hitTri = 0xffffffffu; hitInst = 0xffffffffu;
    const uint32_t oct = ((dir.x < 0.0f ? 1u : 0u) << 2) | ((dir.y < 0.0f ? 1u : 0u) << 1) | (dir.z < 0.0f ? 1u : 0u);
    const uint32_t invOct4 = (7u - oct) * 0x01010101u;
    uint2 ng = make_uint2(0u, 0x80000000u), tg = make_uint2(0u, 0u);
    int instSp = -1;
    bool xformed = false;
    uint32_t instIdx = 0;
    GU4 nodes = tlasNodes;
    GF4 isect = nullptr;
    uint32_t spins = 0u;  // iterations of this call: 64 rays, none of which legitimately takes anywhere near kStallLimit
    while (__ballot(active) != 0ull) {
        if (++spins > kStallLimit) {
            if ((threadIdx.x & (kWave - 1)) == 0) atomicOr(&S->frame->errorWord, kErrTraversalStalled);
            break;
        }
        if (active && tg.y == 0u && (ng.y & 0xff000000u) == 0u) {
            if (sp == 0) {
                active = false;
            } else {
                if (sp == instSp) {
                    if (xformed) {
                        org = worldOrg;
                        dir = worldDir;
                        idir = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
                    }
                    nodes = tlasNodes;
                    instSp = -1;
                }
                const uint2 e = stack_pop<STRIDE>(stackLds, stackSpill, sp);
                if (e.y & 0xff000000u) ng = e;
                else { tg = e; ng = make_uint2(0u, 0u); }
            }
        }
        const bool wantNode = active && tg.y == 0u && (ng.y & 0xff000000u) != 0u;
        const bool wantInst = active && tg.y != 0u && instSp < 0;
        const bool wantTri = active && tg.y != 0u && instSp >= 0;
        unsigned long long recAddr = 0ull;
        if (wantNode) {
            const int nodeOffset = 31 - __clz((int)ng.y);
            ng.y &= ~(1u << nodeOffset);
            if (ng.y & 0xff000000u) stack_push<STRIDE>(stackLds, stackSpill, sp, ng);
            const int slot = (nodeOffset - 24) ^ (int)(invOct4 & 7u);
            const int rel = __popc(ng.y & ~(0xffffffffu << slot));
            recAddr = (unsigned long long)(nodes + (size_t)(ng.x + (uint32_t)rel) * (unsigned)kNodeStride);
        } else if (wantInst) {
            const int off = 31 - __clz((int)tg.y);
            tg.y &= ~(1u << off);
            recAddr = (unsigned long long)&instTrav[tg.x + (uint32_t)off];
            if (tg.y) stack_push<STRIDE>(stackLds, stackSpill, sp, tg);
            if (ng.y & 0xff000000u) stack_push<STRIDE>(stackLds, stackSpill, sp, ng);
            instSp = sp;
        } else if (wantTri) {
            const int off = 31 - __clz((int)tg.y);
            tg.y &= ~(1u << off);
            recAddr = (unsigned long long)(isect + (size_t)(tg.x + (uint32_t)off) * (unsigned)kTriStride);
        }
        uint4 rc[5];
        InstFetch fi;
        if (wantInst) fetch_instance(recAddr, sceneIdentity, fi);
        fetch_record(wantNode || wantInst, wantTri, recAddr + (wantInst ? 80ull : 0ull), rc);
        if (wantInst) {
            nodes = fi.nodes();
            isect = fi.isect();
            instIdx = fi.instIdx;
            f3 o2, d2;
            xformed = enter_instance(fi, sceneIdentity, org, dir, o2, d2);
            if (xformed) {
                org = o2;
                dir = d2;
                idir = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
            }
        }
        if (wantNode || wantInst) child_trace(rc, org, dir, idir, invOct4, hitT, ng, tg);
        if (wantTri) {
            const f3 p0 = mk3(__uint_as_float(rc[0].x), __uint_as_float(rc[0].y), __uint_as_float(rc[0].z));
            const f3 edge0 = mk3(__uint_as_float(rc[1].x), __uint_as_float(rc[1].y), __uint_as_float(rc[1].z));
            const f3 edge1 = mk3(__uint_as_float(rc[2].x), __uint_as_float(rc[2].y), __uint_as_float(rc[2].z));
            const f3 rayCrossEdge1 = cross3(dir, edge1);
            const float det = dot3(edge0, rayCrossEdge1);
            const float invDet = 1.0f / det;
            const f3 s = org - p0;
            const float u = invDet * dot3(s, rayCrossEdge1);
            const f3 sCrossEdge0 = cross3(s, edge0);
            const float v = invDet * dot3(dir, sCrossEdge0);
            const float t = invDet * dot3(edge1, sCrossEdge0);
            const bool hit = !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) && (t > 0.0f && t < hitT);
            if (hit) {
                if (ANY_HIT) {
                    active = false;
                    occluded = true;
                } else {
                    hitT = t; hitU = u; hitV = v;
                    hitTri = rc[0].w;
                    hitInst = instIdx;
                }
            }
        }
    }
    // (an instance record's index word carries the instance's material code above kHitCodeShift: nx_device.h)
    if (!ANY_HIT && hitTri != 0xffffffffu) hitInst &= kHitInstMask;
    return occluded;
}

}  // namespace nxd
