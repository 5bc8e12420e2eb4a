// nx_instbox.h — the world-space box of an instance as the device-side TLAS builder and refit take it: not the transformed BLAS
// root *frame* [p, p + 255 * 2^e] of BVHInstance::SetTransform (the reference's rule, BVHInstance.cpp:4-21: up to twice the
// mesh's extent per axis, because the frame's scale is a power of two) but what the frame holds — the boxes of the root node's
// children and grandchildren, each through the instance transform — intersected with the record's box.  Both contain the geometry, so does
// their intersection.  Shared by nx_lbvh.hip (nxhip_rebuild_tlas) and nx_refit.hip (nxhip_set_instance_transforms on a
// device-built TLAS).
#pragma once
#include "nx_device.h"

namespace nxd {

struct InstBox {
    float lo[3], hi[3];
};

// the boxes of `node`'s children through T, grown into t; returns whether it had any
__device__ __forceinline__ bool grow_by_children(const nx_bvh8_node& node, const float* T, InstBox& t, uint32_t skipSlots)
{
    float sc[3];
    for (int a = 0; a < 3; a++) sc[a] = __uint_as_float((uint32_t)node.e[a] << 23);
    bool any = false;
    for (int s = 0; s < 8; s++) {
        if (node.meta[s] == 0 || ((skipSlots >> s) & 1u)) continue;
        any = true;
        const float lo[3] = {node.p[0] + (float)node.qlox[s] * sc[0], node.p[1] + (float)node.qloy[s] * sc[1], node.p[2] + (float)node.qloz[s] * sc[2]};
        const float hi[3] = {node.p[0] + (float)node.qhix[s] * sc[0], node.p[1] + (float)node.qhiy[s] * sc[1], node.p[2] + (float)node.qhiz[s] * sc[2]};
        for (int corner = 0; corner < 8; corner++) {
            const float cx = (corner & 1) ? hi[0] : lo[0], cy = (corner & 2) ? hi[1] : lo[1], cz = (corner & 4) ? hi[2] : lo[2];
            const float w[3] = {T[0] * cx + T[1] * cy + T[2] * cz + T[3], T[4] * cx + T[5] * cy + T[6] * cz + T[7], T[8] * cx + T[9] * cy + T[10] * cz + T[11]};
            for (int a = 0; a < 3; a++) { t.lo[a] = fminf(t.lo[a], w[a]); t.hi[a] = fmaxf(t.hi[a], w[a]); }
        }
    }
    return any;
}

// `nodes`: the BLAS's node array (80-byte nodes, unpadded; node 0 is the root); `T`: the instance's object-to-world matrix
// (row-major 4 x 4); `b`: in: the record's box, out: the tightened one.  Two levels: the root's leaf children by their own
// boxes, its inner children by the boxes of THEIR children (up to 64 boxes: a rotated mesh is hugged more closely by many small
// boxes than by eight large ones).
// `singular`: the instance is traversed with the identity as its inverse transform — Mat4::Inverted's fallback for a matrix whose
// 4 x 4 cofactor determinant is exactly zero (a mesh flattened by a zero scale, a projection) — which places the hits where no
// box derived from T is: such an instance keeps the record's box.  The caller takes the decision from the SAME computation the
// traversal's inverse comes from (the refit: mat4_invert's own determinant test; the builder: the record's inverse IS the
// fallback), not from a determinant of its own, whose rounding could differ on nearly singular matrices.
template <class NodePtr>
__device__ __forceinline__ void tighten_instance_box(NodePtr nodes, const float* T, InstBox& b, const bool singular)
{
    if (singular) return;
    const nx_bvh8_node root = nodes[0];
    InstBox t;
    for (int a = 0; a < 3; a++) { t.lo[a] = 1e30f; t.hi[a] = -1e30f; }
    bool any = false;
    uint32_t inner = 0;
    for (int s = 0; s < 8; s++) {
        if (!((root.imask >> s) & 1u) || root.meta[s] == 0) continue;
        const nx_bvh8_node child = nodes[root.childBaseIdx + (uint32_t)__popc(root.imask & ((1u << s) - 1u))];
        if (grow_by_children(child, T, t, 0u)) { any = true; inner |= 1u << s; }  // (a childless inner node: its own box below)
    }
    any = grow_by_children(root, T, t, inner) || any;
    bool finite = any;
    for (int a = 0; a < 3; a++) finite = finite && t.lo[a] > -1.0e9f && t.hi[a] < 1.0e9f && t.lo[a] <= t.hi[a];
    if (!finite) return;
    for (int a = 0; a < 3; a++) {
        // the decoded and transformed corners carry a rounding each way: a margin of a few ulps keeps the box conservative
        const float pad = 4.0e-7f * fmaxf(fmaxf(fabsf(t.lo[a]), fabsf(t.hi[a])), t.hi[a] - t.lo[a]);
        const float lo = fmaxf(b.lo[a], t.lo[a] - pad), hi = fminf(b.hi[a], t.hi[a] + pad);
        if (lo <= hi) { b.lo[a] = lo; b.hi[a] = hi; }
    }
}

// rows 0..3 of a row-major 4 x 4 matrix are the identity's, bit for bit
__device__ __forceinline__ bool mat4_is_identity(const float* m)
{
    return rows_are_identity(m) && __float_as_uint(m[12]) == 0u && __float_as_uint(m[13]) == 0u && __float_as_uint(m[14]) == 0u && __float_as_uint(m[15]) == 0x3f800000u;
}

}  // namespace nxd
