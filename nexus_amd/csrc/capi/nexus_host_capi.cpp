// nexus_host_capi.cpp — flat C wrappers over the C++ host classes (see include/nexus_host.h).
#include "nexus_host.h"

#include "Collapse.h"

#include <cstring>
#include <new>
#include <vector>

#include "nexus/BVH.h"
#include "nexus/BVH8Builder.h"
#include "nexus/BVHInstance.h"
#include "nexus/Camera.h"
#include "nexus/TLAS.h"

using namespace nexus;

struct nxh_bvh8 {
    BVH8 bvh;
};

namespace {
std::vector<Triangle> to_triangles(const nx_triangle* tris, uint32_t n)
{
    std::vector<Triangle> v;
    v.reserve(n);
    for (uint32_t i = 0; i < n; i++) v.emplace_back(tris[i]);
    return v;
}
Mat4 to_mat4(const float* m)
{
    Mat4 r;
    std::memcpy(r.cell, m, 64);
    return r;
}
}  // namespace

extern "C" {

int nxh_bvh8_build(const nx_triangle* tris, uint32_t triCount, uint32_t threads, nxh_bvh8** out)
{
    if (!tris || triCount == 0 || !out) return 1;
    try {
        BVH8Builder builder(to_triangles(tris, triCount));
        builder.Init(threads);
        nxh_bvh8* h = new nxh_bvh8();
        h->bvh = builder.Build();
        h->bvh.triangles.clear();  // the caller owns the triangle array
        h->bvh.triangles.shrink_to_fit();
        *out = h;
        return 0;
    } catch (const std::bad_alloc&) {
        return 2;
    }
}

int nxh_tlas_build(const nx_bvh_instance* instances, uint32_t instanceCount, nxh_bvh8** out)
{
    if (!instances || instanceCount == 0 || !out) return 1;
    try {
        TLAS tlas;
        tlas.BuildFromBounds(instances, instanceCount);
        tlas.Convert();
        nxh_bvh8* h = new nxh_bvh8();
        h->bvh = std::move(tlas.bvh8);
        *out = h;
        return 0;
    } catch (const std::bad_alloc&) {
        return 2;
    }
}

int nxh_tlas_refit(nx_bvh8_node* nodes, uint32_t nodeCount, const uint32_t* instanceIdx, const nx_bvh_instance* instances, uint32_t instanceCount)
{
    if (!nodes || nodeCount == 0 || !instanceIdx || !instances || instanceCount == 0) return 1;
    try {
        // the same structural checks as nxhip_set_tlas: a refit must not read outside the arrays it was given
        for (uint32_t i = 0; i < instanceCount; i++)
            if (instanceIdx[i] >= instanceCount) return 1;
        for (uint32_t i = 0; i < nodeCount; i++) {
            const nx_bvh8_node& n = nodes[i];
            int inner = 0, prims = 0;
            for (int s = 0; s < 8; s++) {
                if (n.imask & (1u << s)) inner++;
                else if (n.meta[s]) prims = std::max(prims, (n.meta[s] & 0x1f) + __builtin_popcount(n.meta[s] >> 5));
            }
            if (inner && (n.childBaseIdx <= i || static_cast<uint64_t>(n.childBaseIdx) + inner > nodeCount)) return 1;
            if (prims && static_cast<uint64_t>(n.triangleBaseIdx) + prims > instanceCount) return 1;
        }
        std::vector<AABB> bounds;
        bounds.reserve(instanceCount);
        for (uint32_t i = 0; i < instanceCount; i++) bounds.emplace_back(make_float3(instances[i].boundsMin), make_float3(instances[i].boundsMax));
        std::vector<BVH8Node> v(nodes, nodes + nodeCount);
        collapse::Refit(v, instanceIdx, bounds.data());
        std::memcpy(nodes, v.data(), sizeof(nx_bvh8_node) * nodeCount);
        return 0;
    } catch (const std::bad_alloc&) {
        return 2;
    }
}

uint32_t nxh_bvh8_node_count(const nxh_bvh8* b) { return b ? static_cast<uint32_t>(b->bvh.nodes.size()) : 0; }
uint32_t nxh_bvh8_prim_count(const nxh_bvh8* b) { return b ? static_cast<uint32_t>(b->bvh.triangleIdx.size()) : 0; }
const nx_bvh8_node* nxh_bvh8_nodes(const nxh_bvh8* b) { return b ? b->bvh.nodes.data() : nullptr; }
const uint32_t* nxh_bvh8_prim_indices(const nxh_bvh8* b) { return b ? b->bvh.triangleIdx.data() : nullptr; }
void nxh_bvh8_free(nxh_bvh8* b) { delete b; }

int nxh_bvh2_build(const nx_triangle* tris, uint32_t triCount, uint32_t threads, void* nodes32, uint32_t* triIdx)
{
    if (!tris || triCount == 0 || !nodes32 || !triIdx) return 1;
    BVH2 bvh(to_triangles(tris, triCount));
    bvh.Build(threads);
    std::memcpy(nodes32, bvh.nodes.data(), bvh.nodes.size() * sizeof(BVH2Node));
    std::memcpy(triIdx, bvh.triangleIdx.data(), bvh.triangleIdx.size() * 4);
    return 0;
}

void nxh_mat4_from_trs(const float pos[3], const float rotDeg[3], const float scale[3], float out16[16])
{
    const Mat4 t = Mat4::Translate(make_float3(pos)) * Mat4::RotateZ(Utils::ToRadians(rotDeg[2])) * Mat4::RotateY(Utils::ToRadians(rotDeg[1])) *
                   Mat4::RotateX(Utils::ToRadians(rotDeg[0])) * Mat4::Scale(make_float3(scale));
    std::memcpy(out16, t.cell, 64);
}

void nxh_mat4_invert(const float in16[16], float out16[16])
{
    const Mat4 r = to_mat4(in16).Inverted();
    std::memcpy(out16, r.cell, 64);
}

int nxh_instance_init(nx_bvh_instance* out, uint32_t bvhIdx, int32_t materialId, const float transform16[16], const nx_bvh8_node* blasRoot)
{
    if (!out || !transform16 || !blasRoot) return 1;
    BVH8 root;
    root.nodes.push_back(*blasRoot);
    BVHInstance inst(bvhIdx, &root);
    inst.SetTransform(to_mat4(transform16));
    inst.AssignMaterial(materialId);
    *out = BVHInstance::ToDevice(inst);
    return 0;
}

int nxh_camera_init(nx_camera* out, const float position[3], const float forward[3], float hfov, uint32_t width, uint32_t height, float focusDist,
                    float defocusAngle)
{
    if (!out || !position || !forward || width == 0 || height == 0) return 1;
    const Camera cam(make_float3(position), make_float3(forward), hfov, width, height, focusDist, defocusAngle);
    *out = Camera::ToDevice(cam);
    return 0;
}

}  // extern "C"
