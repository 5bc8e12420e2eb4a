// BVHInstance.cpp — instance transform and world bounds.
// Semantics of /root/reference/Nexus/src/Geometry/BVH/BVHInstance.cpp:4-45: the world AABB is the
// transform of the BLAS *root node's quantisation frame* (p, p + 2^(e-127) * 255), i.e. looser than the
// exact bounds; rotation order Translate * Rz * Ry * Rx * Scale with angles in degrees.
#include "nexus/BVHInstance.h"

namespace nexus {

void BVHInstance::SetTransform(const Mat4& t)
{
    m_Transform = t;
    m_InvTransform = t.Inverted();
    const BVH8Node& root = m_Bvh->nodes[0];
    const float3 bMin = make_float3(root.p);
    const float3 bMax = bMin + make_float3(std::exp2(static_cast<float>(root.e[0] - 127)), std::exp2(static_cast<float>(root.e[1] - 127)),
                                           std::exp2(static_cast<float>(root.e[2] - 127))) * (std::exp2(8.0f) - 1.0f);
    m_Bounds = AABB();
    for (int i = 0; i < 8; i++)
        m_Bounds.Grow(TransformPosition(make_float3(i & 1 ? bMax.x : bMin.x, i & 2 ? bMax.y : bMin.y, i & 4 ? bMax.z : bMin.z), t));
}

void BVHInstance::SetTransform(float3 pos, float3 r, float3 s)
{
    const Mat4 t = Mat4::Translate(pos) * Mat4::RotateZ(Utils::ToRadians(r.z)) * Mat4::RotateY(Utils::ToRadians(r.y)) *
                   Mat4::RotateX(Utils::ToRadians(r.x)) * Mat4::Scale(s);
    SetTransform(t);
}

nx_bvh_instance BVHInstance::ToDevice(const BVHInstance& inst)
{
    nx_bvh_instance d;
    std::memset(&d, 0, sizeof d);
    d.bvhIdx = inst.m_BvhIdx;
    std::memcpy(d.invTransform.cell, inst.m_InvTransform.cell, 64);
    std::memcpy(d.transform.cell, inst.m_Transform.cell, 64);
    store(d.boundsMin, inst.m_Bounds.bMin);
    store(d.boundsMax, inst.m_Bounds.bMax);
    d.materialId = inst.m_MaterialId;
    return d;
}

}  // namespace nexus
