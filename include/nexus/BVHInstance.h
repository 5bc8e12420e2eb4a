// nexus/BVHInstance.h — an instance of a BLAS: transform, inverse, world bounds, material.
// Mirrors /root/reference/Nexus/src/Geometry/BVH/BVHInstance.h:11-38, BVHInstance.cpp:4-45.
#pragma once

#include "BVH8.h"
#include "Math.h"

namespace nexus {

class BVHInstance {
public:
    BVHInstance() = default;
    BVHInstance(unsigned int blasIdx, const BVH8* bvh) : m_BvhIdx(blasIdx), m_Bvh(bvh)
    {
        Mat4 m;
        SetTransform(m);
    }

    void SetTransform(const Mat4& t);
    void SetTransform(float3 pos, float3 rotationDegrees, float3 scale);
    const AABB& GetBounds() const { return m_Bounds; }
    void AssignMaterial(int mIdx) { m_MaterialId = mIdx; }
    void SetBvh(const BVH8* bvh) { m_Bvh = bvh; }  // the owning vector may have been re-allocated
    int GetMaterialId() const { return m_MaterialId; }
    unsigned int GetBvhIdx() const { return m_BvhIdx; }
    const Mat4& GetTransform() const { return m_Transform; }
    const Mat4& GetInvTransform() const { return m_InvTransform; }

    static nx_bvh_instance ToDevice(const BVHInstance& inst);

private:
    unsigned int m_BvhIdx = 0;
    const BVH8* m_Bvh = nullptr;
    Mat4 m_InvTransform;
    Mat4 m_Transform;
    AABB m_Bounds;
    int m_MaterialId = 0;
};

}  // namespace nexus
