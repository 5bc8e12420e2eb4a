// nexus/BVHInstance.h — one placement of a BLAS in the scene.
// Public methods of /root/reference/Nexus/src/Geometry/BVH/BVHInstance.h:11-38 (SetTransform x2, GetBounds, AssignMaterial,
// ToDevice).  The object keeps the 160-byte device record itself up to date, so ToDevice() is a copy and a scene upload is
// a gather of records.
#pragma once

#include "BVH8.h"
#include "Math.h"

namespace nexus {

class BVHInstance {
public:
    BVHInstance();
    // identity placement of BLAS number `blasIdx`; `blas` supplies the root quantisation frame the world bounds derive from
    BVHInstance(unsigned int blasIdx, const BVH8* blas);

    // Place the instance.  Euler angles in degrees, composed as Translate * Rz * Ry * Rx * Scale (BVHInstance.cpp:23-28).
    void SetTransform(float3 pos, float3 rotationDegrees, float3 scale);
    void SetTransform(const Mat4& objectToWorld);

    void AssignMaterial(int materialIdx) { m_Record.materialId = materialIdx; }
    int GetMaterialId() const { return m_Record.materialId; }
    unsigned int GetBvhIdx() const { return m_Record.bvhIdx; }
    // The BLAS lives in a std::vector owned by the AssetManager: re-point after that vector may have re-allocated.
    void SetBvh(const BVH8* blas) { m_Blas = blas; }

    const AABB& GetBounds() const { return m_WorldBounds; }
    Mat4 GetTransform() const;
    Mat4 GetInvTransform() const;

    static nx_bvh_instance ToDevice(const BVHInstance& inst) { return inst.m_Record; }

private:
    nx_bvh_instance m_Record;  // bvhIdx, inverse and forward transform, world bounds, material: what the kernels read
    AABB m_WorldBounds;        // same box as m_Record.boundsMin/Max, in the host's AABB type for the TLAS builder
    const BVH8* m_Blas = nullptr;
};

}  // namespace nexus
