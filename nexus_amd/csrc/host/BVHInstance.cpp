// BVHInstance.cpp — see include/nexus/BVHInstance.h.
// World bounds follow /root/reference/Nexus/src/Geometry/BVH/BVHInstance.cpp:4-21 exactly, because the TLAS (and with it
// every traversal order) is built from them: they are the transformed corners of the BLAS root node's *quantisation frame*
// [p, p + 2^(e-127) * 255], which is looser than the mesh's exact box.
#include "nexus/BVHInstance.h"

namespace nexus {

namespace {

// the root node's decode frame of a BLAS: lower corner and per-axis extent 2^(e-127) * (2^8 - 1)
void root_frame(const BVH8& blas, float3& lo, float3& hi)
{
    const BVH8Node& root = blas.nodes[0];
    const float steps = std::exp2(8.0f) - 1.0f;
    lo = make_float3(root.p);
    hi = lo + make_float3(std::exp2(static_cast<float>(root.e[0] - 127)), std::exp2(static_cast<float>(root.e[1] - 127)),
                          std::exp2(static_cast<float>(root.e[2] - 127))) * steps;
}

Mat4 to_mat4(const nx_mat4& m)
{
    Mat4 r;
    std::memcpy(r.cell, m.cell, sizeof r.cell);
    return r;
}

}  // namespace

BVHInstance::BVHInstance()
{
    std::memset(&m_Record, 0, sizeof m_Record);
    const Mat4 identity;
    std::memcpy(m_Record.transform.cell, identity.cell, sizeof identity.cell);
    std::memcpy(m_Record.invTransform.cell, identity.cell, sizeof identity.cell);
}

BVHInstance::BVHInstance(unsigned int blasIdx, const BVH8* blas) : BVHInstance()
{
    m_Record.bvhIdx = blasIdx;
    m_Blas = blas;
    SetTransform(Mat4::Identity());
}

Mat4 BVHInstance::GetTransform() const { return to_mat4(m_Record.transform); }
Mat4 BVHInstance::GetInvTransform() const { return to_mat4(m_Record.invTransform); }

void BVHInstance::SetTransform(float3 pos, float3 rotationDegrees, float3 scale)
{
    using Utils::ToRadians;
    // evaluated left to right, as the reference's expression is: ((((T * Rz) * Ry) * Rx) * S)
    Mat4 m = Mat4::Translate(pos);
    m = m * Mat4::RotateZ(ToRadians(rotationDegrees.z));
    m = m * Mat4::RotateY(ToRadians(rotationDegrees.y));
    m = m * Mat4::RotateX(ToRadians(rotationDegrees.x));
    SetTransform(m * Mat4::Scale(scale));
}

void BVHInstance::SetTransform(const Mat4& objectToWorld)
{
    const Mat4 worldToObject = objectToWorld.Inverted();
    std::memcpy(m_Record.transform.cell, objectToWorld.cell, sizeof objectToWorld.cell);
    std::memcpy(m_Record.invTransform.cell, worldToObject.cell, sizeof worldToObject.cell);

    float3 lo, hi;
    root_frame(*m_Blas, lo, hi);
    m_WorldBounds = AABB();
    for (int corner = 0; corner < 8; corner++) {  // corner bit 0 / 1 / 2 selects hi on x / y / z
        const float3 c = make_float3((corner & 1) ? hi.x : lo.x, (corner & 2) ? hi.y : lo.y, (corner & 4) ? hi.z : lo.z);
        m_WorldBounds.Grow(TransformPosition(c, objectToWorld));
    }
    store(m_Record.boundsMin, m_WorldBounds.bMin);
    store(m_Record.boundsMax, m_WorldBounds.bMax);
}

}  // namespace nexus
