// nexus/TLAS.h — top-level acceleration structure over BVH instances.
// Mirrors /root/reference/Nexus/src/Geometry/BVH/TLAS.h:8-43, TLAS.cpp:13-100: agglomerative BVH2
// (Bikker), converted to a BVH8 whose "triangles" are instance ids.
#pragma once

#include <vector>

#include "BVH8.h"
#include "BVHInstance.h"

namespace nexus {

struct TLASNode {
    float3 aabbMin;
    float3 aabbMax;
    uint32_t left = 0;
    uint32_t right = 0;
    uint32_t blasCount = 0;
    uint32_t blasIdx = 0;
    bool IsLeaf() const { return left == 0; }
};

struct TLAS {
    TLAS() = default;
    explicit TLAS(const std::vector<BVHInstance>& instancesList) : bvhInstances(instancesList) {}

    void Build();
    // Same clustering from device-layout instances (only their world bounds are used).
    void BuildFromBounds(const nx_bvh_instance* instances, uint32_t count);
    void Convert();
    // Same topology, bounds from the instances' current boxes (after SetBVHInstances): O(n) instead of the O(n^2) Build().
    // Returns false (nothing done) if the instance count differs from the one the tree was built for.
    bool Refit();
    void SetBVHInstances(const std::vector<BVHInstance>& instances) { bvhInstances = instances; }
    std::vector<BVHInstance>& GetInstances() { return bvhInstances; }
    int FindBestMatch(int N, int A) const;

    std::vector<TLASNode> nodes;
    std::vector<BVHInstance> bvhInstances;
    std::vector<uint32_t> instancesIdx;
    BVH8 bvh8;

private:
    void Cluster(const std::vector<AABB>& bounds);
};

class TLASBuilder {
public:
    explicit TLASBuilder(TLAS& tlas) : m_Tlas(tlas) {}
    void Init();
    BVH8 Build();

private:
    TLAS& m_Tlas;
    std::vector<uint8_t> m_EvalStorage;
};

}  // namespace nexus
