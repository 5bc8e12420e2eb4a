// nexus/BVH.h — BVH2: binned-SAH builder with one triangle per leaf (input of the BVH8 collapse).
//
// Same algorithm and same output (node order, index order, bounds) as the reference's
// /root/reference/Nexus/src/Geometry/BVH/BVH.h:17-65, BVH.cpp:13-210, re-designed so that it
// parallelises: a subtree over k triangles always has exactly 2k-1 nodes, so every node's final index
// is known before its subtree is built and subtrees are built as independent tasks into a
// pre-sized array (see nexus_amd/csrc/host/BVH.cpp).
#pragma once

#include <cstdint>
#include <vector>

#include "Math.h"
#include "Triangle.h"

namespace nexus {

struct BVH2Node {
    float3 aabbMin, aabbMax;
    union {
        uint32_t leftNode;
        uint32_t firstTriIdx;
    };
    uint32_t triCount;  // 0 for inner nodes
    bool IsLeaf() const { return triCount > 0; }
};
static_assert(sizeof(BVH2Node) == 32, "BVH2Node is 32 bytes as in the reference");

class BVH2 {
public:
    BVH2() = default;
    explicit BVH2(const std::vector<Triangle>& tri);

    // threads = 0: use std::thread::hardware_concurrency().  The result does not depend on the thread count.
    void Build(unsigned threads = 0);

    std::vector<Triangle> triangles;
    std::vector<uint32_t> triangleIdx;
    std::vector<BVH2Node> nodes;
    std::vector<AABB> trianglesAABB;
};

}  // namespace nexus
