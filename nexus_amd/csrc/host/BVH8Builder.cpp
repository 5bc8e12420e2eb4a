// BVH8Builder.cpp — BLAS builder of the kept API: BVH2::Build + SAH-DP cost table + collapse.
// Call sequence mirrors /root/reference/Nexus/src/Assets/AssetManager.cpp:23-37
// (BVH8Builder builder(tris); builder.Init(); bvh8 = builder.Build();) and
// /root/reference/Nexus/src/Geometry/BVH/BVH8Builder.cpp:7-26.
#include "nexus/BVH8Builder.h"

#include <numeric>

#include "Collapse.h"

namespace nexus {

namespace {

struct Bvh2Tree final : collapse::Tree {
    const BVH2& b;
    std::vector<int> primsBelow;
    explicit Bvh2Tree(const BVH2& bvh) : b(bvh)
    {
        nodeCount = static_cast<uint32_t>(b.nodes.size());
        primsBelow.resize(nodeCount);
        // children have larger indices than their parent (see BVH.cpp), so a descending sweep is bottom-up
        for (uint32_t i = nodeCount; i-- > 0;) {
            const BVH2Node& n = b.nodes[i];
            primsBelow[i] = n.IsLeaf() ? static_cast<int>(n.triCount) : primsBelow[n.leftNode] + primsBelow[n.leftNode + 1];
        }
    }
    bool isLeaf(uint32_t n) const override { return b.nodes[n].IsLeaf(); }
    uint32_t left(uint32_t n) const override { return b.nodes[n].leftNode; }
    uint32_t right(uint32_t n) const override { return b.nodes[n].leftNode + 1; }
    AABB box(uint32_t n) const override { return AABB(b.nodes[n].aabbMin, b.nodes[n].aabbMax); }
    int leafPrims(uint32_t n) const override { return static_cast<int>(b.nodes[n].triCount); }
    int subtreePrims(uint32_t n) const override { return primsBelow[n]; }
    uint32_t sweepOrder(uint32_t k) const override { return nodeCount - 1 - k; }
    int emitLeaf(uint32_t n, uint32_t* dst, uint32_t& cursor) const override
    {
        const BVH2Node& node = b.nodes[n];
        for (uint32_t i = 0; i < node.triCount; i++) dst[cursor++] = b.triangleIdx[node.firstTriIdx + i];
        return static_cast<int>(node.triCount);
    }
};

}  // namespace

BVH8::BVH8(const std::vector<Triangle>& tri) : triangles(tri), triangleIdx(tri.size())
{
    std::iota(triangleIdx.begin(), triangleIdx.end(), 0u);
}

BVH8Builder::BVH8Builder(const std::vector<Triangle>& triangles) : m_Bvh2(triangles) {}

void BVH8Builder::Init(unsigned threads)
{
    m_Bvh2.Build(threads);
    Bvh2Tree tree(m_Bvh2);
    std::vector<collapse::Eval> evals;
    collapse::ComputeCosts(tree, evals);
    m_EvalStorage.resize(evals.size() * sizeof(collapse::Eval));
    std::memcpy(m_EvalStorage.data(), evals.data(), m_EvalStorage.size());
}

BVH8 BVH8Builder::Build()
{
    BVH8 bvh8(m_Bvh2.triangles);
    Bvh2Tree tree(m_Bvh2);
    std::vector<collapse::Eval> evals(m_EvalStorage.size() / sizeof(collapse::Eval));
    std::memcpy(evals.data(), m_EvalStorage.data(), m_EvalStorage.size());
    collapse::Collapse(tree, evals, bvh8);
    return bvh8;
}

}  // namespace nexus
