// TLAS.cpp — top-level BVH over instances: agglomerative clustering + BVH8 conversion.
// Algorithm of /root/reference/Nexus/src/Geometry/BVH/TLAS.cpp:13-91 (Bikker's "best match" clustering;
// node 0 is reserved and finally overwritten by the root; leaves have left == 0) and
// TLASBuilder.cpp:5-26 (same SAH-DP collapse as the BLAS, leaf payload = instance id).
#include "nexus/TLAS.h"

#include "Collapse.h"

namespace nexus {

void TLAS::Build()
{
    std::vector<AABB> bounds;
    bounds.reserve(bvhInstances.size());
    for (const BVHInstance& inst : bvhInstances) bounds.push_back(inst.GetBounds());
    Cluster(bounds);
}

void TLAS::BuildFromBounds(const nx_bvh_instance* instances, uint32_t count)
{
    std::vector<AABB> bounds;
    bounds.reserve(count);
    for (uint32_t i = 0; i < count; i++) bounds.emplace_back(make_float3(instances[i].boundsMin), make_float3(instances[i].boundsMax));
    Cluster(bounds);
}

void TLAS::Cluster(const std::vector<AABB>& bounds)
{
    nodes.clear();
    instancesIdx.clear();
    nodes.emplace_back();
    for (uint32_t i = 0; i < bounds.size(); i++) {
        instancesIdx.push_back(i + 1);
        TLASNode node;
        node.aabbMin = bounds[i].bMin;
        node.aabbMax = bounds[i].bMax;
        node.blasIdx = i;
        node.blasCount = 1;
        nodes.push_back(node);
    }
    int nodeIndices = static_cast<int>(bounds.size());
    if (nodeIndices == 0) return;
    int A = 0, B = FindBestMatch(nodeIndices, A);
    while (nodeIndices > 1) {
        const int C = FindBestMatch(nodeIndices, B);
        if (A == C) {
            const uint32_t nodeIdxA = instancesIdx[A], nodeIdxB = instancesIdx[B];
            TLASNode newNode;
            newNode.left = nodeIdxB;
            newNode.right = nodeIdxA;
            newNode.blasCount = nodes[nodeIdxA].blasCount + nodes[nodeIdxB].blasCount;
            newNode.aabbMin = fminf(nodes[nodeIdxA].aabbMin, nodes[nodeIdxB].aabbMin);
            newNode.aabbMax = fmaxf(nodes[nodeIdxA].aabbMax, nodes[nodeIdxB].aabbMax);
            instancesIdx[A] = static_cast<uint32_t>(nodes.size());
            instancesIdx[B] = instancesIdx[nodeIndices - 1];
            nodes.push_back(newNode);
            B = FindBestMatch(--nodeIndices, A);
        } else {
            A = B;
            B = C;
        }
    }
    nodes[0] = nodes[instancesIdx[A]];
}

int TLAS::FindBestMatch(int N, int A) const
{
    float smallest = 1e30f;
    int bestB = -1;
    for (int B = 0; B < N; B++) {
        if (B == A) continue;
        const float3 bMax = fmaxf(nodes[instancesIdx[A]].aabbMax, nodes[instancesIdx[B]].aabbMax);
        const float3 bMin = fminf(nodes[instancesIdx[A]].aabbMin, nodes[instancesIdx[B]].aabbMin);
        const float3 e = bMax - bMin;
        const float surfaceArea = e.x * e.y + e.y * e.z + e.x * e.z;
        if (surfaceArea < smallest) {
            smallest = surfaceArea;
            bestB = B;
        }
    }
    return bestB;
}

bool TLAS::Refit()
{
    if (bvh8.nodes.empty() || bvh8.triangleIdx.size() != bvhInstances.size()) return false;
    std::vector<AABB> bounds;
    bounds.reserve(bvhInstances.size());
    for (const BVHInstance& inst : bvhInstances) bounds.push_back(inst.GetBounds());
    collapse::Refit(bvh8.nodes, bvh8.triangleIdx.data(), bounds.data());
    return true;
}

void TLAS::Convert()
{
    TLASBuilder builder(*this);
    builder.Init();
    bvh8 = builder.Build();
}

namespace {

struct TlasTree final : collapse::Tree {
    const TLAS& t;
    explicit TlasTree(const TLAS& tlas) : t(tlas) { nodeCount = static_cast<uint32_t>(t.nodes.size()); }
    bool isLeaf(uint32_t n) const override { return t.nodes[n].IsLeaf(); }
    uint32_t left(uint32_t n) const override { return t.nodes[n].left; }
    uint32_t right(uint32_t n) const override { return t.nodes[n].right; }
    AABB box(uint32_t n) const override { return AABB(t.nodes[n].aabbMin, t.nodes[n].aabbMax); }
    int leafPrims(uint32_t n) const override { return static_cast<int>(t.nodes[n].blasCount); }
    int subtreePrims(uint32_t n) const override { return static_cast<int>(t.nodes[n].blasCount); }
    // clusters are appended after their members, so ascending order is bottom-up; node 0 is the root copy
    uint32_t sweepOrder(uint32_t k) const override { return k + 1 < nodeCount ? k + 1 : 0; }
    int emitLeaf(uint32_t n, uint32_t* dst, uint32_t& cursor) const override
    {
        dst[cursor++] = t.nodes[n].blasIdx;
        return 1;
    }
};

}  // namespace

void TLASBuilder::Init()
{
    TlasTree tree(m_Tlas);
    std::vector<collapse::Eval> evals;
    collapse::ComputeCosts(tree, evals);
    m_EvalStorage.resize(evals.size() * sizeof(collapse::Eval));
    std::memcpy(m_EvalStorage.data(), evals.data(), m_EvalStorage.size());
}

BVH8 TLASBuilder::Build()
{
    BVH8 bvh8;
    bvh8.triangleIdx = m_Tlas.instancesIdx;
    bvh8.triangleIdx.resize(m_Tlas.instancesIdx.size());
    TlasTree tree(m_Tlas);
    std::vector<collapse::Eval> evals(m_EvalStorage.size() / sizeof(collapse::Eval));
    std::memcpy(evals.data(), m_EvalStorage.data(), m_EvalStorage.size());
    collapse::Collapse(tree, evals, bvh8);
    return bvh8;
}

}  // namespace nexus
