// TLAS.cpp — top-level BVH over instances: agglomerative clustering + BVH8 conversion.
// Algorithm of /root/reference/Nexus/src/Geometry/BVH/TLAS.cpp:13-91 (Bikker's "best match" clustering;
// node 0 is reserved and finally overwritten by the root; leaves have left == 0) and
// TLASBuilder.cpp:5-26 (same SAH-DP collapse as the BLAS, leaf payload = instance id).
#include "nexus/TLAS.h"

#include <algorithm>

#include "Collapse.h"

namespace nexus {

namespace {

// The clustering below is a nearest-neighbour chain: it ends because "the union with the smallest area" is a symmetric, finite
// measure.  A box with NaNs or infinities (an instance of a damaged mesh, a degenerate transform) breaks both — the reference
// indexes its table with -1 then — so such a box is made the conservative finite one first: every plane that is not a number
// moves out to +-1e9, every other beyond that is clamped to it.  Boxes of ordinary scenes are untouched.
AABB Finite(AABB b)
{
    constexpr float kFar = 1.0e9f;
    float* lo[3] = {&b.bMin.x, &b.bMin.y, &b.bMin.z};
    float* hi[3] = {&b.bMax.x, &b.bMax.y, &b.bMax.z};
    for (int a = 0; a < 3; a++) {
        if (!(*lo[a] >= -kFar)) *lo[a] = -kFar;  // NaN or below
        if (!(*lo[a] <= kFar)) *lo[a] = kFar;
        if (!(*hi[a] <= kFar)) *hi[a] = kFar;    // NaN or above
        if (!(*hi[a] >= -kFar)) *hi[a] = -kFar;
    }
    return b;
}

}  // namespace

void TLAS::Build()
{
    std::vector<AABB> bounds;
    bounds.reserve(bvhInstances.size());
    for (const BVHInstance& inst : bvhInstances) bounds.push_back(Finite(inst.GetBounds()));
    Cluster(bounds);
}

void TLAS::BuildFromBounds(const nx_bvh_instance* instances, uint32_t count)
{
    std::vector<AABB> bounds;
    bounds.reserve(count);
    for (uint32_t i = 0; i < count; i++) bounds.push_back(Finite(AABB(make_float3(instances[i].boundsMin), make_float3(instances[i].boundsMax))));
    Cluster(bounds);
}

namespace {

// The clusters still waiting for a partner, as a structure of arrays: slot k holds the box of BVH2 node `node[k]`.
// Keeping the six box planes in six contiguous float arrays lets the partner search below run as straight-line
// vector code over all live slots (the reference walks an index array and dereferences one 40-byte node per candidate).
// Slot order is part of the result: ties in the search go to the lowest slot, a merged cluster takes the slot of the
// cluster that started the chain link and the last live slot moves into the hole — the same bookkeeping as the
// reference's instance-index array (Geometry/BVH/TLAS.cpp:13-91), so the emitted BVH2 is node for node the same.
struct LiveClusters {
    std::vector<float> lox, loy, loz, hix, hiy, hiz;
    std::vector<uint32_t> node;
    mutable std::vector<float> cost;  // scratch of the partner search
    int live = 0;

    void Append(const AABB& b, uint32_t nodeIdx)
    {
        lox.push_back(b.bMin.x); loy.push_back(b.bMin.y); loz.push_back(b.bMin.z);
        hix.push_back(b.bMax.x); hiy.push_back(b.bMax.y); hiz.push_back(b.bMax.z);
        node.push_back(nodeIdx);
        cost.push_back(0.0f);
        live++;
    }
    void Put(int slot, float3 mn, float3 mx, uint32_t nodeIdx)
    {
        lox[slot] = mn.x; loy[slot] = mn.y; loz[slot] = mn.z;
        hix[slot] = mx.x; hiy[slot] = mx.y; hiz[slot] = mx.z;
        node[slot] = nodeIdx;
    }
    void MoveLastInto(int slot)
    {
        const int last = --live;
        lox[slot] = lox[last]; loy[slot] = loy[last]; loz[slot] = loz[last];
        hix[slot] = hix[last]; hiy[slot] = hiy[last]; hiz[slot] = hiz[last];
        node[slot] = node[last];
    }
    // Slot whose union with `self` has the smallest half-area; the lowest such slot on ties; -1 if every union
    // reaches 1e30 (the reference's sentinel) or nothing else is live.
    int Partner(int self) const
    {
        const float ax = lox[self], ay = loy[self], az = loz[self], bx = hix[self], by = hiy[self], bz = hiz[self];
        float* c = cost.data();
        for (int k = 0; k < live; k++) {
            const float ex = std::max(bx, hix[k]) - std::min(ax, lox[k]);
            const float ey = std::max(by, hiy[k]) - std::min(ay, loy[k]);
            const float ez = std::max(bz, hiz[k]) - std::min(az, loz[k]);
            c[k] = ex * ey + ey * ez + ex * ez;
        }
        c[self] = 1e30f;
        int best = -1;
        float least = 1e30f;
        for (int k = 0; k < live; k++)
            if (c[k] < least) { least = c[k]; best = k; }
        return best;
    }
};

}  // namespace

void TLAS::Cluster(const std::vector<AABB>& bounds)
{
    const uint32_t count = static_cast<uint32_t>(bounds.size());
    nodes.assign(1, TLASNode());  // node 0 is reserved for the root
    instancesIdx.clear();
    if (count == 0) return;
    nodes.reserve(2 * static_cast<size_t>(count));
    LiveClusters set;
    for (uint32_t i = 0; i < count; i++) {
        TLASNode leaf;
        leaf.aabbMin = bounds[i].bMin;
        leaf.aabbMax = bounds[i].bMax;
        leaf.blasIdx = i;
        leaf.blasCount = 1;
        set.Append(bounds[i], static_cast<uint32_t>(nodes.size()));
        nodes.push_back(leaf);
    }
    // Nearest-neighbour chain of length two: `tail` wants `head`; when `head` wants `tail` back the two are joined.
    int tail = 0, head = set.Partner(tail);
    while (set.live > 1) {
        const int wanted = set.Partner(head);
        if (wanted != tail) {  // not mutual: follow the chain
            tail = head;
            head = wanted;
            continue;
        }
        const TLASNode& first = nodes[set.node[head]];
        const TLASNode& second = nodes[set.node[tail]];
        TLASNode joined;
        joined.left = set.node[head];
        joined.right = set.node[tail];
        joined.blasCount = first.blasCount + second.blasCount;
        joined.aabbMin = fminf(second.aabbMin, first.aabbMin);
        joined.aabbMax = fmaxf(second.aabbMax, first.aabbMax);
        set.Put(tail, joined.aabbMin, joined.aabbMax, static_cast<uint32_t>(nodes.size()));
        set.MoveLastInto(head);
        nodes.push_back(joined);
        head = set.Partner(tail);
    }
    nodes[0] = nodes[set.node[tail]];
    // the live-slot table in its final state is what the reference leaves in TLAS::instancesIdx
    instancesIdx.assign(set.node.begin(), set.node.end());
}

int TLAS::FindBestMatch(int N, int A) const
{
    // kept for API compatibility (Geometry/BVH/TLAS.h:35): partner search over the current instancesIdx table
    LiveClusters set;
    for (int k = 0; k < N; k++) set.Append(AABB(nodes[instancesIdx[k]].aabbMin, nodes[instancesIdx[k]].aabbMax), instancesIdx[k]);
    return set.Partner(A);
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
