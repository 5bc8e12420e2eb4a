// Collapse.cpp — binary BVH -> compressed BVH8 (Ylitie, Karras, Laine 2017) by SAH dynamic programming,
// shared by BVH8Builder (BLAS over triangles) and TLASBuilder (TLAS over instances).
//
// Algorithm and output bytes of /root/reference/Nexus/src/Geometry/BVH/BVH8Builder.cpp:28-393 and
// TLASBuilder.cpp:27-370 (the two reference files are the same code over different node types):
//   C(n,0) = min(Cleaf(n), Cdistribute(n,7) + area(n) * C_NODE),  Cleaf = area * prims * C_PRIM if prims <= 3
//   C(n,i) = min(Cdistribute(n,i), C(n,i-1)),  Cdistribute(n,j) = min_k C(left,k) + C(right,j-1-k)
//   greedy child -> octant-slot assignment, 8-bit quantisation, 80-byte node encode.
// Re-design: the cost table is one flat array filled bottom-up in a single sweep (children always
// have larger indices than their parent in the BVH2 this library builds; the TLAS tree is swept in
// creation order), instead of a memoised recursion over vector<vector<NodeEval>>.
// Deviation (documented in DESIGN.md): quantised bounds are clamped to [0,255]; the reference BLAS path
// lets ceil() == 256 wrap to 0.
#include "Collapse.h"

#include <cassert>
#include <cfloat>
#include <cmath>

namespace nexus {
namespace collapse {

namespace {

float CLeaf(const AABB& box, int primCount)
{
    if (primCount > P_MAX) return 1.0e30f;
    return box.Area() * static_cast<float>(primCount) * C_PRIM;
}

float CDistribute(const Eval* evals, uint32_t left, uint32_t right, int j, int8_t& leftCount, int8_t& rightCount)
{
    float best = 1.0e30f;
    for (int k = 0; k < j; k++) {
        const float c = evals[left * 7 + k].cost + evals[right * 7 + (j - 1 - k)].cost;
        if (c < best) {
            best = c;
            leftCount = static_cast<int8_t>(k);
            rightCount = static_cast<int8_t>(j - 1 - k);
        }
    }
    return best;
}

uint8_t Quantize(float v)
{
    if (!(v == v)) return 0;  // 0 * inf on a degenerate (planar) axis
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return static_cast<uint8_t>(v);
}

}  // namespace

void ComputeCosts(const Tree& t, std::vector<Eval>& evals)
{
    const uint32_t n = t.nodeCount;
    evals.assign(static_cast<size_t>(n) * 7, Eval{});
    for (uint32_t step = 0; step < n; step++) {
        const uint32_t node = t.sweepOrder(step);
        Eval* e = &evals[static_cast<size_t>(node) * 7];
        const AABB box = t.box(node);
        if (t.isLeaf(node)) {
            const float c = CLeaf(box, t.leafPrims(node));
            for (int i = 0; i < 7; i++) e[i] = Eval{c, DEC_LEAF, 0, 0};
            continue;
        }
        const uint32_t l = t.left(node), r = t.right(node);
        {
            int8_t lc = 0, rc = 0;
            const float cLeaf = CLeaf(box, t.subtreePrims(node));
            const float cInternal = CDistribute(evals.data(), l, r, 7, lc, rc) + box.Area() * C_NODE;
            // (a leaf by the primitive count as well: with infinite boxes — a damaged mesh — cInternal is +inf, and CLeaf's
            //  1e30 for "too many primitives" would win the comparison for any subtree; the reference asserts there)
            if (t.subtreePrims(node) <= P_MAX && cLeaf < cInternal) e[0] = Eval{cLeaf, DEC_LEAF, 0, 0};
            else e[0] = Eval{cInternal, DEC_INTERNAL, lc, rc};
        }
        for (int i = 1; i < 7; i++) {
            int8_t lc = 0, rc = 0;
            const float cDist = CDistribute(evals.data(), l, r, i, lc, rc);
            if (cDist < e[i - 1].cost) e[i] = Eval{cDist, DEC_DISTRIBUTE, lc, rc};
            else e[i] = e[i - 1];
        }
    }
}

namespace {

struct Collapser {
    const Tree& t;
    const std::vector<Eval>& evals;
    BVH8& out;
    uint32_t usedNodes = 1;
    uint32_t usedIndices = 0;

    const Eval& eval(uint32_t n, int i) const { return evals[static_cast<size_t>(n) * 7 + i]; }

    // The (up to 8) BVH2 nodes that become children of the BVH8 node rooted at `n`, in the order the
    // reference's recursive GetChildrenIndices emits them (left before right, depth first).
    int GatherChildren(uint32_t n, int i, int* children, int count) const
    {
        const Eval& e = eval(n, i);
        if (e.decision == DEC_LEAF) {
            children[count++] = static_cast<int>(n);
            return count;
        }
        const uint32_t l = t.left(n), r = t.right(n);
        if (eval(l, e.leftCount).decision == DEC_DISTRIBUTE) count = GatherChildren(l, e.leftCount, children, count);
        else children[count++] = static_cast<int>(l);
        if (eval(r, e.rightCount).decision == DEC_DISTRIBUTE) count = GatherChildren(r, e.rightCount, children, count);
        else children[count++] = static_cast<int>(r);
        return count;
    }

    // Greedy assignment of children to octant slots: repeatedly take the globally cheapest
    // (child, free slot) pair, cost = dot(childCentroid - parentCentroid, (+-1,+-1,+-1)); first found wins ties.
    void OrderChildren(uint32_t parent, int* children, int childCount) const
    {
        const AABB pb = t.box(parent);
        const float3 pc = (pb.bMax + pb.bMin) * 0.5f;
        float cost[8][8];
        for (int c = 0; c < childCount; c++) {
            const AABB cb = t.box(static_cast<uint32_t>(children[c]));
            const float3 d = (cb.bMin + cb.bMax) * 0.5f - pc;
            for (int s = 0; s < 8; s++) {
                const float3 ds = make_float3((s & 4) ? -1.0f : 1.0f, (s & 2) ? -1.0f : 1.0f, (s & 1) ? -1.0f : 1.0f);
                cost[c][s] = dot(d, ds);
            }
        }
        bool slotTaken[8] = {};
        int slotOf[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        for (int round = 0; round < childCount; round++) {
            float minCost = FLT_MAX;
            int bestChild = -1, bestSlot = -1;
            for (int c = 0; c < childCount; c++) {
                if (slotOf[c] != -1) continue;
                for (int s = 0; s < 8; s++) {
                    if (slotTaken[s]) continue;
                    if (cost[c][s] < minCost) { minCost = cost[c][s]; bestChild = c; bestSlot = s; }
                }
            }
            if (bestChild == -1) {
                // no finite cost (boxes with NaNs or infinities): any free pair keeps the node well formed — the reference breaks
                // out here and loses the children
                for (int c = 0; c < childCount && bestChild == -1; c++)
                    if (slotOf[c] == -1) bestChild = c;
                for (int s = 0; s < 8 && bestSlot == -1; s++)
                    if (!slotTaken[s]) bestSlot = s;
                if (bestChild == -1 || bestSlot == -1) break;
            }
            slotOf[bestChild] = bestSlot;
            slotTaken[bestSlot] = true;
        }
        int ordered[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        for (int c = 0; c < childCount; c++)
            if (slotOf[c] != -1) ordered[slotOf[c]] = children[c];
        for (int s = 0; s < 8; s++) children[s] = ordered[s];
    }

    int EmitLeafPrims(uint32_t n)
    {
        if (t.isLeaf(n)) return t.emitLeaf(n, out.triangleIdx.data(), usedIndices);
        return EmitLeafPrims(t.left(n)) + EmitLeafPrims(t.right(n));
    }

    void CollapseNode(uint32_t n2, uint32_t n8)
    {
        const AABB nb = t.box(n2);
        const float denom = 1.0f / static_cast<float>((1 << N_Q) - 1);
        const float ex = std::ceil(std::log2((nb.bMax.x - nb.bMin.x) * denom));
        const float ey = std::ceil(std::log2((nb.bMax.y - nb.bMin.y) * denom));
        const float ez = std::ceil(std::log2((nb.bMax.z - nb.bMin.z) * denom));
        const float pw[3] = {std::exp2(ex), std::exp2(ey), std::exp2(ez)};

        BVH8Node node;
        std::memset(&node, 0, sizeof node);
        for (int a = 0; a < 3; a++) {
            uint32_t bits;
            std::memcpy(&bits, &pw[a], 4);
            node.e[a] = static_cast<uint8_t>(bits >> 23);
        }
        node.childBaseIdx = usedNodes;
        node.triangleBaseIdx = usedIndices;
        store(node.p, nb.bMin);

        int children[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        const int childCount = GatherChildren(n2, 0, children, 0);
        OrderChildren(n2, children, childCount);

        const float3 invScale = make_float3(1.0f / std::pow(2.0f, ex), 1.0f / std::pow(2.0f, ey), 1.0f / std::pow(2.0f, ez));
        int primsInNode = 0;
        for (int s = 0; s < 8; s++) {
            if (children[s] == -1) continue;
            const uint32_t c = static_cast<uint32_t>(children[s]);
            const AABB cb = t.box(c);
            node.qlox[s] = Quantize(std::floor((cb.bMin.x - node.p[0]) * invScale.x));
            node.qloy[s] = Quantize(std::floor((cb.bMin.y - node.p[1]) * invScale.y));
            node.qloz[s] = Quantize(std::floor((cb.bMin.z - node.p[2]) * invScale.z));
            node.qhix[s] = Quantize(std::ceil((cb.bMax.x - node.p[0]) * invScale.x));
            node.qhiy[s] = Quantize(std::ceil((cb.bMax.y - node.p[1]) * invScale.y));
            node.qhiz[s] = Quantize(std::ceil((cb.bMax.z - node.p[2]) * invScale.z));

            const int8_t decision = eval(c, 0).decision;
            if (decision == DEC_INTERNAL) {
                usedNodes++;
                node.meta[s] = static_cast<uint8_t>(0x20 | (24 + s));
                node.imask |= static_cast<uint8_t>(1u << s);
            } else {
                assert(decision == DEC_LEAF);
                const int prims = EmitLeafPrims(c);
                assert(prims <= P_MAX);
                uint8_t unary = 0;
                for (int j = 0; j < prims; j++) unary |= static_cast<uint8_t>(1u << (j + 5));
                node.meta[s] = static_cast<uint8_t>(unary | primsInNode);
                primsInNode += prims;
            }
        }
        assert(primsInNode <= 24);

        const uint32_t childBase = node.childBaseIdx;
        if (out.nodes.size() < usedNodes) out.nodes.resize(usedNodes);
        out.nodes[n8] = node;

        uint32_t k = 0;
        for (int s = 0; s < 8; s++) {
            if (children[s] == -1) continue;
            if (eval(static_cast<uint32_t>(children[s]), 0).decision == DEC_INTERNAL) CollapseNode(static_cast<uint32_t>(children[s]), childBase + k++);
        }
    }
};

}  // namespace

AABB Refit(std::vector<BVH8Node>& nodes, const uint32_t* primIdx, const AABB* primBounds)
{
    const size_t n = nodes.size();
    std::vector<AABB> nodeBox(n);
    // children are emitted after their parent (childBaseIdx > own index), so a descending sweep sees children first
    for (size_t k = n; k-- > 0;) {
        BVH8Node& node = nodes[k];
        AABB childBox[8];
        bool used[8] = {false, false, false, false, false, false, false, false};
        AABB nb;
        for (int s = 0; s < 8; s++) {
            AABB cb;
            if (node.imask & (1u << s)) {
                const uint32_t rel = static_cast<uint32_t>(__builtin_popcount(node.imask & ((1u << s) - 1u)));
                cb = nodeBox[node.childBaseIdx + rel];
            } else if (node.meta[s]) {
                const uint32_t first = node.triangleBaseIdx + (node.meta[s] & 0x1fu);
                const int count = __builtin_popcount(node.meta[s] >> 5);
                for (int j = 0; j < count; j++) cb.Grow(primBounds[primIdx[first + static_cast<uint32_t>(j)]]);
            } else {
                continue;
            }
            used[s] = true;
            childBox[s] = cb;
            nb.Grow(cb);
        }
        nodeBox[k] = nb;
        const float denom = 1.0f / static_cast<float>((1 << N_Q) - 1);
        const float ex = std::ceil(std::log2((nb.bMax.x - nb.bMin.x) * denom));
        const float ey = std::ceil(std::log2((nb.bMax.y - nb.bMin.y) * denom));
        const float ez = std::ceil(std::log2((nb.bMax.z - nb.bMin.z) * denom));
        const float pw[3] = {std::exp2(ex), std::exp2(ey), std::exp2(ez)};
        for (int a = 0; a < 3; a++) {
            uint32_t bits;
            std::memcpy(&bits, &pw[a], 4);
            node.e[a] = static_cast<uint8_t>(bits >> 23);
        }
        store(node.p, nb.bMin);
        const float3 invScale = make_float3(1.0f / std::pow(2.0f, ex), 1.0f / std::pow(2.0f, ey), 1.0f / std::pow(2.0f, ez));
        for (int s = 0; s < 8; s++) {
            if (!used[s]) continue;
            const AABB& cb = childBox[s];
            node.qlox[s] = Quantize(std::floor((cb.bMin.x - node.p[0]) * invScale.x));
            node.qloy[s] = Quantize(std::floor((cb.bMin.y - node.p[1]) * invScale.y));
            node.qloz[s] = Quantize(std::floor((cb.bMin.z - node.p[2]) * invScale.z));
            node.qhix[s] = Quantize(std::ceil((cb.bMax.x - node.p[0]) * invScale.x));
            node.qhiy[s] = Quantize(std::ceil((cb.bMax.y - node.p[1]) * invScale.y));
            node.qhiz[s] = Quantize(std::ceil((cb.bMax.z - node.p[2]) * invScale.z));
        }
    }
    return n ? nodeBox[0] : AABB();
}

void Collapse(const Tree& t, const std::vector<Eval>& evals, BVH8& out)
{
    out.nodes.clear();
    out.nodes.reserve(t.nodeCount / 8 + 16);
    out.nodes.resize(1);
    Collapser c{t, evals, out};
    c.CollapseNode(0, 0);
    out.nodes.resize(c.usedNodes);
}

}  // namespace collapse
}  // namespace nexus
