// Collapse.h — internal interface of the shared binary-BVH -> BVH8 collapse (see Collapse.cpp).
#pragma once

#include <cstdint>
#include <vector>

#include "nexus/BVH8.h"
#include "nexus/Math.h"

namespace nexus {
namespace collapse {

enum : int8_t { DEC_UNDEFINED = -1, DEC_LEAF = 0, DEC_INTERNAL = 1, DEC_DISTRIBUTE = 2 };

struct Eval {
    float cost = 0.0f;
    int8_t decision = DEC_UNDEFINED;
    int8_t leftCount = 0, rightCount = 0;
};

// Read-only view of a binary tree (BVH2 of a mesh, or the agglomerative TLAS tree).
struct Tree {
    uint32_t nodeCount = 0;
    virtual ~Tree() = default;
    virtual bool isLeaf(uint32_t n) const = 0;
    virtual uint32_t left(uint32_t n) const = 0;
    virtual uint32_t right(uint32_t n) const = 0;
    virtual AABB box(uint32_t n) const = 0;
    virtual int leafPrims(uint32_t n) const = 0;     // primitives stored in leaf n
    virtual int subtreePrims(uint32_t n) const = 0;  // primitives below inner node n (saturating is fine above P_MAX)
    // k-th node of a bottom-up sweep: every node is visited after both of its children.
    virtual uint32_t sweepOrder(uint32_t k) const = 0;
    // Append leaf n's primitive ids at dst[cursor...], advance cursor, return how many.
    virtual int emitLeaf(uint32_t n, uint32_t* dst, uint32_t& cursor) const = 0;
};

void ComputeCosts(const Tree& t, std::vector<Eval>& evals);
// out.triangleIdx must already be sized to the primitive count.
void Collapse(const Tree& t, const std::vector<Eval>& evals, BVH8& out);

// Refit: same topology, new leaf bounds.  primBounds[id] is the (world) box of primitive id (ids as stored in primIdx).
// Every node's frame (p, e) and the quantised boxes of its children are recomputed bottom-up with the formulas of
// Collapse(); imask, meta, child and primitive indices are untouched.  Returns the new root box.
AABB Refit(std::vector<BVH8Node>& nodes, const uint32_t* primIdx, const AABB* primBounds);

}  // namespace collapse
}  // namespace nexus
