// nx_lbvh.hip — BLAS construction on the device: triangles in HBM -> compressed wide BVH (the same 80-byte nodes and
// primitive-index list the host builders and the reference produce, /root/reference/Nexus/src/Geometry/BVH/BVH8.h:24-49).
//
// SURVEY.md section 8 row f1.  The reference builds on one CPU thread (binned-SAH BVH2, BVH.cpp:65-210, then the SAH-DP
// collapse of BVH8Builder.cpp:63-117, 273-393: ~10 s per million triangles); the host builder of this repo is the same
// algorithm task-parallel (0.4 s per million).  This is the GPU path for scenes where even that dominates time to first
// frame (10 M triangles replicated on 8 ranks): a linear BVH — 63-bit Morton codes of the centroids, one radix sort, the
// binary radix tree of Karras 2012 with every internal node found independently (or a clustering build, 4b), bounds fitted
// bottom-up with one arrival counter per node — collapsed top-down, level by level, into 8-wide nodes by the reference's SAH
// dynamic programme (BVH8Builder.cpp:63-117): its cost table is filled bottom-up like the bounds (4c), and each wide node takes
// the up to eight subtrees the table chose, as wide nodes or as leaf slots of at most three triangles.  Children are assigned to octant slots with the reference's greedy rule
// (BVH8Builder.cpp:170-252) and quantised exactly as the host builder does (8-bit grid, floor / ceil, power-of-two scale).
// The tree is a valid, conservative CWBVH for the unchanged traversal kernels; its node bytes differ from the SAH builder's
// (a different tree): hits are identical, traversal visits more nodes (quality is the price of the build speed).
#include <string.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "nx_context.h"
#include "nx_instbox.h"
#include "nx_math.h"

namespace nxd {

namespace {

constexpr int kBlock = 256;

struct Box3 {
    float lo[3], hi[3];
};

__device__ __forceinline__ uint32_t float_ordered(float f)  // monotone float -> uint map for atomicMin / atomicMax
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_float(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// ---- 1. per-triangle boxes, centroid bounds ---------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) tri_bounds_kernel(const nx_triangle* __restrict__ tris, const uint32_t n, Box3* __restrict__ triBox, uint32_t* __restrict__ sceneBounds)
{
    float cl[3] = {1e30f, 1e30f, 1e30f}, ch[3] = {-1e30f, -1e30f, -1e30f};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const nx_triangle t = tris[i];
        Box3 b;
        for (int a = 0; a < 3; a++) {
            b.lo[a] = fminf(fminf(t.pos0[a], t.pos1[a]), t.pos2[a]);
            b.hi[a] = fmaxf(fmaxf(t.pos0[a], t.pos1[a]), t.pos2[a]);
            const float c = 0.5f * (b.lo[a] + b.hi[a]);
            cl[a] = fminf(cl[a], c);
            ch[a] = fmaxf(ch[a], c);
        }
        triBox[i] = b;
    }
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) {
            cl[a] = fminf(cl[a], __shfl_down(cl[a], o));
            ch[a] = fmaxf(ch[a], __shfl_down(ch[a], o));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&sceneBounds[a], float_ordered(cl[a]));
            atomicMax(&sceneBounds[3 + a], float_ordered(ch[a]));
        }
    }
}

// the same for a TLAS: the primitives are the instances' world-space boxes.  BVHInstance::SetTransform computed them from the BLAS
// root's quantisation frame [p, p + 255 * 2^e], as the reference does — up to twice the mesh's extent per axis (the scale is a
// power of two), before the transform's own inflation.  With `blas` the box is taken from what the frame holds instead: the
// boxes of the root's children, each through the instance transform, intersected with the record's box.  Both contain the
// geometry, so does their intersection; rays stop entering instances they pass at a distance.
__global__ void __launch_bounds__(kBlock) instance_bounds_kernel(const nx_bvh_instance* __restrict__ inst, const uint32_t n, const BlasDev* __restrict__ blas,
                                                                 Box3* __restrict__ box, uint32_t* __restrict__ sceneBounds)
{
    float cl[3] = {1e30f, 1e30f, 1e30f}, ch[3] = {-1e30f, -1e30f, -1e30f};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Box3 b;
        for (int a = 0; a < 3; a++) {
            // (as the host builder's Finite(): a bound that is not a number moves out to +-1e9, so that one degenerate instance
            //  does not make the root frame infinite and every other instance unreachable)
            constexpr float kFar = 1.0e9f;
            float lo = inst[i].boundsMin[a], hi = inst[i].boundsMax[a];
            if (!(lo >= -kFar)) lo = -kFar;
            if (!(lo <= kFar)) lo = kFar;
            if (!(hi <= kFar)) hi = kFar;
            if (!(hi >= -kFar)) hi = -kFar;
            b.lo[a] = lo;
            b.hi[a] = hi;
        }
        if (blas && kNodeStride == 5) {
            InstBox t;
            for (int a = 0; a < 3; a++) { t.lo[a] = b.lo[a]; t.hi[a] = b.hi[a]; }
            // (singular: the record carries Mat4::Inverted's identity fallback although its transform is not the identity)
            const bool singular = mat4_is_identity(inst[i].invTransform.cell) && !mat4_is_identity(inst[i].transform.cell);
            tighten_instance_box(reinterpret_cast<const nx_bvh8_node*>(blas[inst[i].bvhIdx].nodes), inst[i].transform.cell, t, singular);
            for (int a = 0; a < 3; a++) { b.lo[a] = t.lo[a]; b.hi[a] = t.hi[a]; }
        }
        for (int a = 0; a < 3; a++) {
            const float c = 0.5f * (b.lo[a] + b.hi[a]);
            cl[a] = fminf(cl[a], c);
            ch[a] = fmaxf(ch[a], c);
        }
        box[i] = b;
    }
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) {
            cl[a] = fminf(cl[a], __shfl_down(cl[a], o));
            ch[a] = fmaxf(ch[a], __shfl_down(ch[a], o));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&sceneBounds[a], float_ordered(cl[a]));
            atomicMax(&sceneBounds[3 + a], float_ordered(ch[a]));
        }
    }
}

// ---- 2. 63-bit Morton codes of the box centres --------------------------------------------------------------------
__device__ __forceinline__ unsigned long long spread21(uint32_t v)  // 21 bits -> every third bit
{
    unsigned long long x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ void __launch_bounds__(kBlock) morton_kernel(const Box3* __restrict__ triBox, const uint32_t n, const uint32_t* __restrict__ sceneBounds,
                                                        unsigned long long* __restrict__ codes, uint32_t* __restrict__ order)
{
    float lo[3], inv[3];
    for (int a = 0; a < 3; a++) {
        lo[a] = ordered_float(sceneBounds[a]);
        const float ext = ordered_float(sceneBounds[3 + a]) - lo[a];
        inv[a] = ext > 0.0f ? 2097151.0f / ext : 0.0f;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Box3 b = triBox[i];
        uint32_t q[3];
        for (int a = 0; a < 3; a++) q[a] = (uint32_t)fminf(fmaxf((0.5f * (b.lo[a] + b.hi[a]) - lo[a]) * inv[a], 0.0f), 2097151.0f);
        codes[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
        order[i] = i;
    }
}

// ---- 3. binary radix tree (Karras 2012): internal nodes 0 .. n-2, leaf k = node n-1+k (k = position in the sorted order)
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ codes, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = codes[i], b = codes[j];
    if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);  // equal codes: the index breaks the tie
    return __clzll((long long)(a ^ b));
}

struct Bvh2 {
    int* left;     // [n-1] child node ids (>= n-1: leaf)
    int* right;
    int* parent;   // [2n-1]
    int* first;    // [n-1] range of sorted positions covered by the internal node (radix tree only)
    int* last;
    int* count;    // [n-1] primitives under the internal node
    Box3* box;     // [2n-1]
    int* arrived;  // [n-1]
    struct Eval* eval;  // [7 (n-1)] cost table of the collapse (internal nodes; a leaf's entries follow from its box), or nullptr
    float primCost;     // the collapse's cost of testing one primitive of a leaf slot, relative to one node test
};

// One entry of the collapse's cost table (Ylitie, Karras, Laine 2017; the reference's BVH8Builder.cpp:63-117 and the host's
// Collapse.cpp): eval[7 node + i] = cheapest way to present the subtree under `node` as at most i + 1 children of a wide node —
// as one leaf slot (at most three primitives), as one wide node of its own (entry 0 only), or by handing leftCount + 1 and
// rightCount + 1 roots to its two children.
enum : int8_t { kDecLeaf = 0, kDecInternal = 1, kDecDistribute = 2 };
struct Eval {
    float cost;
    int8_t decision, leftCount, rightCount, pad_;
};
static_assert(sizeof(Eval) == 8, "one 8-byte word per entry");
// The reference prices a triangle test at 0.3 of a node test (nexus::C_PRIM, C_NODE, include/nexus/BVH8.h; Geometry/BVH/BVH8.h:18-21):
// its kernel tests a leaf's triangles in an inner loop.  In THIS trace kernel a triangle record costs a whole loop iteration, exactly like
// a node (nx_trace.hip: one record per iteration), so the device builder's collapse prices it higher: 0.75, the winner of the sweep
// 0.3 / 0.5 / 0.75 / 1.0 / 1.5 on configs[1], [3], [4] (profiles/r06_prim_cost_sweep.txt: driver command +2.7 %, configs[3] +1.2 %,
// configs[4] -0.8 %; 1.0 and above lose 3.6 % on configs[4]).  The host builder (and the tests' CPU restatement) keep the reference's 0.3: their trees are
// compared byte for byte with the reference's algorithm; the parity tests of device-built trees read the tree back, so they hold.
constexpr float kCostPrimDevice = 0.75f, kCostNode = 1.0f;
static float blas_prim_cost()
{
    float cost = kCostPrimDevice;
    if (const char* on = std::getenv("NX_TUNING_KNOBS"); on && std::atoi(on) == 1)
        if (const char* e = std::getenv("NX_BLAS_PRIM_COST")) cost = (float)std::atof(e);  // sweeps only
    return cost;
}
// A TLAS "primitive" is an instance: entering one costs a loop iteration of its own (record fetch, transform, test of the BLAS
// root) before anything of the BLAS is traversed, so instances that share a leaf slot are all entered by every ray that
// touches the slot's box.  With the triangle's 0.3 the collapse packs up to three instances into a slot wherever that saves
// a node; priced as what they are, instances get slots of their own.
constexpr float kCostInstance = 4.0f;
constexpr int kLeafMax = 3;                            // nexus::P_MAX

__global__ void __launch_bounds__(kBlock) radix_tree_kernel(const unsigned long long* __restrict__ codes, const int n, Bvh2 t)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
        const int d = delta(codes, n, i, i + 1) - delta(codes, n, i, i - 1) >= 0 ? 1 : -1;
        const int dmin = delta(codes, n, i, i - d);
        int lmax = 2;
        while (delta(codes, n, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int s = lmax / 2; s >= 1; s /= 2)
            if (delta(codes, n, i, i + (l + s) * d) > dmin) l += s;
        const int j = i + l * d;
        const int dnode = delta(codes, n, i, j);
        int s = 0, step = l;
        do {  // binary search for the split: step = ceil(l / 2), ceil(l / 4), ... 1
            step = (step + 1) >> 1;
            if (delta(codes, n, i, i + (s + step) * d) > dnode) s += step;
        } while (step > 1);
        const int gamma = i + s * d + (d < 0 ? -1 : 0);
        const int lo = d > 0 ? i : j, hi = d > 0 ? j : i;
        const int leftNode = lo == gamma ? (n - 1) + gamma : gamma;
        const int rightNode = hi == gamma + 1 ? (n - 1) + gamma + 1 : gamma + 1;
        t.left[i] = leftNode;
        t.right[i] = rightNode;
        t.first[i] = lo;
        t.last[i] = hi;
        t.count[i] = hi - lo + 1;
        t.parent[leftNode] = i;
        t.parent[rightNode] = i;
        if (i == 0) t.parent[0] = -1;
    }
}

// A box written by another thread of this launch (the bottom-up passes): device-scope loads / stores of its three 8-byte words —
// a store is written through to where every XCD sees it, a load does not stop at this XCD's L2 — so that the passes need no
// __threadfence() (an L2 write-back + invalidate per node visit: see cost_kernel).  Box arrays are 8-byte aligned (24-byte
// elements from an allocation's start).
// What publishes a child's box here is gfx950 behaviour, not the HIP / LLVM memory model (which gives relaxed atomics no release /
// acquire edge): device-scope stores are written through (sc1), a wave's memory instructions issue in order, and the arrival
// counter is bumped only after `s_waitcnt vmcnt(0)` on the stores.  Another target must put the fences back.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "nx_lbvh.hip: load_fresh / store_shared publish without fences, which is only argued for gfx950 (see the comment above)"
#endif
__device__ __forceinline__ Box3 load_fresh(const Box3* p)
{
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(p);
    unsigned long long v[3];
    for (int k = 0; k < 3; k++) v[k] = __hip_atomic_load(w + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Box3 b;
    memcpy(&b, v, sizeof b);
    return b;
}
__device__ __forceinline__ void store_shared(Box3* p, const Box3& b)
{
    unsigned long long v[3];
    memcpy(v, &b, sizeof b);
    for (int k = 0; k < 3; k++) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- 4. bounds, bottom-up: the second child to arrive at a node computes it and carries on
__global__ void __launch_bounds__(kBlock) fit_kernel(const Box3* __restrict__ triBox, const uint32_t* __restrict__ order, const int n, Bvh2 t)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int leaf = (n - 1) + k;
        store_shared(&t.box[leaf], triBox[order[k]]);
        int node = n > 1 ? t.parent[leaf] : -1;
        while (node >= 0) {
            // this thread's child box is visible before its arrival is: written through (store_shared), and the store has completed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (atomicAdd(&t.arrived[node], 1) == 0) break;  // the sibling subtree is not done yet: its thread will do this node
            const Box3 a = load_fresh(&t.box[t.left[node]]), b = load_fresh(&t.box[t.right[node]]);
            Box3 m;
            for (int x = 0; x < 3; x++) { m.lo[x] = fminf(a.lo[x], b.lo[x]); m.hi[x] = fmaxf(a.hi[x], b.hi[x]); }
            store_shared(&t.box[node], m);
            node = t.parent[node];
        }
    }
}

// ---- 4b. the alternative to steps 3-4: parallel locally-ordered clustering (Meister & Bittner 2018).  The sorted primitives
// start as clusters; every round each cluster looks `radius` places to either side along the Morton order for the neighbour whose
// union with it has the smallest surface area, pairs that chose each other merge into a new BVH2 node, and the survivors are
// compacted (order preserved).  Bottom-up and surface-area driven, the tree comes close to a top-down SAH build where the
// radix tree only follows the bits of the space-filling curve — at the price of a few dozen rounds instead of one launch.
// Node ids as in the radix tree: leaf k = n - 1 + k; internal nodes are numbered downwards from n - 2 so that the last
// merge — the root — is node 0, which the collapse starts from.
__device__ __forceinline__ float half_area(const Box3& b)
{
    const float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
    return ex * ey + ey * ez + ex * ez;
}
__device__ __forceinline__ float union_half_area(const Box3& a, const Box3& b)
{
    const float ex = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]), ey = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]), ez = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    return ex * ey + ey * ez + ex * ez;
}

__global__ void __launch_bounds__(kBlock) ploc_leaves_kernel(const Box3* __restrict__ primBox, const uint32_t* __restrict__ order, const int n, Bvh2 t, int* __restrict__ cluster)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        t.box[(n - 1) + k] = primBox[order[k]];
        cluster[k] = (n - 1) + k;
    }
}

__global__ void __launch_bounds__(kBlock) ploc_neighbour_kernel(const Bvh2 t, const int* __restrict__ cluster, const int m, const int radius, int* __restrict__ nearest)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const Box3 mine = t.box[cluster[i]];
        float best = 0.0f;
        int bestLo = -1, bestHi = -1, pick = -1;
        const int from = max(0, i - radius), to = min(m - 1, i + radius);
        for (int j = from; j <= to; j++) {
            if (j == i) continue;
            float a = union_half_area(mine, t.box[cluster[j]]);
            if (!(a == a)) a = 3.0e38f;  // a box with NaNs pairs up last
            // ties are broken on the PAIR (smaller index, then larger), so that the two members of the best pair pick each other:
            // every round then merges at least one pair
            const int lo = min(i, j), hi = max(i, j);
            if (pick < 0 || a < best || (a == best && (lo < bestLo || (lo == bestLo && hi < bestHi)))) {
                best = a; bestLo = lo; bestHi = hi; pick = j;
            }
        }
        nearest[i] = pick;
    }
}

__global__ void __launch_bounds__(kBlock) ploc_merge_kernel(Bvh2 t, const int n, const int* __restrict__ cluster, const int* __restrict__ nearest, const int m, int* __restrict__ merged,
                                                            int* __restrict__ keep, int* __restrict__ mergeCounter)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int j = nearest[i];
        int node = cluster[i], kept = 1;
        if (j >= 0 && nearest[j] == i) {
            if (i < j) {
                const int a = cluster[i], b = cluster[j];
                const int id = (n - 2) - atomicAdd(mergeCounter, 1);
                const Box3 ba = t.box[a], bb = t.box[b];
                Box3 u;
                for (int x = 0; x < 3; x++) { u.lo[x] = fminf(ba.lo[x], bb.lo[x]); u.hi[x] = fmaxf(ba.hi[x], bb.hi[x]); }
                t.left[id] = a;
                t.right[id] = b;
                t.parent[a] = id;
                t.parent[b] = id;
                if (id == 0) t.parent[0] = -1;  // the last merge is the root
                t.box[id] = u;
                t.count[id] = (a >= n - 1 ? 1 : t.count[a]) + (b >= n - 1 ? 1 : t.count[b]);
                node = id;
            } else {
                kept = 0;  // absorbed by its partner
            }
        }
        merged[i] = node;
        keep[i] = kept;
    }
}

__global__ void __launch_bounds__(kBlock) ploc_compact_kernel(const int* __restrict__ merged, const int* __restrict__ keep, const int* __restrict__ offset, const int m, int* __restrict__ next)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
        if (keep[i]) next[offset[i]] = merged[i];
}

// ---- 4d. the third way to the binary tree: the host builder's rule on the device — top-down, binned surface-area heuristic
// (BVH.cpp; the reference's Geometry/BVH/BVH.cpp:65-210: bins over the centroid bounds of the node, the plane with the smallest
// count x area sum over the three axes, one primitive per leaf, "split in half" when nothing separates).  Level by level: every
// node of a level bins its primitives with atomics (a workgroup whose 256 primitives share a node bins in LDS first), one
// thread per node picks the plane and creates the children, a scan of the "goes left" flags gives every primitive its place
// in the next level's order, and the scatter pass collects the children's centroid bounds.  Nodes of at most kSahSmall primitives
// leave the levels and are finished by one thread each with an exact sweep.
#ifndef NX_SAH_BINS
#define NX_SAH_BINS 16
#endif
#ifndef NX_SAH_SMALL
#define NX_SAH_SMALL 8
#endif
constexpr int kSahBins = NX_SAH_BINS;
constexpr int kSahSmall = NX_SAH_SMALL;
struct SahBin {
    uint32_t count;
    uint32_t lo[3], hi[3];  // float_ordered
};
constexpr int kSahBinWords = 7;
struct SahSeg {
    int first, count, node;
    uint32_t cb[6];   // centroid bounds (float_ordered), collected by the scatter pass that created the segment
    int axis;         // chosen axis; -1: split in half by position
    int plane;        // bins <= plane go left
    int countL;
    int child[2];     // the children's places in the next level's list; -1: not there (a leaf, or finished by sah_small_kernel)
};
struct SahSmall {
    int first, count, node;
};
struct SahCounters {
    uint32_t nodes;       // internal node ids handed out (the root is 0)
    uint32_t nextCount;   // segments of the next level
    uint32_t smallCount;  // segments handed to sah_small_kernel
    uint32_t pad_;
};

__device__ __forceinline__ void box_centre(const Box3& b, float c[3])
{
    for (int a = 0; a < 3; a++) c[a] = 0.5f * (b.lo[a] + b.hi[a]);
}
// the bin of a centroid coordinate — the one expression the binning, the flags and the scatter all use
__device__ __forceinline__ int sah_bin(const float c, const float lo, const float hi)
{
    const float scale = (float)kSahBins / (hi - lo);
    const float f = (c - lo) * scale;
    int b = f >= 0.0f ? (int)fminf(f, (float)(kSahBins - 1)) : 0;  // (a NaN lands in bin 0)
    return min(max(b, 0), kSahBins - 1);
}
__device__ __forceinline__ bool sah_goes_left(const SahSeg& sg, const Box3& pb, const int i)
{
    if (sg.axis < 0) return i - sg.first < sg.countL;
    float c[3];
    box_centre(pb, c);
    return sah_bin(c[sg.axis], ordered_float(sg.cb[sg.axis]), ordered_float(sg.cb[3 + sg.axis])) <= sg.plane;
}

// union of all primitive boxes (the root's box) — 6 words, float_ordered
__global__ void __launch_bounds__(kBlock) box_union_kernel(const Box3* __restrict__ primBox, const uint32_t n, uint32_t* __restrict__ out)
{
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Box3 b = primBox[i];
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], b.lo[a]); hi[a] = fmaxf(hi[a], b.hi[a]); }
    }
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_down(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_down(hi[a], o));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&out[a], float_ordered(lo[a]));
            atomicMax(&out[3 + a], float_ordered(hi[a]));
        }
    }
}

__global__ void sah_root_kernel(const uint32_t* __restrict__ centroidBounds, const uint32_t* __restrict__ rootBox, const int n, Bvh2 t, SahSeg* __restrict__ segs, int* __restrict__ segOf)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        SahSeg sg{};
        sg.first = 0; sg.count = n; sg.node = 0;
        for (int k = 0; k < 6; k++) sg.cb[k] = centroidBounds[k];
        sg.child[0] = sg.child[1] = -1;
        segs[0] = sg;
        Box3 b;
        for (int a = 0; a < 3; a++) { b.lo[a] = ordered_float(rootBox[a]); b.hi[a] = ordered_float(rootBox[3 + a]); }
        t.box[0] = b;
        t.count[0] = n;
        t.parent[0] = -1;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) segOf[i] = 0;
}

__device__ __forceinline__ void sah_bin_add(uint32_t* bin /* 7 words */, const Box3& pb)
{
    atomicAdd(&bin[0], 1u);
    for (int x = 0; x < 3; x++) {
        atomicMin(&bin[1 + x], float_ordered(pb.lo[x]));
        atomicMax(&bin[4 + x], float_ordered(pb.hi[x]));
    }
}

__global__ void __launch_bounds__(kBlock) sah_bins_init_kernel(uint32_t* __restrict__ bins, const size_t words)
{
    for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < words; k += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(k % kSahBinWords);
        bins[k] = (w >= 1 && w <= 3) ? 0xffffffffu : 0u;
    }
}

// bins[seg][axis][bin]: count and box of the primitives whose centroid falls into the bin
__global__ void __launch_bounds__(kBlock) sah_bin_kernel(const Box3* __restrict__ primBox, const int* __restrict__ ids, const int* __restrict__ segOf, const int n,
                                                         const SahSeg* __restrict__ segs, uint32_t* __restrict__ bins)
{
    __shared__ uint32_t local[3 * kSahBins * kSahBinWords];
    for (int tile = blockIdx.x * blockDim.x; tile < n; tile += gridDim.x * blockDim.x) {
        const int i = tile + (int)threadIdx.x;
        const int s = i < n ? segOf[i] : -1;
        const int s0 = segOf[tile];
        // a workgroup whose primitives all belong to one segment (every workgroup of the upper levels) bins in LDS and adds
        // its 336 words to the segment's bins once: 16 x fewer atomics on the few words all primitives of a large node share
        const bool uniform = __syncthreads_and(i >= n || s == s0) != 0 && s0 >= 0;
        if (uniform) {
            for (int k = threadIdx.x; k < 3 * kSahBins * kSahBinWords; k += blockDim.x) {
                const int w = k % kSahBinWords;
                local[k] = w == 0 ? 0u : (w <= 3 ? 0xffffffffu : 0u);
            }
            __syncthreads();
        }
        if (s >= 0) {
            const SahSeg& sg = segs[s];
            const Box3 pb = primBox[ids[i]];
            float c[3];
            box_centre(pb, c);
            for (int a = 0; a < 3; a++) {
                const float lo = ordered_float(sg.cb[a]), hi = ordered_float(sg.cb[3 + a]);
                if (!(lo < hi)) continue;  // dead axis
                const int b = sah_bin(c[a], lo, hi);
                uint32_t* dst = uniform ? &local[(a * kSahBins + b) * kSahBinWords] : &bins[((size_t)s * 3 * kSahBins + a * kSahBins + b) * kSahBinWords];
                sah_bin_add(dst, pb);
            }
        }
        if (uniform) {
            __syncthreads();
            uint32_t* dst = &bins[(size_t)s0 * 3 * kSahBins * kSahBinWords];
            for (int k = threadIdx.x; k < 3 * kSahBins * kSahBinWords; k += blockDim.x) {
                const int w = k % kSahBinWords;
                const uint32_t v = local[k];
                if (w == 0) { if (v) atomicAdd(&dst[k], v); }
                else if (w <= 3) { if (v != 0xffffffffu) atomicMin(&dst[k], v); }
                else if (v) atomicMax(&dst[k], v);
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ void bin_box(const uint32_t* bin, Box3& b)
{
    for (int x = 0; x < 3; x++) { b.lo[x] = ordered_float(bin[1 + x]); b.hi[x] = ordered_float(bin[4 + x]); }
}
__device__ __forceinline__ void grow(Box3& a, const Box3& b)
{
    for (int x = 0; x < 3; x++) { a.lo[x] = fminf(a.lo[x], b.lo[x]); a.hi[x] = fmaxf(a.hi[x], b.hi[x]); }
}
__device__ __forceinline__ Box3 empty_box()
{
    Box3 b;
    for (int x = 0; x < 3; x++) { b.lo[x] = 1e30f; b.hi[x] = -1e30f; }
    return b;
}

// One thread per segment: the plane (BVH.cpp FindBestSplitPlane), the two children, their places in the next level.
__global__ void __launch_bounds__(kBlock) sah_split_kernel(SahSeg* __restrict__ segs, const uint32_t segCount, const uint32_t* __restrict__ bins, const int n, Bvh2 t,
                                                           SahSeg* __restrict__ nextSegs, SahSmall* __restrict__ small, SahCounters* __restrict__ ctr)
{
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < segCount; s += gridDim.x * blockDim.x) {
        SahSeg sg = segs[s];
        const uint32_t* sb = &bins[(size_t)s * 3 * kSahBins * kSahBinWords];
        float bestCost = 3.0e38f;
        int axis = -1, plane = 0, countL = 0;
        for (int a = 0; a < 3; a++) {
            const float lo = ordered_float(sg.cb[a]), hi = ordered_float(sg.cb[3 + a]);
            if (!(lo < hi)) continue;
            // counts x areas of "bins above k", downwards, then the sweep upwards
            float aboveCost[kSahBins];
            int aboveCount[kSahBins];
            Box3 run = empty_box();
            int cnt = 0;
            for (int k = kSahBins - 1; k >= 1; k--) {
                const uint32_t* bin = &sb[(a * kSahBins + k) * kSahBinWords];
                if (bin[0]) { Box3 b; bin_box(bin, b); grow(run, b); cnt += (int)bin[0]; }
                aboveCount[k - 1] = cnt;
                aboveCost[k - 1] = cnt ? (float)cnt * half_area(run) : 0.0f;
            }
            run = empty_box();
            cnt = 0;
            for (int k = 0; k < kSahBins - 1; k++) {
                const uint32_t* bin = &sb[(a * kSahBins + k) * kSahBinWords];
                if (bin[0]) { Box3 b; bin_box(bin, b); grow(run, b); cnt += (int)bin[0]; }
                if (cnt == 0 || aboveCount[k] == 0) continue;  // nothing on one side: not a split
                const float cost = (float)cnt * half_area(run) + aboveCost[k];
                if (cost < bestCost) { bestCost = cost; axis = a; plane = k; countL = cnt; }
            }
        }
        Box3 childBox[2];
        if (axis >= 0) {
            childBox[0] = empty_box();
            childBox[1] = empty_box();
            for (int k = 0; k < kSahBins; k++) {
                const uint32_t* bin = &sb[(axis * kSahBins + k) * kSahBinWords];
                if (bin[0]) { Box3 b; bin_box(bin, b); grow(childBox[k <= plane ? 0 : 1], b); }
            }
        } else {
            // no axis separates the centroids (or the boxes hold NaNs): halves by position under the node's own box — conservative
            countL = sg.count / 2;
            childBox[0] = childBox[1] = t.box[sg.node];
        }
        sg.axis = axis;
        sg.plane = plane;
        sg.countL = countL;
        int childId[2];
        for (int h = 0; h < 2; h++) {
            const int first = h == 0 ? sg.first : sg.first + countL, count = h == 0 ? countL : sg.count - countL;
            sg.child[h] = -1;
            if (count == 1) {
                childId[h] = (n - 1) + first;  // a leaf: its box is its primitive's, filled in when the order is final
            } else {
                const int id = (int)atomicAdd(&ctr->nodes, 1u);
                childId[h] = id;
                t.box[id] = childBox[h];
                t.count[id] = count;
                if (count <= kSahSmall) {
                    small[atomicAdd(&ctr->smallCount, 1u)] = SahSmall{first, count, id};
                } else {
                    const int at = (int)atomicAdd(&ctr->nextCount, 1u);
                    SahSeg c{};
                    c.first = first; c.count = count; c.node = id;
                    for (int x = 0; x < 3; x++) { c.cb[x] = 0xffffffffu; c.cb[3 + x] = 0u; }
                    if (axis < 0)
                        for (int x = 0; x < 6; x++) c.cb[x] = sg.cb[x];  // (the scatter pass collects nothing for a position split)
                    c.child[0] = c.child[1] = -1;
                    nextSegs[at] = c;
                    sg.child[h] = at;
                }
            }
            t.parent[childId[h]] = sg.node;
        }
        t.left[sg.node] = childId[0];
        t.right[sg.node] = childId[1];
        segs[s] = sg;
    }
}

__global__ void __launch_bounds__(kBlock) sah_flag_kernel(const Box3* __restrict__ primBox, const int* __restrict__ ids, const int* __restrict__ segOf, const int n,
                                                          const SahSeg* __restrict__ segs, int* __restrict__ flag)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int s = segOf[i];
        flag[i] = (s >= 0 && sah_goes_left(segs[s], primBox[ids[i]], i)) ? 1 : 0;
    }
}

// every primitive to its place in the next level's order; the children's centroid bounds on the way
__global__ void __launch_bounds__(kBlock) sah_scatter_kernel(const Box3* __restrict__ primBox, const int* __restrict__ ids, const int* __restrict__ segOf, const int n,
                                                             const SahSeg* __restrict__ segs, const int* __restrict__ flag, const int* __restrict__ scan,
                                                             int* __restrict__ idsOut, int* __restrict__ segOfOut, SahSeg* __restrict__ nextSegs)
{
    for (int tile = blockIdx.x * blockDim.x; tile < n; tile += gridDim.x * blockDim.x) {
        const int i = tile + (int)threadIdx.x;
        int child = -1;
        float c[3] = {0.0f, 0.0f, 0.0f};
        bool collect = false;
        if (i < n) {
            const int s = segOf[i];
            const int prim = ids[i];
            if (s < 0) {
                idsOut[i] = prim;
                segOfOut[i] = -1;
            } else {
                const SahSeg& sg = segs[s];
                const int leftBefore = scan[i] - scan[sg.first];
                const bool left = flag[i] != 0;
                const int pos = left ? sg.first + leftBefore : sg.first + sg.countL + ((i - sg.first) - leftBefore);
                child = sg.child[left ? 0 : 1];
                idsOut[pos] = prim;
                segOfOut[pos] = child;
                if (child >= 0 && sg.axis >= 0) {
                    box_centre(primBox[prim], c);
                    collect = true;
                }
            }
        }
        // centroid bounds of the child segments: one set of atomics per wave where the wave's primitives share a child
        const int child0 = __shfl(child, __ffsll((long long)__ballot(collect)) - 1);
        const bool waveUniform = __ballot(collect && child != child0) == 0ull;
        if (waveUniform) {
            float lo[3], hi[3];
            for (int a = 0; a < 3; a++) {
                lo[a] = collect ? c[a] : 1e30f;
                hi[a] = collect ? c[a] : -1e30f;
                for (int o = 32; o > 0; o >>= 1) {
                    lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
                    hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
                }
            }
            if (__ballot(collect) != 0ull && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(collect)) - 1)) {
                for (int a = 0; a < 3; a++) {
                    atomicMin(&nextSegs[child0].cb[a], float_ordered(lo[a]));
                    atomicMax(&nextSegs[child0].cb[3 + a], float_ordered(hi[a]));
                }
            }
        } else if (collect) {
            for (int a = 0; a < 3; a++) {
                atomicMin(&nextSegs[child].cb[a], float_ordered(c[a]));
                atomicMax(&nextSegs[child].cb[3 + a], float_ordered(c[a]));
            }
        }
    }
}

// Segments of at most kSahSmall primitives, one thread each: exact sweep over the three centroid orders (every partition of
// the sorted order is tried, count x area as above), children created depth first.
__global__ void __launch_bounds__(64) sah_small_kernel(const Box3* __restrict__ primBox, int* __restrict__ ids, const SahSmall* __restrict__ small, const uint32_t smallCount,
                                                       const int n, Bvh2 t, SahCounters* __restrict__ ctr)
{
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < smallCount; w += gridDim.x * blockDim.x) {
        const SahSmall sm = small[w];
        int id[kSahSmall];
        Box3 box[kSahSmall];
        for (int k = 0; k < sm.count; k++) {
            id[k] = ids[sm.first + k];
            box[k] = primBox[id[k]];
        }
        int stFirst[kSahSmall], stCount[kSahSmall], stNode[kSahSmall], top = 0;
        stFirst[top] = 0; stCount[top] = sm.count; stNode[top++] = sm.node;
        while (top > 0) {
            top--;
            const int first = stFirst[top], count = stCount[top], node = stNode[top];
            float bestCost = 3.0e38f;
            int bestPerm[kSahSmall], bestK = count / 2;
            for (int k = 0; k < count; k++) bestPerm[k] = first + k;
            for (int a = 0; a < 3; a++) {
                int perm[kSahSmall];
                float key[kSahSmall];
                for (int k = 0; k < count; k++) {  // insertion sort by the centroid coordinate
                    const float c = 0.5f * (box[first + k].lo[a] + box[first + k].hi[a]);
                    int j = k;
                    while (j > 0 && key[j - 1] > c) { key[j] = key[j - 1]; perm[j] = perm[j - 1]; j--; }
                    key[j] = c;
                    perm[j] = first + k;
                }
                float above[kSahSmall];
                Box3 run = empty_box();
                for (int k = count - 1; k >= 1; k--) {
                    grow(run, box[perm[k]]);
                    above[k] = (float)(count - k) * half_area(run);
                }
                run = empty_box();
                for (int k = 1; k < count; k++) {
                    grow(run, box[perm[k - 1]]);
                    const float cost = (float)k * half_area(run) + above[k];
                    if (cost < bestCost) {
                        bestCost = cost;
                        bestK = k;
                        for (int j = 0; j < count; j++) bestPerm[j] = perm[j];
                    }
                }
            }
            {  // the chosen order, in place
                int nid[kSahSmall];
                Box3 nbox[kSahSmall];
                for (int k = 0; k < count; k++) { nid[k] = id[bestPerm[k]]; nbox[k] = box[bestPerm[k]]; }
                for (int k = 0; k < count; k++) { id[first + k] = nid[k]; box[first + k] = nbox[k]; }
            }
            int childId[2];
            for (int h = 0; h < 2; h++) {
                const int cf = h == 0 ? first : first + bestK, cc = h == 0 ? bestK : count - bestK;
                if (cc == 1) {
                    childId[h] = (n - 1) + sm.first + cf;
                } else {
                    const int nid = (int)atomicAdd(&ctr->nodes, 1u);
                    childId[h] = nid;
                    Box3 b = empty_box();
                    for (int k = 0; k < cc; k++) grow(b, box[cf + k]);
                    t.box[nid] = b;
                    t.count[nid] = cc;
                    stFirst[top] = cf; stCount[top] = cc; stNode[top++] = nid;
                }
                t.parent[childId[h]] = node;
            }
            t.left[node] = childId[0];
            t.right[node] = childId[1];
        }
        for (int k = 0; k < sm.count; k++) ids[sm.first + k] = id[k];
    }
}

// the leaves' boxes, once the order is final
__global__ void __launch_bounds__(kBlock) sah_leaves_kernel(const Box3* __restrict__ primBox, const int* __restrict__ ids, const int n, Bvh2 t)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) t.box[(n - 1) + k] = primBox[ids[k]];
}

// ---- 4c. the collapse's cost table, bottom-up like the bounds: the second child to arrive at a node fills its seven entries
__device__ __forceinline__ Eval eval_of(const Bvh2& t, const int n, const int node, const int i, const bool fresh)
{
    if (node >= n - 1) return Eval{half_area(t.box[node]) * t.primCost, kDecLeaf, 0, 0, 0};  // one primitive: a leaf slot whatever i
    // (fresh: written by another thread of this launch — a device-scope load, which does not stop at this XCD's L2)
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(&t.eval[(size_t)node * 7 + i]);
    const unsigned long long w = fresh ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
    Eval e;
    memcpy(&e, &w, 8);
    return e;
}

__global__ void __launch_bounds__(kBlock) cost_kernel(const int n, Bvh2 t)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        int node = t.parent[(n - 1) + k];
        while (node >= 0) {
            // This thread's entries must be visible before its arrival is.  They are written with device-scope stores (below:
            // written through to where every XCD sees them), so waiting for them to complete is enough; the reader fetches them
            // with device-scope loads (eval_of).  The generic form — __threadfence() on either side, i.e. a write-back and an
            // invalidate of the XCD's whole L2 per node visit — made this kernel 7 ms per million primitives (round 4: 1.x ms).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (atomicAdd(&t.arrived[node], 1) == 0) break;
            const int l = t.left[node], r = t.right[node];
            float lc[7], rc[7];
            for (int i = 0; i < 7; i++) {
                lc[i] = eval_of(t, n, l, i, true).cost;
                rc[i] = eval_of(t, n, r, i, true).cost;
            }
            const float area = half_area(t.box[node]);
            const int prims = t.count[node];
            Eval e[7];
            // C(n, 0) = min(leaf, best split into 8 + one node test); C(n, i) = min(best split into i + 1, C(n, i - 1))
            for (int i = 0; i < 7; i++) {
                const int j = i == 0 ? 7 : i;  // roots to hand out: j, one of them at least to each side
                float best = 1.0e30f;
                int bl = 0, br = 0;
                for (int a = 0; a < j; a++) {
                    const float c = lc[a] + rc[j - 1 - a];
                    if (c < best) { best = c; bl = a; br = j - 1 - a; }
                }
                if (i == 0) {
                    // (the root is a wide node whatever the table says: as a "leaf" it would be a node with one slot holding all
                    //  its primitives — the table charges the node test to the other choice only)
                    // A leaf slot holds at most three primitives — by the count, not by a cost comparison: with infinite boxes (a
                    // damaged mesh) the cost of the wide node is +inf and "1e30 < inf" would make a leaf of any subtree.
                    const bool canLeaf = prims <= kLeafMax && t.parent[node] >= 0;
                    const float leaf = area * (float)prims * t.primCost;
                    const float inner = best + area * kCostNode;
                    e[0] = (canLeaf && leaf < inner) ? Eval{leaf, kDecLeaf, 0, 0, 0} : Eval{inner, kDecInternal, (int8_t)bl, (int8_t)br, 0};
                } else {
                    e[i] = best < e[i - 1].cost ? Eval{best, kDecDistribute, (int8_t)bl, (int8_t)br, 0} : e[i - 1];
                }
            }
            for (int i = 0; i < 7; i++) {
                unsigned long long w;
                memcpy(&w, &e[i], 8);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(&t.eval[(size_t)node * 7 + i]), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            node = t.parent[node];
        }
    }
}

// ---- 5. collapse into 8-wide nodes, one level per launch
struct WorkItem {
    int bvh2Node;   // root of the BVH2 subtree this BVH8 node covers (internal node id)
    uint32_t outNode;
    uint32_t mesh;  // batched build: the mesh the node belongs to (its own node numbers, primitive list and counters)
};
// One mesh of a batched build (lbvh_build_batch): its primitives are positions [first, first + count) of the concatenated arrays,
// its wide nodes are written to nodes[first + k] (k < count: a mesh never has more wide nodes than primitives), its primitive
// list to primIdx[first ...] with mesh-local indices.  big: its rank among the meshes that go through the tree build (= the
// binary-tree node that is its root), -1: at most eight primitives, a single node (small_mesh_node).
struct BatchMesh {
    uint32_t first, count;
    int big;
    uint32_t pad_;
};

__device__ __forceinline__ uint32_t quantize(float v)  // nexus::collapse Quantize
{
    if (!(v == v)) return 0u;
    if (v <= 0.0f) return 0u;
    if (v >= 255.0f) return 255u;
    return (uint32_t)v;
}

// a child of a wide node becomes a wide node itself, or a leaf slot of its (at most three) primitives
__device__ __forceinline__ bool is_inner(const Bvh2& t, const int n, const int c)
{
    if (c >= n - 1) return false;
    if (t.count[c] > kLeafMax) return true;  // (whatever the table says: a leaf slot cannot hold more)
    if (t.eval) return eval_of(t, n, c, 0, false).decision == kDecInternal;
    return false;
}

// meshCounters: per mesh {nodes used, prims used}; workCounter: work items of the next level.  meshes == nullptr: one mesh, whose
// arrays start at 0 and whose node capacity is nodeCapacity.
__global__ void __launch_bounds__(kBlock) collapse_level_kernel(const Bvh2 t, const int n, const uint32_t* __restrict__ order, const WorkItem* __restrict__ work,
                                                                const uint32_t workCount, WorkItem* __restrict__ nextWork, uint32_t* __restrict__ meshCounters,
                                                                uint32_t* __restrict__ workCounter, nx_bvh8_node* __restrict__ nodes, uint32_t* __restrict__ primIdx,
                                                                const uint32_t nodeCapacity, const BatchMesh* __restrict__ meshes)
{
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < workCount; w += gridDim.x * blockDim.x) {
        const WorkItem item = work[w];
        const uint32_t mesh = meshes ? item.mesh : 0u;
        const uint32_t meshFirst = meshes ? meshes[mesh].first : 0u, meshCapacity = meshes ? meshes[mesh].count : nodeCapacity;
        uint32_t* const counters = meshCounters + 2 * (size_t)mesh;
        int child[8];
        int count = 0;
        if (t.eval) {
            // the children the cost table chose (Collapse.cpp GatherChildren): follow the "distribute" decisions down from
            // entry 0 of this node; whatever is reached with another decision is a child
            int todoNode[8], todoIdx[8], top = 0;
            todoNode[top] = item.bvh2Node;
            todoIdx[top++] = 0;
            while (top > 0) {
                const int x = todoNode[--top];
                const Eval e = eval_of(t, n, x, todoIdx[top], false);
                const int side[2] = {t.left[x], t.right[x]}, sideIdx[2] = {e.leftCount, e.rightCount};
                for (int h = 0; h < 2; h++) {
                    if (eval_of(t, n, side[h], sideIdx[h], false).decision == kDecDistribute && top < 8) {
                        todoNode[top] = side[h];
                        todoIdx[top++] = sideIdx[h];
                    } else if (count < 8) {
                        child[count++] = side[h];
                    }
                }
            }
        } else {
            count = 2;
            child[0] = t.left[item.bvh2Node];
            child[1] = t.right[item.bvh2Node];
        }
        // without a cost table: open children by largest area until eight (a BVH2 leaf, one triangle, cannot be opened)
        while (!t.eval && count < 8) {
            int best = -1;
            float bestArea = -1.0f;
            for (int k = 0; k < count; k++) {
                if (child[k] >= n - 1) continue;
                const float a = half_area(t.box[child[k]]);
                if (a > bestArea) { bestArea = a; best = k; }
            }
            if (best < 0) break;
            const int open = child[best];
            child[best] = t.left[open];
            child[count++] = t.right[open];
        }
        const Box3 nb = t.box[item.bvh2Node];
        // octant slots: the reference's greedy assignment (BVH8Builder.cpp:170-252) — repeatedly the (child, slot) pair with the
        // smallest dot(child centre - node centre, slot direction), slot bit 2 / 1 / 0 set = -x / -y / -z
        float cx[8], cy[8], cz[8];
        const float ncx = 0.5f * (nb.lo[0] + nb.hi[0]), ncy = 0.5f * (nb.lo[1] + nb.hi[1]), ncz = 0.5f * (nb.lo[2] + nb.hi[2]);
        for (int k = 0; k < count; k++) {
            const Box3 b = t.box[child[k]];
            cx[k] = 0.5f * (b.lo[0] + b.hi[0]) - ncx; cy[k] = 0.5f * (b.lo[1] + b.hi[1]) - ncy; cz[k] = 0.5f * (b.lo[2] + b.hi[2]) - ncz;
        }
        int slotOf[8], childAt[8];
        for (int s = 0; s < 8; s++) childAt[s] = -1;
        for (int k = 0; k < 8; k++) slotOf[k] = -1;
        for (int round = 0; round < count; round++) {
            float best = 1e30f;
            int bk = -1, bs = -1;
            for (int k = 0; k < count; k++) {
                if (slotOf[k] >= 0) continue;
                for (int s = 0; s < 8; s++) {
                    if (childAt[s] >= 0) continue;
                    const float dx = (s & 4) ? -1.0f : 1.0f, dy = (s & 2) ? -1.0f : 1.0f, dz = (s & 1) ? -1.0f : 1.0f;
                    const float cost = cx[k] * dx + cy[k] * dy + cz[k] * dz;
                    if (cost < best) { best = cost; bk = k; bs = s; }
                }
            }
            if (bk < 0) {  // no finite cost (a box with NaNs): any free pair keeps the node well formed
                for (int k = 0; k < count && bk < 0; k++)
                    if (slotOf[k] < 0) bk = k;
                for (int s = 0; s < 8 && bs < 0; s++)
                    if (childAt[s] < 0) bs = s;
            }
            slotOf[bk] = bs;
            childAt[bs] = bk;
        }
        // inner children (more than three triangles) get consecutive node ids in slot order, leaf children consecutive
        // entries of the primitive list in slot order
        uint32_t innerCount = 0, primCount = 0;
        for (int s = 0; s < 8; s++) {
            if (childAt[s] < 0) continue;
            const int c = child[childAt[s]];
            const int tris = c >= n - 1 ? 1 : t.count[c];
            if (is_inner(t, n, c)) innerCount++;
            else primCount += (uint32_t)tris;
        }
        const uint32_t childBase = innerCount ? atomicAdd(&counters[0], innerCount) : 0u;
        const uint32_t primBase = primCount ? atomicAdd(&counters[1], primCount) : 0u;
        if (innerCount && childBase + innerCount > meshCapacity) continue;  // cannot happen: capacity covers the worst case
        const uint32_t workBase = innerCount ? atomicAdd(workCounter, innerCount) : 0u;

        nx_bvh8_node node;
        node = nx_bvh8_node{};
        // quantisation frame, as nexus::collapse: e = ceil(log2(extent / 255)) per axis
        float invScale[3];
        for (int a = 0; a < 3; a++) {
            const float ex = ceilf((float)log2((double)((nb.hi[a] - nb.lo[a]) * (1.0f / 255.0f))));
            uint32_t e = 0;
            if (ex == ex && ex > -127.0f) e = ex >= 128.0f ? 255u : (uint32_t)((int)ex + 127);
            const float pw = ex == ex ? (ex < -200.0f ? 0.0f : (ex > 200.0f ? __uint_as_float(0x7f800000u) : ldexpf(1.0f, (int)ex))) : ex;
            invScale[a] = 1.0f / pw;
            node.p[a] = nb.lo[a];
            node.e[a] = (uint8_t)e;
        }
        node.childBaseIdx = childBase;
        node.triangleBaseIdx = primBase;
        uint32_t innerSeen = 0, primSeen = 0;
        for (int s = 0; s < 8; s++) {
            if (childAt[s] < 0) continue;
            const int c = child[childAt[s]];
            const Box3 b = t.box[c];
            node.qlox[s] = (uint8_t)quantize(floorf((b.lo[0] - nb.lo[0]) * invScale[0]));
            node.qloy[s] = (uint8_t)quantize(floorf((b.lo[1] - nb.lo[1]) * invScale[1]));
            node.qloz[s] = (uint8_t)quantize(floorf((b.lo[2] - nb.lo[2]) * invScale[2]));
            node.qhix[s] = (uint8_t)quantize(ceilf((b.hi[0] - nb.lo[0]) * invScale[0]));
            node.qhiy[s] = (uint8_t)quantize(ceilf((b.hi[1] - nb.lo[1]) * invScale[1]));
            node.qhiz[s] = (uint8_t)quantize(ceilf((b.hi[2] - nb.lo[2]) * invScale[2]));
            const int tris = c >= n - 1 ? 1 : t.count[c];
            if (is_inner(t, n, c)) {
                node.meta[s] = (uint8_t)(0x20 | (24 + s));
                node.imask |= (uint8_t)(1u << s);
                nextWork[workBase + innerSeen] = WorkItem{c, childBase + innerSeen, mesh};
                innerSeen++;
            } else {
                // the (at most three) primitives under this child, left to right: a subtree of at most two internal nodes
                int leaves[3], found = 0, todo[3], top = 0;
                todo[top++] = c;
                while (top > 0 && found < 3) {
                    const int x = todo[--top];
                    if (x >= n - 1) leaves[found++] = x - (n - 1);
                    else {
                        todo[top++] = t.right[x];
                        todo[top++] = t.left[x];
                    }
                }
                uint32_t unary = 0;
                for (int j = 0; j < tris; j++) {
                    unary |= 1u << (j + 5);
                    primIdx[meshFirst + primBase + primSeen + (uint32_t)j] = order[leaves[j]] - meshFirst;
                }
                node.meta[s] = (uint8_t)(unary | primSeen);
                primSeen += (uint32_t)tris;
            }
        }
        nodes[meshFirst + item.outNode] = node;
    }
}

// tiny inputs: a root whose children are the triangles themselves (fewer than 2 internal BVH2 nodes to speak of)
__device__ void small_mesh_node(const Box3* __restrict__ triBox, const int n, nx_bvh8_node* __restrict__ nodes, uint32_t* __restrict__ primIdx)
{
    Box3 nb;
    for (int a = 0; a < 3; a++) { nb.lo[a] = 1e30f; nb.hi[a] = -1e30f; }
    for (int k = 0; k < n; k++)
        for (int a = 0; a < 3; a++) { nb.lo[a] = fminf(nb.lo[a], triBox[k].lo[a]); nb.hi[a] = fmaxf(nb.hi[a], triBox[k].hi[a]); }
    nx_bvh8_node node;
    node = nx_bvh8_node{};
    float invScale[3];
    for (int a = 0; a < 3; a++) {
        const float ex = ceilf((float)log2((double)((nb.hi[a] - nb.lo[a]) * (1.0f / 255.0f))));
        uint32_t e = 0;
        if (ex == ex && ex > -127.0f) e = ex >= 128.0f ? 255u : (uint32_t)((int)ex + 127);
        const float pw = ex == ex ? (ex < -200.0f ? 0.0f : (ex > 200.0f ? __uint_as_float(0x7f800000u) : ldexpf(1.0f, (int)ex))) : ex;
        invScale[a] = 1.0f / pw;
        node.p[a] = nb.lo[a];
        node.e[a] = (uint8_t)e;
    }
    for (int k = 0; k < n; k++) {  // n <= 8: one triangle per slot
        const Box3 b = triBox[k];
        node.qlox[k] = (uint8_t)quantize(floorf((b.lo[0] - nb.lo[0]) * invScale[0]));
        node.qloy[k] = (uint8_t)quantize(floorf((b.lo[1] - nb.lo[1]) * invScale[1]));
        node.qloz[k] = (uint8_t)quantize(floorf((b.lo[2] - nb.lo[2]) * invScale[2]));
        node.qhix[k] = (uint8_t)quantize(ceilf((b.hi[0] - nb.lo[0]) * invScale[0]));
        node.qhiy[k] = (uint8_t)quantize(ceilf((b.hi[1] - nb.lo[1]) * invScale[1]));
        node.qhiz[k] = (uint8_t)quantize(ceilf((b.hi[2] - nb.lo[2]) * invScale[2]));
        node.meta[k] = (uint8_t)(0x20u | (uint32_t)k);
        primIdx[k] = (uint32_t)k;
    }
    nodes[0] = node;
}
__global__ void small_mesh_kernel(const Box3* __restrict__ triBox, const int n, nx_bvh8_node* __restrict__ nodes, uint32_t* __restrict__ primIdx)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    small_mesh_node(triBox, n, nodes, primIdx);
}

// ---- 5b. the kernels only a BATCHED build needs (lbvh_build_batch): M meshes as one forest over the concatenated primitives.
// Everything between the Morton codes and the collapse is the single build's kernels unchanged — they work on segments of a
// position range and on node numbers, and a forest is M root segments and M roots instead of one.

// block m: every position of mesh m learns its mesh and its root segment (-1: a small mesh, not part of the tree build)
__global__ void __launch_bounds__(kBlock) batch_fill_kernel(const BatchMesh* __restrict__ meshes, int* __restrict__ meshOf, int* __restrict__ segOf)
{
    const BatchMesh m = meshes[blockIdx.x];
    for (uint32_t k = threadIdx.x; k < m.count; k += blockDim.x) {
        meshOf[m.first + k] = (int)blockIdx.x;
        segOf[m.first + k] = m.big;
    }
}

// tri_bounds_kernel + box_union_kernel per mesh: bounds[12 m ...] = centroid bounds (6 words), box (6 words), float_ordered
__global__ void __launch_bounds__(kBlock) tri_bounds_batch_kernel(const nx_triangle* __restrict__ tris, const uint32_t n, const int* __restrict__ meshOf, Box3* __restrict__ triBox,
                                                                  uint32_t* __restrict__ bounds)
{
    for (uint32_t tile = blockIdx.x * blockDim.x; tile < n; tile += gridDim.x * blockDim.x) {
        const uint32_t i = tile + threadIdx.x;
        const bool valid = i < n;
        float lo[6], hi[6];  // [0..2] centroid, [3..5] box
        int m = -1;
        if (valid) {
            const nx_triangle t = tris[i];
            m = meshOf[i];
            Box3 b;
            for (int a = 0; a < 3; a++) {
                b.lo[a] = fminf(fminf(t.pos0[a], t.pos1[a]), t.pos2[a]);
                b.hi[a] = fmaxf(fmaxf(t.pos0[a], t.pos1[a]), t.pos2[a]);
                lo[a] = hi[a] = 0.5f * (b.lo[a] + b.hi[a]);
                lo[3 + a] = b.lo[a];
                hi[3 + a] = b.hi[a];
            }
            triBox[i] = b;
        } else {
            for (int a = 0; a < 6; a++) { lo[a] = 1e30f; hi[a] = -1e30f; }
        }
        // one set of atomics per wave where the wave's triangles share a mesh
        const unsigned long long validMask = __ballot(valid);
        if (validMask == 0ull) continue;
        const int leader = __ffsll((long long)validMask) - 1;
        const int m0 = __shfl(m, leader);
        if (__ballot(valid && m != m0) == 0ull) {
            for (int a = 0; a < 6; a++) {
                for (int o = 32; o > 0; o >>= 1) {
                    lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
                    hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
                }
            }
            if ((int)(threadIdx.x & 63) == leader) {
                uint32_t* dst = bounds + 12 * (size_t)m0;
                for (int a = 0; a < 3; a++) {
                    atomicMin(&dst[a], float_ordered(lo[a]));
                    atomicMax(&dst[3 + a], float_ordered(hi[a]));
                    atomicMin(&dst[6 + a], float_ordered(lo[3 + a]));
                    atomicMax(&dst[9 + a], float_ordered(hi[3 + a]));
                }
            }
        } else if (valid) {
            uint32_t* dst = bounds + 12 * (size_t)m;
            for (int a = 0; a < 3; a++) {
                atomicMin(&dst[a], float_ordered(lo[a]));
                atomicMax(&dst[3 + a], float_ordered(hi[a]));
                atomicMin(&dst[6 + a], float_ordered(lo[3 + a]));
                atomicMax(&dst[9 + a], float_ordered(hi[3 + a]));
            }
        }
    }
}
__global__ void __launch_bounds__(kBlock) bounds_init_batch_kernel(uint32_t* __restrict__ bounds, const uint32_t meshCount)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < 12 * meshCount; k += gridDim.x * blockDim.x) bounds[k] = ((k % 6) < 3) ? 0xffffffffu : 0u;
}

// morton_kernel with every mesh's own centroid bounds
__global__ void __launch_bounds__(kBlock) morton_batch_kernel(const Box3* __restrict__ triBox, const uint32_t n, const int* __restrict__ meshOf, const uint32_t* __restrict__ bounds,
                                                              unsigned long long* __restrict__ codes, uint32_t* __restrict__ order)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t* sb = bounds + 12 * (size_t)meshOf[i];
        const Box3 b = triBox[i];
        uint32_t q[3];
        for (int a = 0; a < 3; a++) {
            const float lo = ordered_float(sb[a]);
            const float ext = ordered_float(sb[3 + a]) - lo;
            const float inv = ext > 0.0f ? 2097151.0f / ext : 0.0f;
            q[a] = (uint32_t)fminf(fmaxf((0.5f * (b.lo[a] + b.hi[a]) - lo) * inv, 0.0f), 2097151.0f);
        }
        codes[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
        order[i] = i;
    }
}
// second sort key: the mesh of the primitive at each position of the code-sorted order
__global__ void __launch_bounds__(kBlock) mesh_key_kernel(const uint32_t* __restrict__ order, const int* __restrict__ meshOf, const uint32_t n, uint32_t* __restrict__ key)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) key[i] = (uint32_t)meshOf[order[i]];
}

// sah_root_kernel for every mesh of the forest: root segment `big`, binary-tree root `big`; small meshes: their single node
__global__ void __launch_bounds__(kBlock) batch_roots_kernel(const BatchMesh* __restrict__ meshes, const uint32_t meshCount, const uint32_t* __restrict__ bounds,
                                                             const Box3* __restrict__ triBox, Bvh2 t, SahSeg* __restrict__ segs, WorkItem* __restrict__ work,
                                                             uint32_t* __restrict__ meshCounters, nx_bvh8_node* __restrict__ nodes, uint32_t* __restrict__ primIdx)
{
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < meshCount; m += gridDim.x * blockDim.x) {
        const BatchMesh bm = meshes[m];
        if (bm.big < 0) {
            // (triBox is still in input order here: position first + k is triangle k of the mesh)
            small_mesh_node(triBox + bm.first, (int)bm.count, nodes + bm.first, primIdx + bm.first);
            meshCounters[2 * (size_t)m] = 1u;
            meshCounters[2 * (size_t)m + 1] = bm.count;
            continue;
        }
        const uint32_t* sb = bounds + 12 * (size_t)m;
        SahSeg sg{};
        sg.first = (int)bm.first; sg.count = (int)bm.count; sg.node = bm.big;
        for (int k = 0; k < 6; k++) sg.cb[k] = sb[k];
        sg.child[0] = sg.child[1] = -1;
        segs[bm.big] = sg;
        Box3 b;
        for (int a = 0; a < 3; a++) { b.lo[a] = ordered_float(sb[6 + a]); b.hi[a] = ordered_float(sb[9 + a]); }
        t.box[bm.big] = b;
        t.count[bm.big] = (int)bm.count;
        t.parent[bm.big] = -1;
        work[bm.big] = WorkItem{bm.big, 0u, m};
        meshCounters[2 * (size_t)m] = 1u;      // node 0 is the root
        meshCounters[2 * (size_t)m + 1] = 0u;
    }
}

// the used nodes of every mesh from their staging places (nodes[first ...]) to their final, densely packed ones
__global__ void __launch_bounds__(kBlock) batch_pack_nodes_kernel(const BatchMesh* __restrict__ meshes, const uint32_t* __restrict__ nodePrefix, const uint32_t* __restrict__ meshCounters,
                                                                  const uint4* __restrict__ staging, uint4* __restrict__ packed)
{
    const uint32_t m = blockIdx.x;
    const size_t chunks = 5 * (size_t)meshCounters[2 * (size_t)m];
    const uint4* src = staging + 5 * (size_t)meshes[m].first;
    uint4* dst = packed + 5 * (size_t)nodePrefix[m];
    for (size_t k = threadIdx.x; k < chunks; k += blockDim.x) dst[k] = src[k];
}

// ---- 6. the traversal kernels' leaf-ordered intersection stream {p0 | id}, {e0}, {e1} (nxhip_upload_blas builds it on the host)
// (batched build: position k belongs to mesh meshOf[k], whose triangles start at meshes[...].first; the index stays mesh-local)
__global__ void __launch_bounds__(kBlock) isect_kernel(const nx_triangle* __restrict__ tris, const uint32_t* __restrict__ primIdx, const uint32_t n, float4* __restrict__ isect,
                                                       const int* __restrict__ meshOf, const BatchMesh* __restrict__ meshes)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t id = primIdx[k];
        const nx_triangle t = tris[(meshOf ? meshes[meshOf[k]].first : 0u) + id];
        isect[(size_t)kTriStride * k + 0] = make_float4(t.pos0[0], t.pos0[1], t.pos0[2], __uint_as_float(id));
        isect[(size_t)kTriStride * k + 1] = make_float4(t.pos1[0] - t.pos0[0], t.pos1[1] - t.pos0[1], t.pos1[2] - t.pos0[2], 0.0f);
        isect[(size_t)kTriStride * k + 2] = make_float4(t.pos2[0] - t.pos0[0], t.pos2[1] - t.pos0[1], t.pos2[2] - t.pos0[2], 0.0f);
    }
}

int grid_for(uint32_t n, int cus) { return (int)std::min<uint32_t>((n + kBlock - 1) / kBlock, (uint32_t)(8 * cus)); }

}  // namespace

// Buffers of the top-down SAH build (4d) for n primitives in `roots` trees, and its level loop: from the root segments in
// ws.segsA / ws.segOfA (sah_root_kernel, or batch_roots_kernel + batch_fill_kernel for a forest) and the primitives in idsA, to
// the finished binary trees; *idsOut = the final order (position -> primitive).  Binary-tree nodes 0 .. roots - 1 are the roots.
// Temporaries of one build carved out of ONE allocation (a batched build has some forty of them: forty hipFree calls, each a
// device synchronisation, cost more than the build's kernels).  Used in two passes over the same list of take() calls: the first
// (no block yet) only adds up the sizes, reserve() allocates, the second hands out the ranges.
struct Arena {
    std::shared_ptr<DevBuf> block;
    size_t used = 0;
    bool take(DevBuf& b, size_t bytes)
    {
        const size_t at = (used + 255) & ~(size_t)255;
        used = at + std::max<size_t>(bytes, 16);
        if (block) {
            if (used > block->bytes) return false;
            b = DevBuf::view(block, at, std::max<size_t>(bytes, 16));
        }
        return true;
    }
    bool reserve()
    {
        block = std::make_shared<DevBuf>();
        const bool ok = block->alloc(used + 256);
        used = 0;
        return ok;
    }
};

struct SahWorkspace {
    DevBuf segOfA, segOfB, flag, scan, scanTemp, segsA, segsB, small, ctr, bins;
    size_t maxSegs = 0, scanBytes = 0;
    uint32_t nodesMade = 0;  // internal nodes of the finished trees (roots included)
    // (arena == nullptr: buffers of their own)
    bool alloc(uint32_t n, uint32_t roots, Arena* arena = nullptr)
    {
        maxSegs = (size_t)n / (kSahSmall + 1) + roots + 2;
        const size_t binWords = (size_t)3 * kSahBins * kSahBinWords;
        if (scanBytes == 0 && !hip_ok(rocprim::exclusive_scan(nullptr, scanBytes, (int*)nullptr, (int*)nullptr, 0, (size_t)n, rocprim::plus<int>(), nullptr), "rocprim::exclusive_scan", __FILE__, __LINE__))
            return false;
        auto get = [&](DevBuf& b, size_t bytes) { return arena ? arena->take(b, bytes) : b.alloc(bytes); };
        return get(segOfA, (size_t)n * 4) && get(segOfB, (size_t)n * 4) && get(flag, (size_t)n * 4) && get(scan, (size_t)n * 4) && get(segsA, maxSegs * sizeof(SahSeg)) &&
               get(segsB, maxSegs * sizeof(SahSeg)) && get(small, ((size_t)n / 2 + 2) * sizeof(SahSmall)) && get(ctr, sizeof(SahCounters)) && get(bins, maxSegs * binWords * 4) &&
               get(scanTemp, std::max<size_t>(scanBytes, 16));
    }
};

static int sah_levels(nxhip_ctx* c, const DevBuf& triBox, const uint32_t n, Bvh2 t, SahWorkspace& ws, int* idsA, int* idsB, const uint32_t roots, int** idsOut)
{
    hipStream_t st = c->stream;
    const int cus = std::max(1, c->numCUs);
    const size_t binWords = (size_t)3 * kSahBins * kSahBinWords;
    const SahCounters ctrInit{roots, 0u, 0u, 0u};  // the roots hold the first node numbers
    NX_HIP(hipMemcpyAsync(ws.ctr.p, &ctrInit, sizeof ctrInit, hipMemcpyHostToDevice, st));
    NX_HIP(hipStreamSynchronize(st));  // (ctrInit is a local: the copy has left it)
    int *ids = idsA, *idsNext = idsB, *segOf = ws.segOfA.as<int>(), *segOfNext = ws.segOfB.as<int>();
    SahSeg *segs = ws.segsA.as<SahSeg>(), *segsNext = ws.segsB.as<SahSeg>();
    uint32_t segCount = roots, expectedNodes = 0;
    SahCounters h{};
    for (int level = 0; segCount > 0; level++) {
        if (level > 4096 || segCount > ws.maxSegs) {
            set_error("lbvh_build: the top-down build does not terminate");
            return NXHIP_ERR_INVALID;
        }
        sah_bins_init_kernel<<<grid_for((uint32_t)std::min<size_t>(segCount * binWords, 0x7fffffffu), cus), kBlock, 0, st>>>(ws.bins.as<uint32_t>(), segCount * binWords);
        sah_bin_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(triBox.as<Box3>(), ids, segOf, (int)n, segs, ws.bins.as<uint32_t>());
        sah_split_kernel<<<grid_for(segCount, cus), kBlock, 0, st>>>(segs, segCount, ws.bins.as<uint32_t>(), (int)n, t, segsNext, ws.small.as<SahSmall>(), ws.ctr.as<SahCounters>());
        sah_flag_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(triBox.as<Box3>(), ids, segOf, (int)n, segs, ws.flag.as<int>());
        NX_HIP(rocprim::exclusive_scan(ws.scanTemp.p, ws.scanBytes, ws.flag.as<int>(), ws.scan.as<int>(), 0, (size_t)n, rocprim::plus<int>(), st));
        sah_scatter_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(triBox.as<Box3>(), ids, segOf, (int)n, segs, ws.flag.as<int>(), ws.scan.as<int>(), idsNext, segOfNext, segsNext);
        NX_HIP(hipMemcpyAsync(&h, ws.ctr.p, sizeof h, hipMemcpyDeviceToHost, st));
        NX_HIP(hipStreamSynchronize(st));
        segCount = h.nextCount;
        NX_HIP(hipMemsetAsync(&ws.ctr.as<SahCounters>()->nextCount, 0, 4, st));
        std::swap(ids, idsNext);
        std::swap(segOf, segOfNext);
        std::swap(segs, segsNext);
    }
    if (h.smallCount) sah_small_kernel<<<(int)std::min<uint32_t>((h.smallCount + 63) / 64, (uint32_t)(64 * cus)), 64, 0, st>>>(triBox.as<Box3>(), ids, ws.small.as<SahSmall>(), h.smallCount, (int)n, t, ws.ctr.as<SahCounters>());
    sah_leaves_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(triBox.as<Box3>(), ids, (int)n, t);
    NX_HIP(hipMemcpyAsync(&h, ws.ctr.p, sizeof h, hipMemcpyDeviceToHost, st));
    NX_HIP(hipStreamSynchronize(st));
    (void)expectedNodes;
    *idsOut = ids;
    ws.nodesMade = h.nodes;
    return NXHIP_OK;
}

// The collapse's level loop: `workCount` work items (the roots) in workA, one launch per level of wide nodes.
static int collapse_levels(nxhip_ctx* c, const Bvh2& t, const uint32_t n, const uint32_t* leafOrder, WorkItem* workA, WorkItem* workB, uint32_t workCount, uint32_t* meshCounters,
                           uint32_t* workCounter, nx_bvh8_node* nodes, uint32_t* primIdx, const uint32_t nodeCapacity, const BatchMesh* meshes)
{
    hipStream_t st = c->stream;
    const int cus = std::max(1, c->numCUs);
    WorkItem *cur = workA, *nxt = workB;
    for (int level = 0; workCount > 0; level++) {
        if (level > 128) {
            set_error("lbvh_build: the collapse does not terminate");
            return NXHIP_ERR_INVALID;
        }
        collapse_level_kernel<<<grid_for(workCount, cus), kBlock, 0, st>>>(t, (int)n, leafOrder, cur, workCount, nxt, meshCounters, workCounter, nodes, primIdx, nodeCapacity, meshes);
        uint32_t next = 0;
        NX_HIP(hipMemcpyAsync(&next, workCounter, 4, hipMemcpyDeviceToHost, st));
        NX_HIP(hipStreamSynchronize(st));
        workCount = next;
        NX_HIP(hipMemsetAsync(workCounter, 0, 4, st));
        std::swap(cur, nxt);
    }
    return NXHIP_OK;
}

// The build from primitive boxes on: Morton codes, sort, radix tree, bounds, collapse.  `triBox` / `bounds` (centroid bounds,
// ordered-uint encoded) are on the device and filled by a kernel already queued on the stream.  primIdx: n entries.
static int lbvh_from_boxes(nxhip_ctx* c, const DevBuf& triBox, const DevBuf& bounds, uint32_t n, int plocRadius, float primCost, DevBuf& nodes, DevBuf& primIdx, uint32_t* nodeCount)
{
    hipStream_t st = c->stream;
    const int cus = std::max(1, c->numCUs);
    if (n <= 8) {
        if (!nodes.alloc(sizeof(nx_bvh8_node))) return NXHIP_ERR_HIP;
        small_mesh_kernel<<<1, 64, 0, st>>>(triBox.as<Box3>(), (int)n, nodes.as<nx_bvh8_node>(), primIdx.as<uint32_t>());
        *nodeCount = 1;
        return NXHIP_OK;
    }
    DevBuf codes, codesSorted, order, orderSorted, sortTemp;
    if (!codes.alloc((size_t)n * 8) || !codesSorted.alloc((size_t)n * 8) || !order.alloc((size_t)n * 4) || !orderSorted.alloc((size_t)n * 4)) return NXHIP_ERR_HIP;
    morton_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(triBox.as<Box3>(), n, bounds.as<uint32_t>(), codes.as<unsigned long long>(), order.as<uint32_t>());
    size_t tempBytes = 0;
    NX_HIP(rocprim::radix_sort_pairs(nullptr, tempBytes, codes.as<unsigned long long>(), codesSorted.as<unsigned long long>(), order.as<uint32_t>(), orderSorted.as<uint32_t>(), n, 0, 63, st));
    if (!sortTemp.alloc(std::max<size_t>(tempBytes, 16))) return NXHIP_ERR_HIP;
    NX_HIP(rocprim::radix_sort_pairs(sortTemp.p, tempBytes, codes.as<unsigned long long>(), codesSorted.as<unsigned long long>(), order.as<uint32_t>(), orderSorted.as<uint32_t>(), n, 0, 63, st));

    // the binary tree over the sorted primitives: radix tree + bottom-up bounds, or locally-ordered clustering
    const size_t inner = (size_t)n - 1, all = 2 * (size_t)n - 1;
    DevBuf left, right, parent, first, last, count, box, arrived;
    if (!left.alloc(inner * 4) || !right.alloc(inner * 4) || !parent.alloc(all * 4) || !first.alloc(inner * 4) || !last.alloc(inner * 4) || !count.alloc(inner * 4) ||
        !box.alloc(all * sizeof(Box3)) || !arrived.alloc(inner * 4)) return NXHIP_ERR_HIP;
    Bvh2 t{left.as<int>(), right.as<int>(), parent.as<int>(), first.as<int>(), last.as<int>(), count.as<int>(), box.as<Box3>(), arrived.as<int>(), nullptr, primCost};
    const uint32_t* leafOrder = orderSorted.as<uint32_t>();  // leaf k of the binary tree = primitive leafOrder[k]
    DevBuf sahIdsA, sahIdsB;
    if (plocRadius < 0) {
        // top-down binned SAH (4d), level by level
        SahWorkspace ws;
        if (!sahIdsA.alloc((size_t)n * 4) || !sahIdsB.alloc((size_t)n * 4) || !ws.alloc(n, 1)) return NXHIP_ERR_HIP;
        DevBuf rootBox;
        if (!rootBox.alloc(6 * 4)) return NXHIP_ERR_HIP;
        const uint32_t boxInit[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
        NX_HIP(hipMemcpyAsync(rootBox.p, boxInit, sizeof boxInit, hipMemcpyHostToDevice, st));
        NX_HIP(hipMemcpyAsync(sahIdsA.p, orderSorted.p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));  // start from the Morton order: neighbours in memory are neighbours in space
        box_union_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(triBox.as<Box3>(), n, rootBox.as<uint32_t>());
        sah_root_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(bounds.as<uint32_t>(), rootBox.as<uint32_t>(), (int)n, t, ws.segsA.as<SahSeg>(), ws.segOfA.as<int>());
        int* ids = nullptr;
        const int rcLevels = sah_levels(c, triBox, n, t, ws, sahIdsA.as<int>(), sahIdsB.as<int>(), 1u, &ids);
        if (rcLevels != NXHIP_OK) return rcLevels;
        if (ws.nodesMade != n - 1) {
            set_error("lbvh_build: the top-down build made " + std::to_string(ws.nodesMade) + " internal nodes for " + std::to_string(n) + " primitives");
            return NXHIP_ERR_INVALID;
        }
        leafOrder = reinterpret_cast<const uint32_t*>(ids);
    } else if (plocRadius == 0) {
        NX_HIP(hipMemsetAsync(arrived.p, 0, inner * 4, st));
        radix_tree_kernel<<<grid_for(n - 1, cus), kBlock, 0, st>>>(codesSorted.as<unsigned long long>(), (int)n, t);
        fit_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(triBox.as<Box3>(), orderSorted.as<uint32_t>(), (int)n, t);
    } else {
        DevBuf clusterA, clusterB, nearest, merged, keep, offset, mergeCounter, scanTemp;
        if (!clusterA.alloc((size_t)n * 4) || !clusterB.alloc((size_t)n * 4) || !nearest.alloc((size_t)n * 4) || !merged.alloc((size_t)n * 4) || !keep.alloc((size_t)n * 4) ||
            !offset.alloc((size_t)n * 4) || !mergeCounter.alloc(4)) return NXHIP_ERR_HIP;
        size_t scanBytes = 0;
        NX_HIP(rocprim::exclusive_scan(nullptr, scanBytes, keep.as<int>(), offset.as<int>(), 0, (size_t)n, rocprim::plus<int>(), st));
        if (!scanTemp.alloc(std::max<size_t>(scanBytes, 16))) return NXHIP_ERR_HIP;
        NX_HIP(hipMemsetAsync(mergeCounter.p, 0, 4, st));
        int* cur = clusterA.as<int>();
        int* nxt = clusterB.as<int>();
        ploc_leaves_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(triBox.as<Box3>(), orderSorted.as<uint32_t>(), (int)n, t, cur);
        int m = (int)n;
        for (int round = 0; m > 1; round++) {
            if (round > 4096) {
                set_error("lbvh_build: the clustering does not terminate");
                return NXHIP_ERR_INVALID;
            }
            ploc_neighbour_kernel<<<grid_for((uint32_t)m, cus), kBlock, 0, st>>>(t, cur, m, plocRadius, nearest.as<int>());
            ploc_merge_kernel<<<grid_for((uint32_t)m, cus), kBlock, 0, st>>>(t, (int)n, cur, nearest.as<int>(), m, merged.as<int>(), keep.as<int>(), mergeCounter.as<int>());
            NX_HIP(rocprim::exclusive_scan(scanTemp.p, scanBytes, keep.as<int>(), offset.as<int>(), 0, (size_t)m, rocprim::plus<int>(), st));
            ploc_compact_kernel<<<grid_for((uint32_t)m, cus), kBlock, 0, st>>>(merged.as<int>(), keep.as<int>(), offset.as<int>(), m, nxt);
            int tail[2];
            NX_HIP(hipMemcpyAsync(&tail[0], offset.as<int>() + (m - 1), 4, hipMemcpyDeviceToHost, st));
            NX_HIP(hipMemcpyAsync(&tail[1], keep.as<int>() + (m - 1), 4, hipMemcpyDeviceToHost, st));
            NX_HIP(hipStreamSynchronize(st));
            const int next = tail[0] + tail[1];
            if (next >= m || next < 1) {
                set_error("lbvh_build: a clustering round merged nothing");
                return NXHIP_ERR_INVALID;
            }
            m = next;
            std::swap(cur, nxt);
        }
    }

    // which subtrees become wide nodes, which leaf slots: the cost table of the SAH collapse, bottom-up over the finished tree
    // (NX_DEVICE_COLLAPSE=greedy with NX_TUNING_KNOBS=1: the round-2 rule instead — open the largest child until there are eight,
    // subtrees of at most three primitives become leaf slots — kept for the quality comparison of tools/builder_quality.py)
    DevBuf evals;
    bool greedy = false;
    if (const char* on = std::getenv("NX_TUNING_KNOBS"); on && std::atoi(on) == 1)
        if (const char* e = std::getenv("NX_DEVICE_COLLAPSE")) greedy = std::strcmp(e, "greedy") == 0;
    if (!greedy) {
        if (!evals.alloc(inner * 7 * sizeof(Eval))) return NXHIP_ERR_HIP;
        t.eval = evals.as<Eval>();
        NX_HIP(hipMemsetAsync(arrived.p, 0, inner * 4, st));
        cost_kernel<<<grid_for(n, cus), kBlock, 0, st>>>((int)n, t);
    }

    // collapse.  Every BVH8 node but the root stands for a distinct internal BVH2 node, so n - 1 nodes cannot be exceeded
    // (typical use: 0.1 - 0.2 n); the caller shrinks the array to the used size.
    const uint32_t nodeCapacity = (uint32_t)inner;
    DevBuf workA, workB, counters;
    if (!nodes.alloc((size_t)nodeCapacity * sizeof(nx_bvh8_node)) || !workA.alloc((size_t)nodeCapacity * sizeof(WorkItem)) || !workB.alloc((size_t)nodeCapacity * sizeof(WorkItem)) ||
        !counters.alloc(3 * 4)) return NXHIP_ERR_HIP;
    {
        static const uint32_t init[3] = {1u, 0u, 0u};  // node 0 is the root
        static const WorkItem root{0, 0u, 0u};
        NX_HIP(hipMemcpyAsync(counters.p, init, sizeof init, hipMemcpyHostToDevice, st));
        NX_HIP(hipMemcpyAsync(workA.p, &root, sizeof root, hipMemcpyHostToDevice, st));
    }
    const int rcCollapse = collapse_levels(c, t, n, leafOrder, workA.as<WorkItem>(), workB.as<WorkItem>(), 1u, counters.as<uint32_t>(), counters.as<uint32_t>() + 2, nodes.as<nx_bvh8_node>(),
                                           primIdx.as<uint32_t>(), nodeCapacity, nullptr);
    if (rcCollapse != NXHIP_OK) return rcCollapse;
    uint32_t used = 0;
    NX_HIP(hipMemcpy(&used, counters.p, 4, hipMemcpyDeviceToHost));
    if (used > nodeCapacity) {
        set_error("lbvh_build: node capacity exceeded");
        return NXHIP_ERR_INVALID;
    }
    *nodeCount = used;
    return NXHIP_OK;
}

// Builds nodes / primIdx / isect for the `n` triangles at dTris (device).  All device buffers come back through the
// arguments; *nodeCount = nodes used.
int lbvh_build(nxhip_ctx* c, const nx_triangle* dTris, uint32_t n, int plocRadius, DevBuf& nodes, DevBuf& primIdx, DevBuf& isect, uint32_t* nodeCount)
{
    hipStream_t st = c->stream;
    const int cus = std::max(1, c->numCUs);
    DevBuf triBox, bounds;
    if (!triBox.alloc((size_t)n * sizeof(Box3)) || !bounds.alloc(6 * 4) || !primIdx.alloc((size_t)n * 4) || !isect.alloc((size_t)n * kTriStride * sizeof(float4))) return NXHIP_ERR_HIP;
    const uint32_t init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    NX_HIP(hipMemcpyAsync(bounds.p, init, sizeof init, hipMemcpyHostToDevice, st));
    tri_bounds_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(dTris, n, triBox.as<Box3>(), bounds.as<uint32_t>());
    const float primCost = blas_prim_cost();
    const int rc = lbvh_from_boxes(c, triBox, bounds, n, plocRadius, primCost, nodes, primIdx, nodeCount);
    if (rc != NXHIP_OK) return rc;
    isect_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(dTris, primIdx.as<uint32_t>(), n, isect.as<float4>(), nullptr, nullptr);
    NX_HIP(hipStreamSynchronize(st));
    NX_HIP(hipGetLastError());
    return NXHIP_OK;
}

// The same builder over instance boxes: a TLAS for `n` instances (device array, world bounds filled in).  nodes / primIdx
// (the TLAS's instance index list, leaf order) stay on the device; *nodeCount = nodes used.
int lbvh_build_tlas(nxhip_ctx* c, const nx_bvh_instance* dInstances, uint32_t n, int plocRadius, DevBuf& nodes, DevBuf& primIdx, DevBuf& box, bool* boxesAreTight, uint32_t* nodeCount)
{
    hipStream_t st = c->stream;
    const int cus = std::max(1, c->numCUs);
    DevBuf bounds;
    if (!box.alloc((size_t)n * sizeof(Box3)) || !bounds.alloc(6 * 4) || !primIdx.alloc((size_t)n * 4)) return NXHIP_ERR_HIP;
    const uint32_t init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    NX_HIP(hipMemcpyAsync(bounds.p, init, sizeof init, hipMemcpyHostToDevice, st));
    float instanceCost = kCostInstance;
    bool tight = true;
    if (const char* on = std::getenv("NX_TUNING_KNOBS"); on && std::atoi(on) == 1) {  // sweeps only
        if (const char* e = std::getenv("NX_TLAS_PRIM_COST")) instanceCost = (float)std::atof(e);
        if (const char* e = std::getenv("NX_TLAS_TIGHT")) tight = std::atoi(e) != 0;
    }
    tight = tight && kNodeStride == 5;
    *boxesAreTight = tight;
    instance_bounds_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(dInstances, n, tight ? c->blasTable.as<BlasDev>() : nullptr, box.as<Box3>(), bounds.as<uint32_t>());
    const int rc = lbvh_from_boxes(c, box, bounds, n, plocRadius, instanceCost, nodes, primIdx, nodeCount);
    if (rc != NXHIP_OK) return rc;
    NX_HIP(hipStreamSynchronize(st));
    NX_HIP(hipGetLastError());
    return NXHIP_OK;
}

// M meshes in ONE build (nxhip_build_blas_batch).  The reference makes one BVH8 per aiMesh (Assets/OBJLoader.cpp:213-239,
// Assets/AssetManager.cpp:23-37): a glTF scene is thousands of small meshes, and one build per mesh pays its ~25 level
// synchronisations and its allocations every time.  Here the meshes are a FOREST over the concatenated triangles: per-mesh
// centroid bounds and Morton codes, a stable sort by code and then by mesh (each mesh keeps the order its own build would start
// from), one root segment and one root per mesh, and then the single build's level loops — binning, split, partition; cost
// table; collapse — over all meshes at once: the level count is the deepest mesh's, not the sum.  Meshes of at most eight
// triangles are a single node, as in the single build.  Every mesh gets its own node numbers and its own primitive list
// (mesh-local indices).  What comes back: pooled arrays — nodes (packed, mesh after mesh), primIdx and isect (positions of the
// concatenation) — and per mesh the first node and the node count.  Top-down SAH only (the default builder).
int lbvh_build_batch(nxhip_ctx* c, const nx_triangle* dTris, const std::vector<uint32_t>& counts, DevBuf& nodes, DevBuf& primIdx, DevBuf& isect, std::vector<uint32_t>& nodeFirst,
                     std::vector<uint32_t>& nodeCounts)
{
    hipStream_t st = c->stream;
    const int cus = std::max(1, c->numCUs);
    const bool timing = std::getenv("NX_TUNING_KNOBS") && std::atoi(std::getenv("NX_TUNING_KNOBS")) == 1 && std::getenv("NX_BATCH_TIMING");
    auto tLap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[lbvh_build_batch]       %-34s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - tLap).count());
        tLap = t1;
    };
    const uint32_t M = (uint32_t)counts.size();
    std::vector<BatchMesh> meshes(M);
    uint64_t total = 0;
    uint32_t big = 0;
    for (uint32_t m = 0; m < M; m++) {
        meshes[m] = BatchMesh{(uint32_t)total, counts[m], counts[m] > 8u ? (int)big++ : -1, 0u};
        total += counts[m];
    }
    if (total > 0x7fffffffull) {
        set_error("nxhip_build_blas_batch: more than 2^31 triangles in one batch");
        return NXHIP_ERR_INVALID;
    }
    const uint32_t n = (uint32_t)total;
    DevBuf dMeshes, meshOf, triBox, bounds, staging, meshCounters, workCounter;
    DevBuf codes, codesSorted, order, orderSorted, meshKey, meshKeySorted, idsA, idsB, sortTemp;
    DevBuf left, right, parent, count, box, arrived, evals, workA, workB, prefix;
    SahWorkspace ws;
    int meshBits = 1;
    while ((1ull << meshBits) < (unsigned long long)M) meshBits++;
    size_t tempA = 0, tempB = 0;
    NX_HIP(rocprim::radix_sort_pairs(nullptr, tempA, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 63, st));
    NX_HIP(rocprim::radix_sort_pairs(nullptr, tempB, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, meshBits, st));
    const size_t inner = (size_t)std::max(n, 2u) - 1, all = 2 * (size_t)std::max(n, 2u) - 1;
    // every temporary of the build from one allocation (Arena): sizes first, then the ranges
    Arena arena;
    auto layout = [&]() {
        return arena.take(dMeshes, (size_t)M * sizeof(BatchMesh)) && arena.take(meshOf, (size_t)n * 4) && arena.take(triBox, (size_t)n * sizeof(Box3)) && arena.take(bounds, (size_t)M * 12 * 4) &&
               arena.take(staging, (size_t)n * sizeof(nx_bvh8_node)) && arena.take(meshCounters, (size_t)M * 2 * 4) && arena.take(workCounter, 4) && ws.alloc(n, std::max(1u, big), &arena) &&
               arena.take(codes, (size_t)n * 8) && arena.take(codesSorted, (size_t)n * 8) && arena.take(order, (size_t)n * 4) && arena.take(orderSorted, (size_t)n * 4) &&
               arena.take(meshKey, (size_t)n * 4) && arena.take(meshKeySorted, (size_t)n * 4) && arena.take(idsA, (size_t)n * 4) && arena.take(idsB, (size_t)n * 4) &&
               arena.take(sortTemp, std::max<size_t>(std::max(tempA, tempB), 16)) && arena.take(left, inner * 4) && arena.take(right, inner * 4) && arena.take(parent, all * 4) &&
               arena.take(count, inner * 4) && arena.take(box, all * sizeof(Box3)) && arena.take(arrived, inner * 4) && arena.take(evals, inner * 7 * sizeof(Eval)) &&
               arena.take(workA, (size_t)n * sizeof(WorkItem)) && arena.take(workB, (size_t)n * sizeof(WorkItem)) && arena.take(prefix, (size_t)M * 4);
    };
    if (!layout() || !arena.reserve() || !layout()) return NXHIP_ERR_HIP;
    if (!primIdx.alloc((size_t)n * 4) || !isect.alloc((size_t)n * kTriStride * sizeof(float4))) return NXHIP_ERR_HIP;
    // (zeroed: isect_kernel below runs before the per-mesh counters are read back and checked, and an entry the collapse left
    //  unwritten — its 'cannot happen' branch, a mesh that came out short — must index a triangle of the mesh, not wherever
    //  the allocation's previous contents point; the build then ends in NXHIP_ERR_INVALID instead of a wild device read)
    NX_HIP(hipMemsetAsync(primIdx.p, 0, (size_t)n * 4, c->stream));
    lap("allocations (1)");
    NX_HIP(hipMemcpyAsync(dMeshes.p, meshes.data(), (size_t)M * sizeof(BatchMesh), hipMemcpyHostToDevice, st));  // (`meshes` outlives the synchronisations below)
    batch_fill_kernel<<<M, kBlock, 0, st>>>(dMeshes.as<BatchMesh>(), meshOf.as<int>(), ws.segOfA.as<int>());
    bounds_init_batch_kernel<<<grid_for(12 * M, cus), kBlock, 0, st>>>(bounds.as<uint32_t>(), M);
    tri_bounds_batch_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(dTris, n, meshOf.as<int>(), triBox.as<Box3>(), bounds.as<uint32_t>());

    lap("boxes and bounds");
    // the order every mesh's own build would start from: its triangles by Morton code (a stable sort by code, then by mesh)
    morton_batch_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(triBox.as<Box3>(), n, meshOf.as<int>(), bounds.as<uint32_t>(), codes.as<unsigned long long>(), order.as<uint32_t>());
    NX_HIP(rocprim::radix_sort_pairs(sortTemp.p, tempA, codes.as<unsigned long long>(), codesSorted.as<unsigned long long>(), order.as<uint32_t>(), orderSorted.as<uint32_t>(), n, 0, 63, st));
    mesh_key_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(orderSorted.as<uint32_t>(), meshOf.as<int>(), n, meshKey.as<uint32_t>());
    NX_HIP(rocprim::radix_sort_pairs(sortTemp.p, tempB, meshKey.as<uint32_t>(), meshKeySorted.as<uint32_t>(), orderSorted.as<uint32_t>(), idsA.as<uint32_t>(), n, 0, meshBits, st));

    lap("Morton codes, two sorts (+ allocations)");
    // the forest
    Bvh2 t{left.as<int>(), right.as<int>(), parent.as<int>(), nullptr, nullptr, count.as<int>(), box.as<Box3>(), arrived.as<int>(), evals.as<Eval>(), blas_prim_cost()};
    NX_HIP(hipMemsetAsync(parent.p, 0xff, all * 4, st));  // -1: the triangles of small meshes hang under no tree
    NX_HIP(hipMemsetAsync(workCounter.p, 0, 4, st));
    batch_roots_kernel<<<grid_for(M, cus), kBlock, 0, st>>>(dMeshes.as<BatchMesh>(), M, bounds.as<uint32_t>(), triBox.as<Box3>(), t, ws.segsA.as<SahSeg>(), workA.as<WorkItem>(),
                                                            meshCounters.as<uint32_t>(), staging.as<nx_bvh8_node>(), primIdx.as<uint32_t>());
    uint64_t bigPrims = 0;
    for (const BatchMesh& m : meshes) bigPrims += m.big >= 0 ? m.count : 0u;
    lap("roots (+ allocations)");
    if (big) {
        int* ids = nullptr;
        int rc = sah_levels(c, triBox, n, t, ws, idsA.as<int>(), idsB.as<int>(), big, &ids);
        if (rc != NXHIP_OK) return rc;
        lap("binary trees: SAH levels");
        if ((uint64_t)ws.nodesMade != bigPrims - big) {
            set_error("nxhip_build_blas_batch: the top-down build made " + std::to_string(ws.nodesMade) + " internal nodes for " + std::to_string(bigPrims) + " triangles in " + std::to_string(big) + " meshes");
            return NXHIP_ERR_INVALID;
        }
        NX_HIP(hipMemsetAsync(arrived.p, 0, inner * 4, st));
        cost_kernel<<<grid_for(n, cus), kBlock, 0, st>>>((int)n, t);
        lap("cost table");
        rc = collapse_levels(c, t, n, reinterpret_cast<const uint32_t*>(ids), workA.as<WorkItem>(), workB.as<WorkItem>(), big, meshCounters.as<uint32_t>(), workCounter.as<uint32_t>(),
                             staging.as<nx_bvh8_node>(), primIdx.as<uint32_t>(), 0u, dMeshes.as<BatchMesh>());
        if (rc != NXHIP_OK) return rc;
        lap("collapse levels");
    }
    isect_kernel<<<grid_for(n, cus), kBlock, 0, st>>>(dTris, primIdx.as<uint32_t>(), n, isect.as<float4>(), meshOf.as<int>(), dMeshes.as<BatchMesh>());

    // pack the nodes: every mesh's used nodes, mesh after mesh
    std::vector<uint32_t> ctr((size_t)M * 2);
    NX_HIP(hipMemcpyAsync(ctr.data(), meshCounters.p, ctr.size() * 4, hipMemcpyDeviceToHost, st));
    NX_HIP(hipStreamSynchronize(st));
    nodeFirst.assign(M, 0u);
    nodeCounts.assign(M, 0u);
    uint64_t nodeTotal = 0;
    for (uint32_t m = 0; m < M; m++) {
        if (ctr[2 * (size_t)m] > counts[m] || ctr[2 * (size_t)m + 1] != counts[m]) {
            set_error("nxhip_build_blas_batch: mesh " + std::to_string(m) + " came out with " + std::to_string(ctr[2 * (size_t)m]) + " nodes and " + std::to_string(ctr[2 * (size_t)m + 1]) + " of its " +
                      std::to_string(counts[m]) + " triangles");
            return NXHIP_ERR_INVALID;
        }
        nodeFirst[m] = (uint32_t)nodeTotal;
        nodeCounts[m] = ctr[2 * (size_t)m];
        nodeTotal += nodeCounts[m];
    }
    if (!nodes.alloc((size_t)nodeTotal * sizeof(nx_bvh8_node))) return NXHIP_ERR_HIP;
    NX_HIP(hipMemcpyAsync(prefix.p, nodeFirst.data(), (size_t)M * 4, hipMemcpyHostToDevice, st));
    batch_pack_nodes_kernel<<<M, kBlock, 0, st>>>(dMeshes.as<BatchMesh>(), prefix.as<uint32_t>(), meshCounters.as<uint32_t>(), staging.as<uint4>(), nodes.as<uint4>());
    NX_HIP(hipStreamSynchronize(st));
    NX_HIP(hipGetLastError());
    lap("intersection stream, node packing");
    return NXHIP_OK;
}

// the device-side layouts this translation unit was compiled with (nx_device.h layout_stamp; compared by nxhip_create)
uint64_t layout_stamp_lbvh() { return layout_stamp(); }

}  // namespace nxd
