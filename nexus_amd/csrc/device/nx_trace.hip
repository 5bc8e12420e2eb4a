// nx_trace.hip — closest-hit and any-hit traversal of the two-level compressed BVH8 (gfx950 / CDNA4).
//
// What it computes is the reference's TraceKernel / TraceShadowKernel
// (/root/reference/Nexus/src/Cuda/PathTracer/PathTracer.cu:125-133 ->
//  Cuda/BVH/BVH8Traversal.cuh:55-146 ChildTrace, :148-322 BVH8Trace, :326-518 BVH8TraceShadow,
//  Cuda/Geometry/Triangle.cuh:53-118 Moeller-Trumbore): per ray the same nodes are visited in the same
// order and the same hit record results.  How it is organised is CDNA4-first:
//   * one ray per lane of a 64-wide wave; persistent workgroups; a wave reserves 256 rays with ONE atomic and hands them
//     to idle lanes by ballot + popcount rank (the reference does one atomicAdd per ray); waves a small queue does not
//     need leave at once, and a wave whose shard is dry picks the next one from a single load of all fetch heads;
//   * the ray queue is cut into 8 contiguous shards, one fetch head per XCD group (blockIdx % 8 share an
//     XCD and its private 4 MiB L2): waves of one XCD walk one band of the image / queue so the BVH
//     subtrees they touch stay in that XCD's L2, and the head word is not hammered by 256 CUs; a wave
//     whose shard runs dry steals from the next one;
//   * every lane's loop iteration is "pop if out of work -> fetch ONE record (node, instance or triangle) -> process
//     it": the three fetch kinds of a wave are issued together and waited for once, so an iteration has a single
//     memory round trip, and the 64 lanes reconverge at every stage (the reference's `while (triangleEntry.y)` inner
//     loop serialises lanes with long triangle lists on a 64-wide wave).  Per-ray visiting order is unchanged;
//   * a record is an 80-byte node (five 16-byte loads), a 48-byte triangle record or a 160-byte instance record (transform,
//     BLAS pointers and a copy of the BLAS root node: entering an instance and testing its root share an iteration); each
//     lane loads its own with 16-byte global loads;
//   * quantised bounds are converted with v_cvt_f32_ubyteN and the slab test is 6 v_fma per child + integer
//     max3/min3 on the float bit patterns (identical ordering to the reference's vmax.s32/vmin.s32 PTX, including
//     its NaN behaviour);
//   * triangles come from a leaf-ordered 48-byte stream (p0|id, e0, e1) built at upload: no triangleIdx
//     indirection; instances from an 80-byte leaf-ordered traversal record;
//   * the traversal stack lives in LDS, entry-major ([depth][lane]) so ds_read/write_b64 are
//     conflict-free, with a scratch overflow; the world-space ray of a lane inside a transformed instance is
//     parked in LDS instead of being kept in 9 VGPRs.
// No MFMA: this is pointer chasing, bounded by memory latency / bandwidth.
#define NX_KERNEL_TU 1
#include "nx_queue.h"
#include "nx_traverse.h"

namespace nxd {

#ifndef NX_RESERVE
#define NX_RESERVE 512
#endif
// most rays one fetch atomic reserves.  Rounds 1-4: a block per WAVE, 256 (measured on large queues: 128 -3 %, 512 -1 %, 1024 -8 %).
// Round 5: a block per WORKGROUP (NX_WG_RANGE, below), whose four waves share it: 512 (128: -2.7 %, 256: -1 %, 1024: =)
constexpr int kReserve = NX_RESERVE;
#ifndef NX_REFILL_BELOW
#define NX_REFILL_BELOW 40
#endif
#ifndef NX_WG_RANGE
#define NX_WG_RANGE 1
#endif
static_assert((kRaySurvives << 30) == 0x80000000u, "the roulette bit of rayO.w moves to bit 31 of the lane's ray index");
constexpr int kRefillBelow = NX_REFILL_BELOW;  // refill idle lanes when fewer than this many of the 64 are still traversing

// ------------------------------------------------------------------------------------------------------
// The last long rays of a dry wave, all 64 lanes on ONE ray.
//
// A launch ends with its slowest ray, and a ray that skims the displaced surface visits thousands of records at one record per
// ~microsecond: every level of a pass has a floor of a few hundred microseconds of one-to-four-lane waves, and some frame ranges
// hold a ray that adds two milliseconds to its level (dropping such rays — wrong results — makes every
// repetition of the driver's command 18.4 ms instead of 19.1 ... 21.3).  Splitting a ray INSIDE the loop was tried in round 3 and
// cost the loop 11 % by its live values; the search below as a part of this kernel cost it 15 % out of line and 160 % inlined
// (round 5: its registers around the refill point, not its execution).  So the trace kernels only HAND OVER: when the queue is
// dry, at most 16 lanes of a wave are still busy and they have been for 16 iterations and four average rays, the wave appends those
// rays — WITH their traversal state (round 6: ThinState: stack, current groups, hit so far, instance) — to a list and ends.
// thin_kernel — one launch per level, behind the closest-hit and any-hit launches — puts a whole wave on each listed ray: all 64
// lanes search what is LEFT of the ray's tree together, in any order: a pool of work items (node, instance entry, triangle) in LDS,
// seeded with the groups of the ray's stack; every round each lane takes one item, tests it against the ray with the loop's own
// functions and puts the children the ray enters back; the smallest triangle distance found so far — from the start: the hit the
// ray arrived with — prunes, as the ray's own hit distance would.  (Rounds 5: the search started again at the TLAS root with no
// bound.  Measured, round 6: carrying the state is worth 0 ... +0.3 % on the driver's command at the same rule — the thin launches
// are a few per cent of a pass either way — and what an eager rule could win, +4.2 % if handed-over rays cost nothing
// (profiles/r06_handover.txt, the `drop` rows), the search's cost per ray takes back: rays a wave still holds when its queue is dry
// are many and short, the search pays for the long ones only.)
//   * Any-hit ray: occluded is occluded whatever the order.
//   * Closest-hit ray: the reference's result (BVH8Traversal.cuh:148-322) is what ITS visiting order finds, and that depends on
//     the order in exactly one situation: two triangles whose distances differ by less than the rounding of the slab test (the
//     traversal may or may not prune the box of the slightly closer one, depending on which it met first; at equal distances the
//     first met wins).  So the search keeps the two smallest distances it sees, accepting and pruning with a WINDOW above the
//     smallest (1e-3 of distance + coordinate magnitude: thousands of times the slab test's rounding).  A closest triangle with
//     nothing else inside its window is what every visiting order returns — no box on its path can have been pruned by a hit that
//     far behind it — with the t, u, v the ordinary loop computes (same frame, same arithmetic): that is the ray's record.
//     Anything else inside the window: the ray is traversed again from the root in the reference's own order (traverse_wave).
//   * A pool that would overflow ends the search; the ray is traversed in order likewise.
// Launches of the pass graph only (kTraceThinFlag), never the counting variant or the ray-batch hooks.
#ifndef NX_THIN_FACTOR
#define NX_THIN_FACTOR 4
#endif
constexpr int kThinFactor = NX_THIN_FACTOR;  // (the other two numbers of the rule travel in the device state: DeviceState::thinLanes / thinIters)
#ifdef NX_NO_THIN_CODE
constexpr bool kThinCode = false;  // (measurement: the kernel without the search's text)
#else
constexpr bool kThinCode = true;
#endif
#ifndef NX_THIN_FRAME_CACHE
#define NX_THIN_FRAME_CACHE 1
#endif
#ifndef NX_POOL_SLOTS
#define NX_POOL_SLOTS 1024
#endif
constexpr int kPoolSlots = NX_POOL_SLOTS;  // work items per wave of the thin kernel (8 KiB of LDS + 4 KiB of gates; most rays stay below 150, one in a few hundred passes 450)
constexpr int kPoolLimit = kPoolSlots - kWave;
constexpr uint32_t kItemNode = 0u, kItemInst = 1u, kItemTri = 2u;  // item.y = kind << 30 | frame (instance record + 1, 0 = TLAS); item.x = index

struct ThinResult {
    float t, u, v;
    uint32_t tri, inst;
    int count;        // closest hit: 1 = a closest triangle was found (t, u, v, tri, inst), 0 = none; any hit: 1 = occluded
    float second;     // closest hit: the second smallest distance seen (3e38: none)
    float window;     // ... and how far above t a second one makes the result depend on the visiting order
    float gate;       // closest hit: the largest computed entry distance of the boxes on the found triangle's path from the root (see thin_wave_search)
    bool replaced;    // closest hit, continued ray: a triangle closer than the hit the ray arrived with was found
    bool complete;    // the whole tree was searched
    uint32_t rounds;  // rounds of the search
    int poolMax;      // most items the pool held
};

// Searches ray (o, d) with the whole wave.  `bound`: closest hit — the ray's current hit distance (triangles at t <= bound count);
// any hit — its tmax (a triangle at 0 < t < bound occludes).  `pool`: kPoolSlots LDS entries of the wave's own.
// Everything in the result is wave-uniform.
// `seed` (round 6): the state the ray was handed over with — the search then continues it: its first work items are the groups of the
// ray's stack and its two current groups instead of the TLAS root, its bound and (closest hit) its best triangle so far the ray's own.
template <bool ANY_HIT>
NXD ThinResult thin_wave_search(const DeviceState* __restrict__ S, lds_u64* const pool, lds_f32* const poolGate, const f3 o, const f3 d, const float bound, const bool sceneIdentity,
                                const NX_G ThinState* const seed = nullptr)
{
    GU4 tlasNodes = S->tlasNodes;
    const NX_G InstTrav* instTrav = S->instTrav;
    const int lane = threadIdx.x & (kWave - 1);
    const f3 idirW = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const uint32_t oct = ((d.x < 0.0f ? 1u : 0u) << 2) | ((d.y < 0.0f ? 1u : 0u) << 1) | (d.z < 0.0f ? 1u : 0u);
    const uint32_t invOct4 = (7u - oct) * 0x01010101u;
    ThinResult r;
    r.t = bound; r.u = 0.0f; r.v = 0.0f; r.tri = 0xffffffffu; r.inst = 0xffffffffu; r.count = 0; r.second = 3.0e38f; r.complete = true;
    const float magnitude = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
    r.window = ANY_HIT ? 0.0f : 1.0e-3f * (fminf(bound, 1.0e30f) + magnitude);
    r.rounds = 0u; r.poolMax = 1; r.gate = 0.0f; r.replaced = false;
    int n = 1;  // items in the pool (uniform)
    if (seed == nullptr) {
        if (lane == 0) { pool[0] = (unsigned long long)kItemNode << 62; if (!ANY_HIT) poolGate[0] = 0.0f; }  // the TLAS root
    } else {
        // The ray's remaining work, as the loop would have taken it up: stack entry k (lane k), the current node group (lane sp) and the
        // current primitive group (lane sp + 1).  A node group's hit bits (24-31, octant-permuted) name inner children by slot; a
        // primitive group's bits (0-23) name primitives from its base: instances in the TLAS frame, triangles in an instance's.
        // Entries below instSp were pushed before the ray entered its instance: TLAS frame.  Every seed's gate is 0: the boxes above it
        // were entered by the ray itself — the loop never tests them again either (a popped group is visited unconditionally).
        const uint4 cur = *(const NX_G uint4*)&seed->ng;
        const uint4 hh = *(const NX_G uint4*)&seed->hitInst;  // hitInst, sp, instSp, leaf
        const int sp = min(max((int)hh.y, 0), kLdsDepth + kSpillDepth), instSp = (int)hh.z;
        const uint32_t inFrame = instSp >= 0 ? hh.w + 1u : 0u;
        uint2 e = make_uint2(0u, 0u);
        bool nodeGroup = false;
        uint32_t frame = inFrame;
        if (lane < sp) {
            e = seed->stack[lane];
            nodeGroup = (e.y & 0xff000000u) != 0u;
            if (instSp < 0 || lane < instSp) frame = 0u;
        } else if (lane == sp) { e = make_uint2(cur.x, cur.y); nodeGroup = true; }
        else if (lane == sp + 1) e = make_uint2(cur.z, cur.w);
        uint32_t bits = nodeGroup ? (e.y >> 24) : (e.y & 0x00ffffffu);
        const int mine = __popc(bits);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        n = __builtin_amdgcn_readlane(incl, kWave - 1);  // at most 34 groups x 24 < kPoolLimit
        int w = incl - mine;
        const uint32_t imask = e.y & 0xffu;
        while (bits) {
            const int b = 31 - __clz((int)bits);
            bits &= ~(1u << b);
            unsigned long long item;
            if (nodeGroup) {
                const int slot = b ^ (int)(invOct4 & 7u);
                const uint32_t rel = (uint32_t)__popc(imask & ~(0xffffffffu << slot));
                item = ((unsigned long long)((kItemNode << 30) | frame) << 32) | (unsigned long long)(e.x + rel);
            } else {
                const uint32_t tag = frame == 0u ? (kItemInst << 30) : ((kItemTri << 30) | frame);
                item = ((unsigned long long)tag << 32) | (unsigned long long)(e.x + (uint32_t)b);
            }
            pool[w] = item;
            if (!ANY_HIT) poolGate[w] = 0.0f;
            w++;
        }
        r.poolMax = n;
        if (!ANY_HIT) {
            const uint4 hit = *(const NX_G uint4*)&seed->hitT;
            r.t = __uint_as_float(hit.x);  // (the caller passes it as `bound` too)
            if (hit.w != 0xffffffffu) { r.count = 1; r.u = __uint_as_float(hit.y); r.v = __uint_as_float(hit.z); r.tri = hit.w; r.inst = hh.x; }
        }
    }
    // the frame the wave derived last (see below; uniform): 0 = none yet
    uint32_t cFrame = 0u, cInst = 0u;
    GU4 cNodes = tlasNodes;
    GF4 cIsect = nullptr;
    f3 cO = o, cD = d, cI = idirW;
    uint32_t rounds = 0u;
    bool careful = false;
    // (the product's limit, or the lower one a test sets to drive the put-back path: nxhip_debug_set_thin_pool)
    const int poolLimit = S->thinPoolLimit ? min(kPoolLimit, max(kWave, (int)S->thinPoolLimit)) : kPoolLimit;
    while (n > 0) {
        if (++rounds > (1u << 16)) { r.complete = false; break; }  // (a tree that is not a tree: the ordinary loop's stall guard deals with it)
        // (after a round whose children did not all fit — below — only as many items as can expand whatever they hold: 24 each)
        const int take = careful ? min(min(n, kWave), max(1, (poolLimit - n) / 24)) : min(n, kWave);
        const bool have = lane < take;
        const int at = n - 1 - lane;
        const unsigned long long item = have ? pool[at] : 0ull;
        const float itemGate = (!ANY_HIT && have) ? poolGate[at] : 0.0f;  // (closest hit: see the push below)
        n -= take;
        const uint32_t idx = (uint32_t)item, tag = (uint32_t)(item >> 32);
        const uint32_t kind = tag >> 30;
        uint32_t frame = tag & 0x3fffffffu;  // instance record + 1 whose frame the item lives in
        if (have && kind == kItemInst) frame = idx + 1u;
        // the ray in the item's frame (BVH8Traversal.cuh:259-264, as enter_instance computes it).  It is ONE ray, and nearly all items
        // of a round live in one instance: the wave remembers the last frame it derived — BLAS arrays, instance word, the ray in that
        // frame and its reciprocal direction, all wave-uniform — and an item of that frame takes them from there instead of fetching
        // the instance record first (a dependent round trip in front of the record fetch of EVERY round) and transforming the ray
        // again (same values: a function of the record and the ray alone).
        f3 ro = o, rd = d, ri = idirW;
        GU4 nodes = tlasNodes;
        GF4 isect = nullptr;
        uint32_t instIdx = 0u;
        const bool inFrame = have && frame != 0u;
        const bool known = NX_THIN_FRAME_CACHE && inFrame && frame == cFrame;  // (NX_THIN_FRAME_CACHE=0: measurement, every item derives its frame)
        if (known) { nodes = cNodes; isect = cIsect; instIdx = cInst; ro = cO; rd = cD; ri = cI; }
        const bool derive = inFrame && !known;
        if (derive) {
            const unsigned long long recAddr = (unsigned long long)&instTrav[frame - 1u];
            InstFetch fi;
            fetch_instance(recAddr, sceneIdentity, fi);
            nodes = fi.nodes();
            isect = fi.isect();
            instIdx = fi.instIdx;
            f3 o2, d2;
            if (enter_instance(fi, sceneIdentity, o, d, o2, d2)) {
                ro = o2;
                rd = d2;
                ri = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
            }
        }
        const unsigned long long deriveMask = __ballot(derive);
        if (deriveMask != 0ull) {  // remember the first such lane's frame
            const int src = __ffsll((long long)deriveMask) - 1;
            const auto bcast = [&](const uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); };
            const auto bcastf = [&](const float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), src)); };
            cFrame = bcast(frame);
            const unsigned long long nb = (unsigned long long)nodes, ib = (unsigned long long)isect;
            cNodes = (GU4)(((unsigned long long)bcast((uint32_t)(nb >> 32)) << 32) | bcast((uint32_t)nb));
            cIsect = (GF4)(((unsigned long long)bcast((uint32_t)(ib >> 32)) << 32) | bcast((uint32_t)ib));
            cInst = bcast(instIdx);
            cO = mk3(bcastf(ro.x), bcastf(ro.y), bcastf(ro.z));
            cD = mk3(bcastf(rd.x), bcastf(rd.y), bcastf(rd.z));
            cI = mk3(bcastf(ri.x), bcastf(ri.y), bcastf(ri.z));
        }
        unsigned long long recAddr = 0ull;
        if (have) {
            if (kind == kItemNode) recAddr = (unsigned long long)(nodes + (size_t)idx * (unsigned)kNodeStride);
            else if (kind == kItemInst) recAddr = (unsigned long long)&instTrav[idx] + 80ull;  // the copy of the BLAS's root node
            else recAddr = (unsigned long long)(isect + (size_t)idx * (unsigned)kTriStride);
        }
        const bool isNode = have && kind != kItemTri, isTri = have && kind == kItemTri;
        uint4 rc[5];
        fetch_record(isNode, isTri, recAddr, rc);
        uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
        float ct = 3.0e38f, cu = 0.0f, cv = 0.0f;  // this lane's candidate of the round
        bool cand = false;
        // closest hit: the children one by one with their computed entry distances (child_trace_gate); any hit: the loop's two masks
        uint32_t entered = 0u;
        float tminC[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (isNode) {
            if constexpr (ANY_HIT) child_trace(rc, ro, rd, ri, invOct4, r.t + r.window, ng, tg);
            else child_trace_gate(rc, ro, rd, ri, r.t + r.window, entered, tminC);
        }
        if (isTri) {
            const f3 p0 = mk3(__uint_as_float(rc[0].x), __uint_as_float(rc[0].y), __uint_as_float(rc[0].z));
            const f3 edge0 = mk3(__uint_as_float(rc[1].x), __uint_as_float(rc[1].y), __uint_as_float(rc[1].z));
            const f3 edge1 = mk3(__uint_as_float(rc[2].x), __uint_as_float(rc[2].y), __uint_as_float(rc[2].z));
            const f3 rayCrossEdge1 = cross3(rd, edge1);
            const float det = dot3(edge0, rayCrossEdge1);
            const float invDet = 1.0f / det;
            const f3 sv = ro - p0;
            const float u = invDet * dot3(sv, rayCrossEdge1);
            const f3 sCrossEdge0 = cross3(sv, edge0);
            const float v = invDet * dot3(rd, sCrossEdge0);
            const float t = invDet * dot3(edge1, sCrossEdge0);
            // (closest hit: "at or below" the distance found so far, so that a second triangle at the same distance is noticed)
            cand = !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) && t > 0.0f && (ANY_HIT ? t < r.t : t <= r.t + r.window);
            if (cand) { ct = t; cu = u; cv = v; }
        }
        // the children the ray enters go back into the pool
        const bool inTlas = isNode && kind == kItemNode && frame == 0u;
        int mine = isNode ? __popc(ng.y & 0xff000000u) + __popc(tg.y) : 0;
        if constexpr (!ANY_HIT) {
            // a child's items: one for an inner node, one per primitive of a leaf (the unary count above bit 5 of its meta byte)
            mine = 0;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const uint32_t meta = ((c < 4 ? rc[1].z : rc[1].w) >> (8 * (c & 3))) & 0xffu;
                mine += ((entered >> c) & 1u) ? __popc(meta >> 5) : 0;
            }
        }
        int incl = mine;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        int total = __builtin_amdgcn_readlane(incl, kWave - 1);
        // A round whose children do not all fit: the lanes up to the last one that fits push theirs, the others put their OWN item
        // back (one slot each, on top of those: the 64 slots above kPoolLimit are kept for that) and the search goes on more
        // carefully.  Only when not even one item can be expanded is the ray given to the in-order replay.
        const bool fits = n + incl <= poolLimit;  // (a prefix of the lanes: incl never decreases)
        const bool putBack = mine > 0 && !fits;
        const unsigned long long backMask = __ballot(putBack);
        if (backMask != 0ull) {
            const int fitting = (int)__popcll(__ballot(fits));
            total = fitting > 0 ? __builtin_amdgcn_readlane(incl, fitting - 1) : 0;
            if (putBack) {
                const int at2 = n + total + (int)__popcll(backMask & ((1ull << lane) - 1ull));
                pool[at2] = item;
                if (!ANY_HIT) poolGate[at2] = itemGate;
                mine = 0;
            }
            total += (int)__popcll(backMask);
            if (careful && take == 1) r.complete = false;  // (a single item that cannot expand into a pool this full)
            careful = true;
        }
        if (!ANY_HIT && mine) {
            // Closest hit: every item carries its GATE — the largest computed entry distance (child_trace's tmin, the number its hit test
            // compares with the ray's current hit distance) among the boxes on its path from the root.  What it is for: see the end
            // of thin_kernel's closest-hit branch.
            int w = n + incl - mine;
            const uint32_t imask = rc[0].w >> 24, baseChild = rc[1].x, basePrim = rc[1].y;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (((entered >> c) & 1u) == 0u) continue;
                const uint32_t meta = ((c < 4 ? rc[1].z : rc[1].w) >> (8 * (c & 3))) & 0xffu;
                const float g = fmaxf(itemGate, tminC[c]);
                if ((meta & (meta << 1)) & 0x10u) {  // an inner node (child_trace: isInner4); its slot among the node's inner children
                    const uint32_t rel = (uint32_t)__popc(imask & ~(0xffffffffu << (meta & 7u)));
                    pool[w] = ((unsigned long long)((kItemNode << 30) | frame) << 32) | (unsigned long long)(baseChild + rel);
                    poolGate[w] = g;
                    w++;
                } else {
                    const uint32_t leafTag = inTlas ? (kItemInst << 30) : ((kItemTri << 30) | frame);
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        if (((meta >> (5 + k)) & 1u) == 0u) continue;
                        pool[w] = ((unsigned long long)leafTag << 32) | (unsigned long long)(basePrim + (meta & 0x1fu) + (uint32_t)k);
                        poolGate[w] = g;
                        w++;
                    }
                }
            }
        }
        if (ANY_HIT && mine) {
            int w = n + incl - mine;
            uint32_t inner = ng.y;
            while (inner & 0xff000000u) {
                const int nodeOffset = 31 - __clz((int)inner);
                inner &= ~(1u << nodeOffset);
                const int slot = (nodeOffset - 24) ^ (int)(invOct4 & 7u);
                const int rel = __popc(inner & ~(0xffffffffu << slot));
                pool[w] = ((unsigned long long)((kItemNode << 30) | frame) << 32) | (unsigned long long)(ng.x + (uint32_t)rel);
                w++;
            }
            uint32_t leaves = tg.y;
            while (leaves) {
                const int off = 31 - __clz((int)leaves);
                leaves &= ~(1u << off);
                const uint32_t leafTag = inTlas ? (kItemInst << 30) : ((kItemTri << 30) | frame);
                pool[w] = ((unsigned long long)leafTag << 32) | (unsigned long long)(tg.x + (uint32_t)off);
                w++;
            }
        }
        if (r.complete) n += total;
        r.poolMax = max(r.poolMax, n);
        // the round's closest candidate
        const unsigned long long anyCand = __ballot(cand);
        if (anyCand != 0ull) {
            if (ANY_HIT) { r.count = 1; break; }
            float m = ct;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off));
            const unsigned long long atMin = __ballot(cand && ct == m);
            const int winner = __ffsll((long long)atMin) - 1;
            // the round's second smallest: the same distance again if two lanes hold it, else the smallest of the others
            float m2 = (cand && lane != winner) ? ct : 3.0e38f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m2 = fminf(m2, __shfl_xor(m2, off));
            if (r.count == 0 || m < r.t) {
                r.replaced = true;
                r.second = r.count ? fminf(r.t, m2) : m2;
                r.t = m;
                r.u = __shfl(cu, winner);
                r.v = __shfl(cv, winner);
                r.tri = __shfl(rc[0].w, winner);
                r.inst = __shfl(instIdx, winner);
                r.gate = __shfl(itemGate, winner);
                r.count = 1;
                r.window = 1.0e-3f * (m + magnitude);
            } else {
                r.second = fminf(r.second, m);
            }
        }
        if (!r.complete) break;
    }
    r.rounds = rounds;
    return r;
}

#ifdef NX_WAVE_TIMELINE
constexpr int kTimelineBounces = 12, kTimelineWaves = 8192;
__device__ unsigned long long g_waveTimeline[2][kTimelineBounces][kTimelineWaves][5];
#endif

template <bool ANY_HIT, bool STATS>
// 5 waves per SIMD for both variants (96 VGPRs, no spills in the loop).  Before an instance entry also carried its BLAS
// root (17 more live registers in the fetch), 6 waves at 80 VGPRs was the best point (5: -3 %, 7: -0.3 %, 8: -1.5 %); with it,
// 6 waves spill 23 VGPRs inside the loop (-10 %), 5 and 4 measure +4.5 % and +1 % over the old kernel at 6.
#ifndef NX_WAVES_PER_EU
#define NX_WAVES_PER_EU 5
#endif
__attribute__((amdgpu_waves_per_eu(NX_WAVES_PER_EU, NX_WAVES_PER_EU)))
__global__ void __launch_bounds__(kTraceBlock) trace_kernel(const DeviceState* __restrict__ S, const int bounceArg)
{
    // bounce | kTraceScanFlag: a launch of the SCAN pipeline — its rays are the set of the bounce's parity, and the closest-hit
    // record says what the path does next (see the flush below); without the flag: rays[0] and plain hit records
    const int bounce = bounceArg & 0xff;
    const bool scan = !ANY_HIT && (bounceArg & kTraceScanFlag) != 0;
    // ... | kTraceEntryFlag: the rays name the entry state of their run (rayO.w): installed at refill instead of the root's
    const bool entryLaunch = !ANY_HIT && (bounceArg & kTraceEntryFlag) != 0 && S->entry != nullptr;
    // ... | kTraceThinFlag: the last long rays of a dry wave may be handed to the thin kernel (below)
    bool thinAllowed = !STATS && kThinCode && (bounceArg & kTraceThinFlag) != 0;
    const int thinLanes = (int)(S->thinLanes & 0xffu);
    // (test hook, nxhip_debug_set_thin inHooks bit 1: hand over after thinIters iterations of EVERY stretch between two refill points, dry
    //  queue or not — rays then arrive at the thin kernel with the state of exactly that many steps)
    const bool thinAnyTime = (S->thinLanes >> 31) != 0u;
    const uint32_t thinIters = S->thinIters;
    uint32_t itersTotal = 0u, taken = 0u, thinAfter = thinIters;  // (wave-uniform: loop iterations and rays of this wave so far)
    const NX_G EntryState* const entryTable = S->entry;
    const int raySet = (bounceArg & kTraceScanFlag) ? (bounce & 1) : 0;
    const bool requeue = S->debugRequeue != 0u;  // (nxhip_debug_set_requeue; scalar)
    __shared__ unsigned long long ldsStack[kLdsDepth * kTraceBlock];
    // the world-space ray (origin, direction, 1 / direction) of a lane that is inside a transformed instance: parked here on
    // entry and taken back on exit.  (Re-reading the ray from its queue on exit put a second, dependent memory round trip
    // and three IEEE divisions into every iteration in which any lane left an instance — most of them on instanced scenes.)
    __shared__ float ldsWorld[9 * kTraceBlock];

#if NX_WG_RANGE
    // The workgroup's block of rays (see the refill): {next ray, end} packed in one 64-bit word the four waves draw from with
    // LDS atomics; sLock: a wave is fetching the next block; sDry: no shard holds rays any more; sShard: the shard the blocks come from
    // (sBegin / sEnd: the eight shards' first ray and end, sReserve: the block size — kept here, not in every wave's registers: they
    //  are needed once per block)
    __shared__ unsigned long long sRange;
    __shared__ int sLock, sDry, sShard, sReserve;
    __shared__ int sBegin[kXcds], sEnd[kXcds];
#endif
    NX_G Counters* C = S->counters;
    // the queue's eight regions (nx_device.h): region k = slots [k * cap, k * cap + regionRays[k]), fetch head k counts the
    // rays handed out of it
    // (word k of either: [k * kRegionStride])
    const NX_G int32_t* regionRays = ANY_HIT ? &C->region[0].traceShadowSize[bounce] : &C->region[0].traceSize[bounce];
    const int cap = (int)S->queueShardCap;
    int size = 0;
#pragma unroll
    for (int k = 0; k < kXcds; k++) size += regionRays[k * kRegionStride];
    if (size <= 0) return;
    // The eight fetch heads walk one SHARD of the queue each: the queue's regions — or, when the queue is kept in one region
    // (ordered compaction: the reference's serial slot order), eight 64-aligned pieces of it, so that such a pass still runs on
    // all eight XCDs (until round 4 it ran on the one whose workgroups call region 0 home).
    const bool oneRegion = S->queueShards == 1u;
    const int piece = (int)dense_piece((uint32_t)size, (uint32_t)kXcds);
    const auto shard_begin = [&](const int k) { return oneRegion ? k * piece : k * cap; };
    const auto shard_rays = [&](const int k) { return oneRegion ? max(0, min(piece, size - k * piece)) : (int)regionRays[k * kRegionStride]; };
    NX_G int* heads = ANY_HIT ? &C->region[0].shadowHead[bounce] : &C->region[0].traceHead[bounce];
    GF4 rayO = ANY_HIT ? S->shadow.rayO : S->trace.rays[raySet].rayO;
    GF4 rayD = ANY_HIT ? S->shadow.rayD : S->trace.rays[raySet].rayD;
    GU4 tlasNodes = S->tlasNodes;
    const NX_G InstTrav* instTrav = S->instTrav;
#ifdef NX_NO_SCENE_FLAG
    const bool sceneIdentity = false;
#else
    const bool sceneIdentity = (S->sceneFlags & kSceneAllIdentity) != 0u;  // wave-uniform: no instance of the scene transforms a ray
#endif

    const int lane = threadIdx.x & (kWave - 1);
    const unsigned long long laneLt = (1ull << lane) - 1ull;
    const int homeShard = blockIdx.x & (kXcds - 1);
    const int homeRays = shard_rays(homeShard);
    // this wave's rank among the waves that call this shard home
    const int rankInShard = (int)(blockIdx.x >> 3) * (kTraceBlock / kWave) + (int)(threadIdx.x / kWave);
#ifdef NX_WAVE_TIMELINE
    // measurement variant: every wave's life in the launch (wall_clock64: 100 MHz, one clock for the chip), kept in a device array
    // and read with nxhip_debug_read_wave_timeline (bench.py NX_WAVE_TIMELINE_OUT)
    const unsigned long long wpStart = wall_clock64();
    unsigned long long wpDry = 0ull;
    int wpHanded = 0;
#endif
    // A wave the queue does not need leaves without touching a fetch head.  The grid is sized for the largest queue; on a
    // small one most waves would otherwise each walk all 8 heads with returning atomics to find out that nothing is left,
    // which made every launch cost about 0.5 ms however few rays it carried.
    // How many waves a queue needs.  Up to 64 K rays: one wave per 64 rays — every ray has a lane at once, the launch is as
    // long as its slowest ray, and at most one wave sits on each SIMD.  Larger queues first give those waves more rays
    // each (up to 256, refilled as lanes free up: a wave with a backlog keeps its lanes busy through what would be its
    // drain), then use more waves.  Against one wave per 64 rays throughout: the driver's 20-frame pass +4 % (its late
    // bounces carry 0.3-4 M rays), one-frame passes unchanged; 256 rays per wave throughout: one-frame passes -5 %.
#ifndef NX_RPW_SHIFT
#define NX_RPW_SHIFT 10
#endif
    const int raysPerWave = min(256, max(kWave, (size >> NX_RPW_SHIFT) & ~(kWave - 1)));
#if !NX_WG_RANGE
    if (rankInShard * raysPerWave >= homeRays) return;
#endif
    // Reservation size: kReserve rays, but no more than half a wave's even share of the queue, so that on a small queue
    // every wave draws a few times and the launch does not end with a handful of waves still holding full blocks
    // (one frame per pass: +14 %; 64 frames per pass: within noise).
    const int gridWaves = (int)(gridDim.x * (kTraceBlock / kWave));
    const int wavesAtWork = min(gridWaves, size / raysPerWave + kXcds);
    const int reserve = min(kReserve, max(kWave, (size / (wavesAtWork * 2)) & ~(kWave - 1)));
#if NX_WG_RANGE
    if (threadIdx.x < kXcds) {
        sBegin[threadIdx.x] = shard_begin((int)threadIdx.x);
        sEnd[threadIdx.x] = shard_begin((int)threadIdx.x) + shard_rays((int)threadIdx.x);
    }
    if (threadIdx.x == 0) { sRange = 0ull; sLock = 0; sDry = 0; sShard = homeShard; sReserve = reserve; }
    __syncthreads();  // (before any wave leaves: the exit below is per wave)
    if (rankInShard * raysPerWave >= homeRays) return;
#endif
    int shard = homeShard;
    bool exhausted = false;
    int rngCur = 0, rngEnd = 0;  // rays reserved by this wave and not handed to a lane yet

    lds_u64* const stackLds = (lds_u64*)&ldsStack[threadIdx.x];
    uint2 stackSpill[kSpillDepth];
    int sp = 0;

#ifdef NX_EXTRA_VALU
    float dummy0 = (float)lane, dummy1 = dummy0 + 1.0f, dummy2 = dummy0 + 2.0f, dummy3 = dummy0 + 3.0f;
#endif
    bool active = false;
    bool resultPending = false;  // this lane's ray has finished and its result has not been written yet
    f3 org = mk3(0.0f), dir = mk3(0.0f), idir = mk3(0.0f);
    float hitT = 0.0f, hitU = 0.0f, hitV = 0.0f;
    uint32_t hitTri = 0xffffffffu, hitInst = 0xffffffffu;
    uint32_t rayIdx = 0, pixelBits = 0, instIdx = 0, invOct4 = 0;
    int instSp = -1;
    bool xformed = false;  // the current instance's inverse transform is not the identity: the world ray must be restored on exit
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    GU4 nodes = tlasNodes;
    GF4 isect = nullptr;
    unsigned long long nNodes = 0, nTris = 0, nInst = 0, nRays = 0;
    unsigned long long wIters = 0, wActive = 0, wNode = 0, wPrim = 0;  // lane 0 only (STATS)
    unsigned long long cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tPrev = STATS ? (unsigned long long)clock64() : 0ull;
#define NX_STAMP(k) do { if (STATS) { const unsigned long long tNow = (unsigned long long)clock64(); cyc[k] += tNow - tPrev; tPrev = tNow; } } while (0)

    for (;;) {
        // ---- flush: write the results of the rays that finished since the last refill point.  Done here, outside the
        //      traversal loop, because a retirement inside it made EVERY iteration pay for a handful of lanes: the store
        //      sequence (closest hit) or, worse, a dependent load -> add -> store of the pixel's radiance with its memory
        //      round trip (any hit), executed whenever at least one of the 64 lanes retired — nearly always.
        if (resultPending) {
            resultPending = false;
            if (ANY_HIT) {
                // unoccluded: pathRadiance[pixelIdx] += radiance (BVH8Traversal.cuh:515-516);
                // at most one shadow ray per pixel and bounce, so no atomic is needed
                const float4 r = S->shadow.radiance[rayIdx];
                NX_G float4* dst = &S->radiance[pixelBits];
                float4 cur = *dst;
                cur.x += r.x; cur.y += r.y; cur.z += r.z;
                *dst = cur;
            } else {
                // The hit's instance arrives with its material type + 1 above kHitCodeShift (InstTrav::instIdx), the ray with the
                // Russian-roulette draw of the next logic step in bit 31 of rayIdx (kRaySurvives, made by the ray's producer).  SCAN
                // pipeline: the record's instance word then IS the logic step's decision (PathTracer.cu:136-210) — a miss, a path
                // the roulette ends (0), or the material kernel that shades the hit — and no logic kernel runs; otherwise the plain
                // instance index.
                const uint32_t slot = rayIdx & 0x7fffffffu;
                const bool missed = hitTri == 0xffffffffu;
                S->trace.hit[slot] = make_float4(hitT, hitU, hitV, __uint_as_float(hitTri));
                S->trace.hitInst[slot] = scan ? (missed ? (kHitCodeMiss << kHitCodeShift) : ((rayIdx >> 31) ? hitInst : 0u))
                                              : (missed ? 0xffffffffu : (hitInst & kHitInstMask));
            }
        }
        // ---- refill idle lanes from the wave's reserved range; one atomic reserves kReserve rays of a shard at a time
        //      (a returning atomic on a contended head costs microseconds during which the whole wave stalls, so it
        //      must not be paid per refill)
        if (!exhausted || rngCur < rngEnd) {
            bool need = !active;
            for (;;) {
                const unsigned long long needMask = __ballot(need);
                if (needMask == 0ull) break;
                if (rngCur >= rngEnd) {
                    if (exhausted) break;
                    const int leader = __ffsll((long long)needMask) - 1;
#if NX_WG_RANGE
                    // The rays a wave hands its idle lanes come out of the WORKGROUP's block: one LDS atomic takes as many as the wave
                    // needs now.  A block is what one returning atomic on the shard's head reserves (below) — as before, but for the
                    // four waves together, so that the last block of a launch is finished by four waves side by side instead of by
                    // the one that happened to draw it while the others leave (wave by wave, a launch's last 200-450 us ran on 2 % of
                    // the chip: tools/wave_timeline.py).  (readfirstlane: what comes out of LDS is the same in every lane, and the
                    // compiler must know it — the range, `exhausted` and the loop's conditions stay scalar.)
                    {
                        const int want = (int)__popcll(needMask);
                        unsigned long long old = 0ull;
                        if (lane == leader) old = __hip_atomic_fetch_add(&sRange, (unsigned long long)want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const int cur = __builtin_amdgcn_readlane((int)(uint32_t)old, leader);
                        const int end = __builtin_amdgcn_readlane((int)(uint32_t)(old >> 32), leader);
                        if (cur < end) {
                            rngCur = cur;
                            rngEnd = min(end, cur + want);
                        }
                    }
                    if (rngCur >= rngEnd)
#endif
                    {
#if NX_WG_RANGE
                        // the workgroup's block is used up
                        if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&sDry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0) {
                            exhausted = true;
#ifdef NX_WAVE_TIMELINE
                            wpDry = wall_clock64();
#endif
                            thinAfter = thinIters ? max(thinIters, (uint32_t)((float)kThinFactor * 40.0f * (float)itersTotal / (float)max(taken, 1u))) : 0u;
                            break;
                        }
                        int lock = 1;
                        if (lane == leader) lock = __hip_atomic_exchange(&sLock, 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        lock = __builtin_amdgcn_readlane(lock, leader);
                        if (lock != 0) {
                            // another wave of the workgroup is fetching the next block: go on with the lanes that have rays (back here
                            // after the next iteration); a wave without any waits
                            if (__ballot(active) != 0ull) break;
                            __builtin_amdgcn_s_sleep(4);
                            continue;
                        }
                        // this wave fetches (unless a block was published since it looked)
                        const unsigned long long now = __hip_atomic_load(&sRange, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (__builtin_amdgcn_readfirstlane((int)((uint32_t)now < (uint32_t)(now >> 32))) != 0) {
                            if (lane == leader) __hip_atomic_store(&sLock, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            continue;
                        }
                        shard = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sShard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                        const int shardBegin = __builtin_amdgcn_readfirstlane(sBegin[shard]), shardEnd = __builtin_amdgcn_readfirstlane(sEnd[shard]);
                        const int blockRays = __builtin_amdgcn_readfirstlane(sReserve);
#else
                        const int shardBegin = shard_begin(shard);
                        const int shardEnd = shardBegin + shard_rays(shard);
                        const int blockRays = reserve;
#endif
                        // one returning atomic reserves a block of rays of `shard` (the head counts rays handed out)
                        int base = 0;
                        if (lane == leader) base = atomicAdd(&heads[shard * kRegionStride], blockRays);
                        base = __builtin_amdgcn_readlane(base, leader);
                        rngCur = shardBegin + base;
                        rngEnd = min(shardEnd, rngCur + blockRays);
#if NX_WG_RANGE
                        if (rngCur < shardEnd) {  // the workgroup's new block
                            if (lane == leader) {
                                __hip_atomic_store(&sRange, ((unsigned long long)(uint32_t)rngEnd << 32) | (unsigned long long)(uint32_t)rngCur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_store(&sLock, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                            rngCur = rngEnd = 0;
                            continue;
                        }
#endif
                        if (rngCur >= shardEnd) {
                            // this shard is dry: one load of all 8 heads tells which shards still hold rays (a load is served
                            // in parallel with other waves', returning atomics on a head are serialised); go to the fullest
                            rngCur = rngEnd = 0;
                            int left = 0;
                            if (lane < kXcds) {
                                const int taken = __hip_atomic_load(&heads[lane * kRegionStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if NX_WG_RANGE
                                left = max(0, sEnd[lane] - sBegin[lane] - taken);
#else
                                left = max(0, shard_rays(lane) - taken);
#endif
                            }
                            int best = 0, bestLeft = 0;
#pragma unroll
                            for (int k = 0; k < kXcds; k++) {
                                const int l = __shfl(left, k);
                                if (l > bestLeft) { bestLeft = l; best = k; }
                            }
                            best = __builtin_amdgcn_readfirstlane(best);
                            bestLeft = __builtin_amdgcn_readfirstlane(bestLeft);
                            if (bestLeft <= 0) {
                                exhausted = true;
#ifdef NX_WAVE_TIMELINE
                                wpDry = wall_clock64();
#endif
                                // how long "long" is for this launch: kThinFactor times what a ray of this wave took on average (a wave
                                // iteration advances its busy lanes — about 40 of 64 — by one record each), at least thinIters
                                // (thinIters 0 — a test hook — hands a wave's rays over after their first iteration, dry queue or not)
                                thinAfter = thinIters ? max(thinIters, (uint32_t)((float)kThinFactor * 40.0f * (float)itersTotal / (float)max(taken, 1u))) : 0u;
                            }
                            else shard = best;
#if NX_WG_RANGE
                            // (the workgroup's state: no rays anywhere, or the shard its next block comes from; the other waves find out at
                            //  their next look)
                            if (lane == leader) {
                                if (bestLeft <= 0) __hip_atomic_store(&sDry, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                else __hip_atomic_store(&sShard, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_store(&sLock, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
#endif
                            continue;
                        }
                    }
                }
                const int avail = rngEnd - rngCur;
                const int rank = __popcll(needMask & laneLt);
                if (need && rank < avail) {
                    const int idx = rngCur + rank;
                    need = false;
                    active = true;
                    const float4 o = rayO[idx], d = rayD[idx];
                    // (closest hit: the producer's roulette draw rides in the top bit, see the flush; a queue holds < 2^31 rays)
                    rayIdx = ANY_HIT ? (uint32_t)idx : ((uint32_t)idx | ((__float_as_uint(o.w) & kRaySurvives) << 30));
                    org = mk3(o.x, o.y, o.z);
                    dir = mk3(d.x, d.y, d.z);
                    pixelBits = __float_as_uint(d.w);
                    idir = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
                    hitT = ANY_HIT ? o.w : 1e30f;
                    hitU = 0.0f; hitV = 0.0f; hitTri = 0xffffffffu; hitInst = 0xffffffffu;
                    const uint32_t oct = ((dir.x < 0.0f ? 1u : 0u) << 2) | ((dir.y < 0.0f ? 1u : 0u) << 1) | (dir.z < 0.0f ? 1u : 0u);
                    invOct4 = (7u - oct) * 0x01010101u;
                    ng = make_uint2(0u, 0x80000000u);
                    tg = make_uint2(0u, 0u);
                    sp = 0;
                    instSp = -1;
                    xformed = false;
                    nodes = tlasNodes;
                    if (STATS) nRays++;
                    if (!ANY_HIT && entryLaunch) {
                        // the state the first node steps of this ray's run provably lead to (nx_entry.hip), instead of the root's
                        const uint32_t en = __float_as_uint(o.w) >> kRayEntryShift;
                        if (en != 0u) {
                            const NX_G EntryState* es = entryTable + (en - 1u);
                            const int4 hdr = *(const NX_G int4*)&es->sp;  // sp, instSp, leafSlot, steps
                            if (hdr.w > 0) {
                                const uint4 g = *(const NX_G uint4*)&es->ng;
                                ng = make_uint2(g.x, g.y);
                                tg = make_uint2(g.z, g.w);
#pragma unroll
                                for (int k = 0; k < kEntryMaxStack; k += 2) {
                                    const uint4 sk = *(const NX_G uint4*)&es->stack[k];
                                    if (k < hdr.x) stack_push(stackLds, stackSpill, sp, make_uint2(sk.x, sk.y));
                                    if (k + 1 < hdr.x) stack_push(stackLds, stackSpill, sp, make_uint2(sk.z, sk.w));
                                }
                                instSp = hdr.y;
                                if (hdr.z >= 0) {  // inside an (identity) instance: its BLAS arrays and index
                                    const NX_G InstTrav* rec = instTrav + hdr.z;
                                    nodes = rec->nodes;
                                    isect = rec->isect;
                                    instIdx = rec->instIdx;
                                }
                            }
                        }
                    }
                }
                // (the test hook — a wave-uniform scalar — leaves the range where it is: the same rays go out again at the next refill)
                rngCur += requeue ? 0 : min(__popcll(needMask), avail);
                taken += (uint32_t)min(__popcll(needMask), avail);
#ifndef NX_NO_STALL_GUARD
                // Progress per LAUNCH: a wave cannot be handed more rays than the queue holds.  One that is holds rays that come back into
                // the queue — every one of them retires, so the iteration guard below, which counts between two refill points, never
                // fires.  Same ending: the rays in hand are abandoned, the wave takes no more, the host gets NXHIP_ERR_TRAVERSAL.
                if (taken > (uint32_t)size) {
                    if (lane == 0) atomicOr(&S->frame->errorWord, kErrRaysRetaken);
                    active = false;  // (the rays in hand end without a record of this visit: the launch's results are void, the status says so)
                    resultPending = false;
                    need = false;
                    exhausted = true;
                    rngCur = rngEnd = 0;
                    break;
                }
#endif
            }
        }
        unsigned long long activeMask = __ballot(active);
        NX_STAMP(0);
        if (activeMask == 0ull) break;
        uint32_t spins = 0u;  // iterations since the wave last came through its refill point (wave-uniform: a scalar register)

        // ---- traverse until too many lanes have run out of work
        do {
            if (STATS) {
                wIters++;
                wActive += __popcll(__ballot(active));
            }
            // A: acquire work from the stack, or retire the ray
            if (active && tg.y == 0u && (ng.y & 0xff000000u) == 0u) {
                if (sp == 0) {
                    // the ray is finished; its result stays in the lane's registers and is written out at the next refill
                    // point (see "flush"), for all lanes that finished since the last one together
                    active = false;
                    resultPending = true;
                } else {
                    if (sp == instSp) {  // leaving the instance: back to the world-space ray and the TLAS
                        if (xformed) {
                            const float* w = &ldsWorld[threadIdx.x];
                            org = mk3(w[0 * kTraceBlock], w[1 * kTraceBlock], w[2 * kTraceBlock]);
                            dir = mk3(w[3 * kTraceBlock], w[4 * kTraceBlock], w[5 * kTraceBlock]);
                            idir = mk3(w[6 * kTraceBlock], w[7 * kTraceBlock], w[8 * kTraceBlock]);
                        }
                        nodes = tlasNodes;
                        instSp = -1;
                    }
                    const uint2 e = stack_pop(stackLds, stackSpill, sp);
                    if (e.y & 0xff000000u) ng = e;
                    else { tg = e; ng = make_uint2(0u, 0u); }
                }
            }
            NX_STAMP(1);
            // Every busy lane now needs exactly one record: a node (its node group has unvisited children and no leaf
            // work is pending), an instance record (pending TLAS leaf) or a triangle record (pending BLAS leaf).
            const bool wantNode = active && tg.y == 0u && (ng.y & 0xff000000u) != 0u;
            const bool wantInst = active && tg.y != 0u && instSp < 0;
            const bool wantTri = active && tg.y != 0u && instSp >= 0;
            if (STATS) {
                wNode += __popcll(__ballot(wantNode || wantInst));  // an instance entry also tests the BLAS root (below)
                wPrim += __popcll(__ballot(wantInst || wantTri));
            }
            unsigned long long recAddr = 0ull;
            if (wantNode) {
                const int nodeOffset = 31 - __clz((int)ng.y);
                ng.y &= ~(1u << nodeOffset);
                if (ng.y & 0xff000000u) stack_push(stackLds, stackSpill, sp, ng);
                const int slot = (nodeOffset - 24) ^ (int)(invOct4 & 7u);
                const int rel = __popc(ng.y & ~(0xffffffffu << slot));
                recAddr = (unsigned long long)(nodes + (size_t)(ng.x + (uint32_t)rel) * (unsigned)kNodeStride);
            } else if (wantInst) {
                const int off = 31 - __clz((int)tg.y);
                tg.y &= ~(1u << off);
                recAddr = (unsigned long long)&instTrav[tg.x + (uint32_t)off];
                if (tg.y) stack_push(stackLds, stackSpill, sp, tg);
                if (ng.y & 0xff000000u) stack_push(stackLds, stackSpill, sp, ng);
                instSp = sp;
            } else if (wantTri) {
                const int off = 31 - __clz((int)tg.y);
                tg.y &= ~(1u << off);
                recAddr = (unsigned long long)(isect + (size_t)(tg.x + (uint32_t)off) * (unsigned)kTriStride);
            }
            // An instance record carries a copy of its BLAS's root node behind the transform: the lane fetches both, enters the
            // instance and tests the root in this same iteration (the reference's next step for that ray, BVH8Traversal.cuh:
            // 259-266 then :180 — no other record of the ray lies between them), which saves one iteration per instance
            // visit and puts the lane into the node block that runs for the other lanes anyway.
            uint4 rc[5];
            InstFetch fi;
            if (wantInst) fetch_instance(recAddr, sceneIdentity, fi);
            fetch_record(wantNode || wantInst, wantTri, recAddr + (wantInst ? 80ull : 0ull), rc);
            NX_STAMP(2);
            if (wantInst) {
                nodes = fi.nodes();
                isect = fi.isect();
                instIdx = fi.instIdx;
                // the octant order keeps using the world-space direction (BVH8Traversal.cuh:259-264).
                // A transform that maps this ray onto itself bit for bit leaves 1/dir as it is, so the three divisions here and
                // the reload + three divisions on exit are skipped; in a scene of identity instances only the transform is not
                // even computed (enter_instance).
                f3 o2, d2;
                xformed = enter_instance(fi, sceneIdentity, org, dir, o2, d2);
                if (xformed) {
                    float* w = &ldsWorld[threadIdx.x];
                    w[0 * kTraceBlock] = org.x; w[1 * kTraceBlock] = org.y; w[2 * kTraceBlock] = org.z;
                    w[3 * kTraceBlock] = dir.x; w[4 * kTraceBlock] = dir.y; w[5 * kTraceBlock] = dir.z;
                    w[6 * kTraceBlock] = idir.x; w[7 * kTraceBlock] = idir.y; w[8 * kTraceBlock] = idir.z;
                    org = o2;
                    dir = d2;
                    idir = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
                }
                if (STATS) nInst++;
            }
            NX_STAMP(4);
            if (wantNode || wantInst) {
                child_trace(rc, org, dir, idir, invOct4, hitT, ng, tg);
                if (STATS) nNodes++;
            }
            NX_STAMP(3);
            if (wantTri) {
                // Moeller-Trumbore on the leaf-ordered stream — Triangle.cuh:53-86 / :89-118
                const f3 p0 = mk3(__uint_as_float(rc[0].x), __uint_as_float(rc[0].y), __uint_as_float(rc[0].z));
                const f3 edge0 = mk3(__uint_as_float(rc[1].x), __uint_as_float(rc[1].y), __uint_as_float(rc[1].z));
                const f3 edge1 = mk3(__uint_as_float(rc[2].x), __uint_as_float(rc[2].y), __uint_as_float(rc[2].z));
                const f3 rayCrossEdge1 = cross3(dir, edge1);
                const float det = dot3(edge0, rayCrossEdge1);
                const float invDet = 1.0f / det;
                const f3 s = org - p0;
                const float u = invDet * dot3(s, rayCrossEdge1);
                const f3 sCrossEdge0 = cross3(s, edge0);
                const float v = invDet * dot3(dir, sCrossEdge0);
                const float t = invDet * dot3(edge1, sCrossEdge0);
                const bool hit = !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) && (t > 0.0f && t < hitT);
                if (STATS) nTris++;
                if (hit) {
                    if (ANY_HIT) {
                        active = false;  // occluded: nothing to add
                    } else {
                        hitT = t; hitU = u; hitV = v;
                        hitTri = rc[0].w;
                        hitInst = instIdx;
                    }
                }
            }
            NX_STAMP(5);
#ifndef NX_NO_EARLY_RETIRE
            // a ray that has nothing left (no pending child, no pending leaf, empty stack) retires now instead of spending the
            // next iteration's acquire step on finding that out: its lane counts as free one iteration earlier
            if (active && tg.y == 0u && (ng.y & 0xff000000u) == 0u && sp == 0) {
                active = false;
                resultPending = true;
            }
#endif
#ifdef NX_EXTRA_VALU
            // experiment: how does the kernel's rate respond to its VALU instruction count?  NX_EXTRA_VALU
            // independent FMAs per iteration that change no result: +64 on the ~330 of an iteration costs 15 % of the kernel's time
#pragma unroll
            for (int k = 0; k < NX_EXTRA_VALU; k += 4) {
                dummy0 = fmaf(dummy0, 1.0001f, 0.5f); dummy1 = fmaf(dummy1, 1.0001f, 0.5f);
                dummy2 = fmaf(dummy2, 1.0001f, 0.5f); dummy3 = fmaf(dummy3, 1.0001f, 0.5f);
            }
#endif
            activeMask = __ballot(active);
#ifndef NX_NO_STALL_GUARD
            // A wave leaves this loop when enough of its rays have finished, or — once the queue is dry — when all have.  One that
            // is still here after kStallLimit iterations holds rays that go round in circles: not a tree.  It abandons them
            // (closest hit: they end with what they found so far; any hit: as occluded), takes no more, and tells the host.
            if (++spins > kStallLimit) {
                if (lane == 0) atomicOr(&S->frame->errorWord, kErrTraversalStalled);
                if (active) { active = false; resultPending = !ANY_HIT; }
                activeMask = 0ull;
                exhausted = true;
                rngCur = rngEnd = 0;
            }
#endif
            if (!STATS && kThinCode && thinAllowed && ((exhausted && rngCur >= rngEnd) || thinIters == 0u || thinAnyTime) && activeMask != 0ull && __popcll(activeMask) <= thinLanes && spins >= thinAfter) {
                // the wave is dry and down to its last few long rays: they go to the thin kernel (below), which puts all 64 lanes of
                // a wave on each of them; this wave is done
#if defined(NX_THIN_DROP) && defined(NX_THIN_DROP_KIND)
                if (ANY_HIT != (NX_THIN_DROP_KIND == 1)) { thinAllowed = false; }  // (bound per ray kind: the other kind's waves finish their rays themselves)
                else
#endif
#ifdef NX_THIN_DROP
                // BOUND EXPERIMENT (wrong results): the rays a hand-over rule would give away simply end here — closest hit with what they
                // have found so far, any hit as occluded — so that the launch's time is what ANY hand-over, however fast, could reach
                // at most (tools/ab_prebuilt.sh drop+NX_THIN_LANES=..+NX_THIN_ITERS=..)
                {
                    if (active) { active = false; resultPending = !ANY_HIT; }
                    activeMask = 0ull;
                    thinAllowed = false;
                }
#else
                NX_G int* const count = &C->thinCount[ANY_HIT ? 1 : 0][bounce];
                int base = 0;
                const int leader = __ffsll((long long)activeMask) - 1, n = (int)__popcll(activeMask);
                if (lane == leader) base = atomicAdd(count, n);
                base = __builtin_amdgcn_readfirstlane(__shfl(base, leader));
                const int place = base + (int)__popcll(activeMask & laneLt);
                if (active && place < (int)S->thinCapacity) {  // (every entry below min(count, capacity) is written)
                    (ANY_HIT ? S->thinAny : S->thinClosest)[place] = rayIdx;
                    // ... with the lane's loop-top state: the thin kernel continues the ray from here (ThinState, nx_device.h)
                    NX_G ThinState* st = S->thinStates + (size_t)(ANY_HIT ? (int)S->thinCapacity : 0) + (size_t)place;
                    *(NX_G uint4*)&st->ng = make_uint4(ng.x, ng.y, tg.x, tg.y);
                    *(NX_G uint4*)&st->hitT = make_uint4(__float_as_uint(hitT), __float_as_uint(hitU), __float_as_uint(hitV), hitTri);
                    // (the ray's instance record, by the instance it is inside of: its index rides in instIdx below the material code)
                    const uint32_t leaf = instSp >= 0 ? S->leafOfInstance[instIdx & kHitInstMask] : 0u;
                    *(NX_G uint4*)&st->hitInst = make_uint4(hitInst, (uint32_t)sp, (uint32_t)instSp, leaf);
                    for (int k = 0; k < min(sp, kLdsDepth + kSpillDepth); k++) {  // (entries beyond the 32nd were dropped by stack_push, as in the reference)
                        uint2 e;
                        if (k < kLdsDepth) {
                            const unsigned long long v = stackLds[k * kTraceBlock];
                            e = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
                        } else e = stackSpill[k - kLdsDepth];
                        st->stack[k] = e;
                    }
                    active = false;
                }
                activeMask = __ballot(active);
                // (once per wave; lanes the list had no room for: this wave finishes them itself.  The test hook's rule — thinIters 0 —
                //  hands over after EVERY refill as long as the list has room, so that nearly all rays of a batch go through the search
                //  however the waves share the queue)
                if ((thinIters != 0u && !thinAnyTime) || base + n > (int)S->thinCapacity) thinAllowed = false;
#ifdef NX_WAVE_TIMELINE
                wpHanded = n;
#endif
#endif  // NX_THIN_DROP
            }
        } while (activeMask != 0ull && ((exhausted && rngCur >= rngEnd) || __popcll(activeMask) >= kRefillBelow));
        itersTotal += spins;
    }

#ifdef NX_EXTRA_VALU
    if (dummy0 + dummy1 + dummy2 + dummy3 == 123.456f) S->traceStats[0].rays = 1;  // keeps the filler alive
#endif
#ifdef NX_WAVE_TIMELINE
    if (!STATS && lane == 0 && bounce < kTimelineBounces) {
        const int w = (int)blockIdx.x * (kTraceBlock / kWave) + (int)(threadIdx.x / kWave);
        if (w < kTimelineWaves) {
            unsigned long long* rec = g_waveTimeline[ANY_HIT ? 1 : 0][bounce][w];
            rec[0] = wpStart; rec[1] = wpDry; rec[2] = wall_clock64(); rec[3] = ((unsigned long long)taken << 32) | (unsigned long long)itersTotal;
            rec[4] = (unsigned long long)wpHanded;
        }
    }
#endif
    if (STATS) {
        // wave-level reduction, one atomic per wave and counter
        for (int o = 32; o > 0; o >>= 1) {
            nRays += __shfl_down(nRays, o);
            nNodes += __shfl_down(nNodes, o);
            nTris += __shfl_down(nTris, o);
            nInst += __shfl_down(nInst, o);
        }
        if (lane == 0) {
            NX_G TraceStatsDev* ts = &S->traceStats[ANY_HIT ? 1 : 0];
            atomicAdd(&ts->rays, nRays);
            atomicAdd(&ts->nodes, nNodes);
            atomicAdd(&ts->tris, nTris);
            atomicAdd(&ts->instances, nInst);
            atomicAdd(&ts->waveIters, wIters);
            atomicAdd(&ts->lanesActive, wActive);
            atomicAdd(&ts->lanesNode, wNode);
            atomicAdd(&ts->lanesPrim, wPrim);
            for (int k = 0; k < 8; k++) atomicAdd(&ts->cycles[k], cyc[k]);
        }
    }
}

template __global__ void trace_kernel<false, false>(const DeviceState*, int);
template __global__ void trace_kernel<false, true>(const DeviceState*, int);
template __global__ void trace_kernel<true, false>(const DeviceState*, int);
template __global__ void trace_kernel<true, true>(const DeviceState*, int);

// The listed rays of one level (closest-hit first, then any-hit; kThinClosestOnly / kThinAnyOnly: one list), one wave per ray,
// grid-stride.  `bounceArg` as the trace launches got it: the ray set and the meaning of the closest-hit record follow kTraceScanFlag.
#ifdef NX_THIN_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(NX_THIN_WAVES_PER_EU, NX_THIN_WAVES_PER_EU)))  // (measurement knob: DESIGN.md section 7)
#endif
__global__ void __launch_bounds__(kTraceBlock) thin_kernel(const DeviceState* __restrict__ S, const int bounceArg)
{
    __shared__ unsigned long long sPool[(kTraceBlock / kWave) * kPoolSlots];
    __shared__ float sGate[(kTraceBlock / kWave) * kPoolSlots];
    const int bounce = bounceArg & 0xff;
    const bool scan = (bounceArg & kTraceScanFlag) != 0;
    const int raySet = scan ? (bounce & 1) : 0;
    NX_G Counters* C = S->counters;
    const int cap = (int)S->thinCapacity;
    const int nClosest = (bounceArg & kThinAnyOnly) ? 0 : min(C->thinCount[0][bounce], cap);
    const int nAny = (bounceArg & kThinClosestOnly) ? 0 : min(C->thinCount[1][bounce], cap);
    if (nClosest + nAny <= 0) return;
    const bool sceneIdentity = (S->sceneFlags & kSceneAllIdentity) != 0u;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    lds_u64* const pool = (lds_u64*)&sPool[wave * kPoolSlots];
    lds_f32* const poolGate = (lds_f32*)&sGate[wave * kPoolSlots];
    // (the rare in-order replay, traverse_wave, keeps its stack in the wave's pool, which the finished search has left: 8 entries x 64 lanes)
    static_assert(kLdsDepth * kWave <= kPoolSlots, "the replay's stack fits the pool");
    lds_u64* const stackLds = pool + lane;
    const int waves = (int)gridDim.x * (kTraceBlock / kWave);
    for (int e = (int)blockIdx.x * (kTraceBlock / kWave) + wave; e < nClosest + nAny; e += waves) {
        if (e < nClosest) {
            const uint32_t word = S->thinClosest[e];
            const uint32_t slot = word & 0x7fffffffu;
            const float4 o = S->trace.rays[raySet].rayO[slot], d = S->trace.rays[raySet].rayD[slot];
            const f3 org = mk3(o.x, o.y, o.z), dir = mk3(d.x, d.y, d.z);
            // (round 6) the ray arrives with its traversal state: the search continues it — the stack's groups are its first items, the
            // hit found so far its bound — instead of starting again at the root with no bound at all
            const NX_G ThinState* st = S->thinStates + e;
            ThinResult r = thin_wave_search<false>(S, pool, poolGate, org, dir, st->hitT, sceneIdentity, st);
            // Is the closest triangle found (computed distance t, the smallest of all the ray's triangles) what the reference's order
            // returns?  It is, if that order TESTS it: nothing tested can replace it (acceptance is `t < hit distance`, strictly,
            // on computed numbers — equal distances: the first met wins, so a second triangle at exactly t is left to the replay
            // below).  It is tested iff no box on its path is culled, and a box is culled when its parent is decoded and its computed
            // entry distance exceeds the hit distance of that moment — which is the distance of some triangle accepted earlier,
            // all of them > t.  With the gate — the largest computed entry distance on the path, the kernel's own numbers — at or
            // below t, no such moment exists: whatever was accepted before, every box on the path passes.  A gate above t — the ray
            // meets the triangle in front of where it enters a box around it, by rounding: a triangle IN a face of its box, every
            // floor and wall — still leaves the path open unless some other triangle of the ray has its distance at or below the
            // gate.  So: replay only when a second triangle lies at exactly t, or between t and a gate above it; the kernel's own
            // numbers decide, no tolerance.  (Until the second half of round 5 the rule was a window — "no second triangle within 1e-3
            // of t" — which sent a few of every level's ~1 500 rays into the one-lane replay, a skimming ray's hundreds of records at
            // a microsecond each: the 100-230 us of every thin launch.)  The window still bounds the SEARCH (boxes up to t + window are
            // opened), so that those second triangles are seen: it is thousands of roundings wide, a gate exceeds t by one or two.
            const float gate = r.gate == r.gate ? r.gate : 3.0e38f;  // (a NaN entry distance never passes child_trace's test; belt and braces)
            // A continued ray: what the loop would go on to do from the handed-over state is test SOME of the remaining triangles in its
            // order, each against the hit distance of the moment, which starts at the carried hit's.  (i) No remaining triangle in
            // front of the carried hit (`replaced` false): nothing can be accepted (`t < hit distance`, strictly: a triangle AT the
            // carried distance loses to the one met first) — the carried hit is the result, no second distance matters.  (ii) A
            // remaining triangle at the smallest distance t in front of it: the argument above, with the carried distance among the
            // "distances of triangles accepted earlier" — it is: `second` starts from it when the first closer triangle is found
            // (thin_wave_search), so a gate above t with the carried hit at or below the gate sends the ray to the replay.  The seeds'
            // own gate is 0: the loop visits a popped group without testing its box again.
            const bool ambiguous = r.count != 0 && r.replaced && (r.second == r.t || (gate > r.t && r.second <= gate));
#ifdef NX_THIN_PRINTF
            if (lane == 0 && (e == 0 || r.rounds >= 24u || !r.complete || ambiguous)) printf("thin closest bounce %d list %d ray %d rounds %u pool %d complete %d ambiguous %d t %g gate %g second %g\n", bounce, nClosest, e, r.rounds, r.poolMax, (int)r.complete, (int)ambiguous, r.t, r.gate, r.second);
#endif
            float hitT = r.count ? r.t : 1e30f, hitU = r.u, hitV = r.v;
            uint32_t hitTri = r.count ? r.tri : 0xffffffffu, hitInst = r.inst;
            if (!r.complete || ambiguous) {
                // the reference's own order, one lane at work (rare: a second triangle within rounding of the closest, or a pool that ran over)
                traverse_wave<false, kWave>(S, stackLds, lane == 0, org, dir, hitT, hitU, hitV, hitTri, hitInst);
                hitT = __shfl(hitT, 0); hitU = __shfl(hitU, 0); hitV = __shfl(hitV, 0);
                hitTri = __shfl(hitTri, 0); hitInst = __shfl(hitInst, 0);
                // (traverse_wave returns the bare instance: its material code again, as inst_code_kernel writes it — needed by the SCAN
                //  pipeline's record only; the ray-batch hooks run without shading records)
                if (scan && hitTri != 0xffffffffu) {
                    const int type = (int)S->shadeInst[hitInst].material.type;
                    hitInst |= (type >= 0 && type <= 3) ? (uint32_t)(type + 1) << kHitCodeShift : 0u;
                }
            }
            if (lane == 0) {  // the record the closest-hit kernel's flush writes
                const bool missed = hitTri == 0xffffffffu;
                S->trace.hit[slot] = make_float4(hitT, hitU, hitV, __uint_as_float(hitTri));
                S->trace.hitInst[slot] = scan ? (missed ? (kHitCodeMiss << kHitCodeShift) : ((word >> 31) ? hitInst : 0u))
                                              : (missed ? 0xffffffffu : (hitInst & kHitInstMask));
            }
        } else {
            const uint32_t slot = S->thinAny[e - nClosest];
            const float4 o = S->shadow.rayO[slot], d = S->shadow.rayD[slot];
            const f3 org = mk3(o.x, o.y, o.z), dir = mk3(d.x, d.y, d.z);
            const ThinResult r = thin_wave_search<true>(S, pool, poolGate, org, dir, o.w, sceneIdentity, S->thinStates + (size_t)cap + (size_t)(e - nClosest));
            bool occluded = r.count != 0;
#ifdef NX_THIN_PRINTF
            if (lane == 0 && (e == nClosest || r.rounds >= 24u || !r.complete)) printf("thin any bounce %d list %d ray %d rounds %u pool %d complete %d occluded %d\n", bounce, nAny, e - nClosest, r.rounds, r.poolMax, (int)r.complete, (int)occluded);
#endif
            if (!occluded && !r.complete) {
                float t = o.w, u, v;
                uint32_t tri, inst;
                occluded = traverse_wave<true, kWave>(S, stackLds, lane == 0, org, dir, t, u, v, tri, inst);
                occluded = __shfl((int)occluded, 0) != 0;
            }
            if (!occluded && lane == 0) {  // pathRadiance[pixelIdx] += radiance, as the any-hit kernel's flush (BVH8Traversal.cuh:515-516)
                const float4 rad = S->shadow.radiance[slot];
                NX_G float4* dst = &S->radiance[__float_as_uint(d.w)];
                float4 cur = *dst;
                cur.x += rad.x; cur.y += rad.y; cur.z += rad.z;
                *dst = cur;
            }
        }
    }
}

const void* thin_kernel_ptr() { return (const void*)thin_kernel; }

#ifdef NX_WAVE_TIMELINE
}  // namespace nxd
// (measurement variant only) the timeline of the LAST launches of every level; clears it
extern "C" int nxhip_debug_read_wave_timeline(void* out, unsigned long long bytes)
{
    if (bytes != sizeof(nxd::g_waveTimeline)) return (int)sizeof(nxd::g_waveTimeline) > 0 ? -1 : -2;
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nxd::g_waveTimeline), bytes) != hipSuccess) return -4;
    return 0;
}
namespace nxd {
#endif

const void* trace_kernel_ptr(bool anyHit, bool stats)
{
    if (anyHit) return stats ? (const void*)trace_kernel<true, true> : (const void*)trace_kernel<true, false>;
    return stats ? (const void*)trace_kernel<false, true> : (const void*)trace_kernel<false, false>;
}

// the device-side layouts this translation unit was compiled with (nx_device.h layout_stamp; compared by nxhip_create)
uint64_t layout_stamp_trace() { return layout_stamp(); }

}  // namespace nxd
