// nx_entry.hip — entry states of the primary rays (nx_device.h EntryState): the first node steps of a run of 64 consecutive paths,
// walked once for the whole run.
//
// The reference starts every ray at the TLAS root (Cuda/BVH/BVH8Traversal.cuh:165-192).  The 64 primary rays of an 8 x 8 pixel
// tile then decode the same nodes and arrive at the same hit masks for their first four of ten node steps
// (tools/entry_point_probe.py: 40 % of the primary level's node steps).  Here one thread per run walks those steps with the run's
// ray BUNDLE — the pyramid from the camera through the bounding rectangle of the run's pixels, any sub-pixel position, any frame
// of the pass — and a conservative classification of every child box:
//   hit by all   the four edge rays of the pyramid pass through the box shrunk by a margin (the directions from a point that hit
//                a convex body form a convex cone: four edge rays inside it put the whole pyramid inside it)
//   missed by all  the box grown by the margin lies outside one of the pyramid's four side planes, or behind its apex
//   undecided    anything else: the walk stops in front of this node
// The margin (1e-4 of the coordinates involved) is a hundred times the rounding of the kernel's float slab test, and a flat box
// (a quantisation axis with exponent 0, or equal bounds) is a plane both bounds of which the kernel computes with the same
// operands, so "hit by all" and "missed by all" are statements about what the kernel's own arithmetic returns for every ray of
// the bundle.  The walk mirrors the traversal loop's bookkeeping (nx_trace.hip: node step, instance entry + root test) and ends
// at the first step whose result holds leaf bits it may not consume (triangles: from there on every ray has its own hit
// distance), at an undecided child, or at its depth limit.  What it leaves is a loop-top state of the traversal.
//
// Conditions, checked here: pinhole camera (no lens: one origin), one octant for the whole bundle, an origin without negative zero,
// infinity, NaN or denormal components, and — to go through an instance — a scene of identity instances (the ray is then the
// same in the BLAS's frame; a transformed instance ends the walk in front of it).
#define NX_KERNEL_TU 1
#include "nx_device.h"
#include "nx_math.h"

namespace nxd {

namespace {

struct D3 { double x, y, z; };
NXD D3 dsub(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
NXD D3 dcross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
NXD double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
NXD double dcomp(D3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

struct Bundle {
    D3 org;
    D3 dir[4];    // edge rays, in order round the pyramid
    D3 plane[4];  // inward normals of the side planes
    D3 axis;      // sum of the edge directions
};

enum { kMissAll = 0, kHitAll = 1, kUndecided = 2 };

// one child box [lo, hi] against the bundle
NXD int classify(const Bundle& b, const double lo[3], const double hi[3])
{
    double m[3];
    bool flat[3];
    for (int a = 0; a < 3; a++) {
        const double o = dcomp(b.org, a);
        m[a] = 1.0e-4 * (fabs(o) + fabs(lo[a]) + fabs(hi[a]) + (hi[a] - lo[a]));
        flat[a] = hi[a] == lo[a];
    }
    // missed by all: a separating axis between the grown box and the pyramid (an unbounded cone: apex + t * edge, t >= 0).  Along
    // an axis n the cone covers [n.apex, +inf) when every edge has n.edge >= 0, (-inf, n.apex] when every edge has n.edge <= 0,
    // everything otherwise; the box covers [bmin, bmax].  Candidates: the pyramid's four side planes, the axis of the view
    // (behind the apex), the box's three face normals and the twelve cross products of a box edge with a pyramid edge — the last
    // are what separates a box that lies diagonally beside a thin bundle.
    {
        const double glo[3] = {lo[0] - m[0], lo[1] - m[1], lo[2] - m[2]}, ghi[3] = {hi[0] + m[0], hi[1] + m[1], hi[2] + m[2]};
        const auto separated = [&](const D3 n) {
            int nonNeg = 0, nonPos = 0;
            for (int k = 0; k < 4; k++) {
                const double c = ddot(n, b.dir[k]);
                nonNeg += c >= 0.0 ? 1 : 0;
                nonPos += c <= 0.0 ? 1 : 0;
            }
            if (nonNeg != 4 && nonPos != 4) return false;
            const double nx[3] = {n.x, n.y, n.z};
            double bmin = 0.0, bmax = 0.0;
            for (int a = 0; a < 3; a++) {
                bmin += nx[a] >= 0.0 ? nx[a] * glo[a] : nx[a] * ghi[a];
                bmax += nx[a] >= 0.0 ? nx[a] * ghi[a] : nx[a] * glo[a];
            }
            const double apex = ddot(n, b.org);
            const double slack = 1.0e-9 * (fabs(apex) + fabs(bmin) + fabs(bmax));
            if (nonNeg == 4 && bmax < apex - slack) return true;
            if (nonPos == 4 && bmin > apex + slack) return true;
            return false;
        };
        for (int k = 0; k < 4; k++)
            if (separated(b.plane[k])) return kMissAll;
        if (separated(b.axis)) return kMissAll;
        if (separated(D3{1.0, 0.0, 0.0}) || separated(D3{0.0, 1.0, 0.0}) || separated(D3{0.0, 0.0, 1.0})) return kMissAll;
        for (int k = 0; k < 4; k++) {
            const D3 d = b.dir[k];
            if (separated(D3{0.0, -d.z, d.y}) || separated(D3{d.z, 0.0, -d.x}) || separated(D3{-d.y, d.x, 0.0})) return kMissAll;  // e_x, e_y, e_z cross d
        }
    }
    // hit by all: every edge ray through the shrunk box (a flat axis stays the plane it is; an axis too thin to shrink: undecided)
    for (int a = 0; a < 3; a++)
        if (!flat[a] && hi[a] - lo[a] <= 4.0 * m[a]) return kUndecided;
    for (int k = 0; k < 4; k++) {
        double tn = 0.0, tf = 1.0e300;
        for (int a = 0; a < 3; a++) {
            const double o = dcomp(b.org, a), d = dcomp(b.dir[k], a);  // (d != 0: one octant, with a margin)
            const double l = flat[a] ? lo[a] : lo[a] + m[a], h = flat[a] ? hi[a] : hi[a] - m[a];
            const double t0 = (l - o) / d, t1 = (h - o) / d;
            tn = fmax(tn, fmin(t0, t1));
            tf = fmin(tf, fmax(t0, t1));
        }
        // (a clear pass: the entry and exit distances apart by more than rounding, unless a flat axis makes them one plane's)
        if (!(tn <= tf) || tf <= 0.0) return kUndecided;
        if (!(flat[0] || flat[1] || flat[2]) && !(tn < tf)) return kUndecided;
    }
    return kHitAll;
}

// ChildTrace (nx_traverse.h child_trace) for the bundle: the hit mask every ray of it gets, or false when a child is undecided
NXD bool bundle_child_trace(const Bundle& b, const NX_G uint4* node, const uint32_t invOct4, uint2& ng, uint2& tg)
{
    const uint4 n0 = node[0], n1 = node[1], n2 = node[2], n3 = node[3], n4 = node[4];
    const double p[3] = {(double)__uint_as_float(n0.x), (double)__uint_as_float(n0.y), (double)__uint_as_float(n0.z)};
    const uint32_t e_imask = n0.w;
    const double scale[3] = {(double)__uint_as_float((e_imask & 0xffu) << 23), (double)__uint_as_float((e_imask << 15) & 0x7f800000u),
                             (double)__uint_as_float((e_imask << 7) & 0x7f800000u)};
    uint32_t hitMask = 0;
    for (int i = 0; i < 2; i++) {
        const uint32_t meta4 = i ? n1.w : n1.z;
        const uint32_t isInner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
        const uint32_t innerMask4 = (isInner4 >> 4) * 0xffu;
        const uint32_t bitIndex4 = (meta4 ^ (invOct4 & innerMask4)) & 0x1f1f1f1fu;
        const uint32_t childBits4 = (meta4 >> 5) & 0x07070707u;
        const uint32_t qlox = i ? n2.y : n2.x, qloy = i ? n2.w : n2.z, qloz = i ? n3.y : n3.x;
        const uint32_t qhix = i ? n3.w : n3.z, qhiy = i ? n4.y : n4.x, qhiz = i ? n4.w : n4.z;
        for (int j = 0; j < 4; j++) {
            const uint32_t childBits = (childBits4 >> (8 * j)) & 0xffu;
            if (childBits == 0u) continue;  // an empty slot sets no bit whatever its bounds say
            const uint32_t bitIndex = (bitIndex4 >> (8 * j)) & 0xffu;
            const double lo[3] = {p[0] + scale[0] * (double)((qlox >> (8 * j)) & 0xffu), p[1] + scale[1] * (double)((qloy >> (8 * j)) & 0xffu),
                                  p[2] + scale[2] * (double)((qloz >> (8 * j)) & 0xffu)};
            const double hi[3] = {p[0] + scale[0] * (double)((qhix >> (8 * j)) & 0xffu), p[1] + scale[1] * (double)((qhiy >> (8 * j)) & 0xffu),
                                  p[2] + scale[2] * (double)((qhiz >> (8 * j)) & 0xffu)};
            if (hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2]) return false;  // (an inverted slot: leave it to the kernel's own arithmetic)
            const int c = classify(b, lo, hi);
#ifdef NX_ENTRY_DEBUG
            if (blockIdx.x * blockDim.x + threadIdx.x == NX_ENTRY_DEBUG) printf("child %d.%d bits %u class %d box [%g %g %g] [%g %g %g]\n", i, j, childBits, c, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]);
#endif
            if (c == kUndecided) return false;
            if (c == kHitAll) hitMask |= childBits << bitIndex;
        }
    }
    ng = make_uint2(n1.x, (hitMask & 0xff000000u) | (e_imask >> 24));
    tg = make_uint2(n1.y, hitMask & 0x00ffffffu);
    return true;
}

}  // namespace

// the state of run `run`: the root's when a condition is not met
NXD EntryState walk_run(const DeviceState* __restrict__ S, const uint32_t run)
{
    EntryState st;
    for (int k = 0; k < kEntryMaxStack; k++) st.stack[k] = make_uint2(0u, 0u);
    st.ng = make_uint2(0u, 0x80000000u);
    st.tg = make_uint2(0u, 0u);
    st.sp = 0; st.instSp = -1; st.leafSlot = -1; st.steps = 0;
    const EntryState root = st;

    const nx_camera cam = S->camera;
    if (cam.lensRadius != 0.0f) return root;
    // the run's pixels: their bounding rectangle, in pixels
    const uint32_t resX = cam.resolution[0], resY = cam.resolution[1];
    uint32_t imin = 0xffffffffu, imax = 0u, jmin = 0xffffffffu, jmax = 0u;
    for (uint32_t k = 0; k < 64u; k++) {
        const uint32_t lp = run * 64u + k;
        if (lp >= S->localCount) break;
        const uint32_t g = S->pixelMap ? S->pixelMap[lp] : lp;
        const uint32_t j = g / resX, i = g - j * resX;
        imin = min(imin, i); imax = max(imax, i); jmin = min(jmin, j); jmax = max(jmax, j);
    }
    if (imin > imax) return root;
    Bundle b;
    b.org = D3{(double)cam.position[0], (double)cam.position[1], (double)cam.position[2]};
    {
        // NaN, +-inf, -0, +-denormal.  (+0 is fine: such a ray is not "ordinary" for nx_traverse.h's enter_instance, which then takes it
        // through the identity's rows — 1 * x + 0 * y + 0 * z + 0 — and that returns x bit for bit unless x is a NEGATIVE zero)
        constexpr int kSpecial = 0x003 | 0x004 | 0x200 | 0x020 | 0x010 | 0x080;
        if (__builtin_amdgcn_classf(cam.position[0], kSpecial) || __builtin_amdgcn_classf(cam.position[1], kSpecial) || __builtin_amdgcn_classf(cam.position[2], kSpecial)) return root;
    }
    // The bundle's margin, in pixels.  generate_kernel computes a direction as (llc + vpX * x + vpY * y) - position in binary32: each
    // component carries a few roundings of its LARGEST operand — half an ulp each for the two products, the two sums and the
    // difference, so at most 8 x 2^-24 of M = max(|llc| + |vpX| + |vpY|, |position|) per component with room to spare — and that, over the
    // world-space size of a pixel on the view plane, is how far the kernel's ray can lie from the exact one through its sub-pixel position.
    // A camera near the origin: ~1e-4 pixel, the fixed 0.01 covers it (and the float x, y themselves).  A camera far from the origin with
    // a short focus distance (position ~1e3, a view plane of a few units at 4k): tenths of a pixel — the margin follows, and beyond a
    // quarter of a pixel the run starts at the root like a run that meets no other condition.
    double eps = 0.01;
    {
        double M = 0.0, px2 = 0.0, py2 = 0.0;
        for (int a = 0; a < 3; a++) {
            M = fmax(M, fmax(fabs((double)cam.lowerLeftCorner[a]) + fabs((double)cam.viewportX[a]) + fabs((double)cam.viewportY[a]), fabs((double)cam.position[a])));
            px2 += (double)cam.viewportX[a] * (double)cam.viewportX[a];
            py2 += (double)cam.viewportY[a] * (double)cam.viewportY[a];
        }
        const double pixel = fmin(sqrt(px2) / (double)resX, sqrt(py2) / (double)resY);  // world-space size of a pixel on the view plane
        if (!(pixel > 0.0)) return root;
        eps += 1.7320508 * 8.0 * 5.9604644775390625e-8 * M / pixel;
        if (!(eps <= 0.25)) return root;
    }
    const double xs[4] = {((double)imin - eps) / resX, ((double)imax + 1.0 + eps) / resX, ((double)imax + 1.0 + eps) / resX, ((double)imin - eps) / resX};
    const double ys[4] = {((double)jmin - eps) / resY, ((double)jmin - eps) / resY, ((double)jmax + 1.0 + eps) / resY, ((double)jmax + 1.0 + eps) / resY};
    b.axis = D3{0.0, 0.0, 0.0};
    for (int k = 0; k < 4; k++) {
        b.dir[k] = D3{(double)cam.lowerLeftCorner[0] + (double)cam.viewportX[0] * xs[k] + (double)cam.viewportY[0] * ys[k] - b.org.x,
                      (double)cam.lowerLeftCorner[1] + (double)cam.viewportX[1] * xs[k] + (double)cam.viewportY[1] * ys[k] - b.org.y,
                      (double)cam.lowerLeftCorner[2] + (double)cam.viewportX[2] * xs[k] + (double)cam.viewportY[2] * ys[k] - b.org.z};
        b.axis = D3{b.axis.x + b.dir[k].x, b.axis.y + b.dir[k].y, b.axis.z + b.dir[k].z};
    }
    // one octant, with a margin: every direction of the bundle is a combination of the four with non-negative weights
    uint32_t oct = 0;
    for (int a = 0; a < 3; a++) {
        const double len = sqrt(ddot(b.axis, b.axis)) * 0.25;
        int neg = 0, pos = 0;
        for (int k = 0; k < 4; k++) {
            const double c = dcomp(b.dir[k], a);
            if (c < -1.0e-6 * len) neg++;
            else if (c > 1.0e-6 * len) pos++;
        }
        if (neg != 4 && pos != 4) return root;
        if (neg == 4) oct |= (a == 0 ? 4u : (a == 1 ? 2u : 1u));
    }
    const uint32_t invOct4 = (7u - oct) * 0x01010101u;
    for (int k = 0; k < 4; k++) {
        b.plane[k] = dcross(b.dir[k], b.dir[(k + 1) & 3]);
        if (ddot(b.plane[k], b.axis) < 0.0) b.plane[k] = D3{-b.plane[k].x, -b.plane[k].y, -b.plane[k].z};
    }

    const bool identityScene = (S->sceneFlags & kSceneAllIdentity) != 0u;
    const NX_G uint4* nodes = S->tlasNodes;
    for (;;) {
        // (loop top of nx_trace.hip's traversal: a node group with unvisited children and no pending leaf work)
        if (st.tg.y != 0u || (st.ng.y & 0xff000000u) == 0u) break;
        if (st.steps >= 8 || st.sp >= kEntryMaxStack - 2) break;
        EntryState next = st;
        const int nodeOffset = 31 - __clz((int)next.ng.y);
        next.ng.y &= ~(1u << nodeOffset);
        if (next.ng.y & 0xff000000u) next.stack[next.sp++] = next.ng;
        const int slot = (nodeOffset - 24) ^ (int)(invOct4 & 7u);
        const int rel = __popc(next.ng.y & ~(0xffffffffu << slot));
        uint2 ng, tg;
        if (!bundle_child_trace(b, nodes + (size_t)(next.ng.x + (uint32_t)rel) * (unsigned)kNodeStride, invOct4, ng, tg)) break;
        next.ng = ng;
        next.tg = tg;
        next.steps++;
        if (tg.y != 0u && next.instSp < 0) {
            // TLAS leaves: the first instance and its BLAS's root in one step, as the kernel takes them — through identity instances
            // only (the bundle is then the same in the BLAS's frame)
            if (!identityScene || next.sp >= kEntryMaxStack - 2) { st = next; break; }
            EntryState in = next;
            const int off = 31 - __clz((int)in.tg.y);
            in.tg.y &= ~(1u << off);
            const uint32_t leaf = in.tg.x + (uint32_t)off;
            if (in.tg.y) in.stack[in.sp++] = in.tg;
            if (in.ng.y & 0xff000000u) in.stack[in.sp++] = in.ng;
            in.instSp = in.sp;
            in.leafSlot = (int32_t)leaf;
            const NX_G InstTrav* rec = &S->instTrav[leaf];
            if (!bundle_child_trace(b, (const NX_G uint4*)rec->root, invOct4, ng, tg)) { st = next; break; }
            in.ng = ng;
            in.tg = tg;
            in.steps++;
            st = in;
            nodes = rec->nodes;
            continue;
        }
        st = next;
    }
    return st;
}

// One thread per run.  The table is the SLOT's own (DeviceState::entry / entryRuns of the slot whose pass graph this launch is part
// of): written here, read by the same pass's primary closest-hit launch one level later, by nobody else.
__global__ void __launch_bounds__(64) entry_state_kernel(const DeviceState* __restrict__ S)
{
    const uint32_t run = blockIdx.x * blockDim.x + threadIdx.x;
    if (run >= S->entryRuns || S->entry == nullptr) return;
    S->entry[run] = walk_run(S, run);
}

const void* entry_state_kernel_ptr() { return (const void*)entry_state_kernel; }

// the device-side layouts this translation unit was compiled with (nx_device.h layout_stamp; compared by nxhip_create)
uint64_t layout_stamp_entry() { return layout_stamp(); }

}  // namespace nxd
