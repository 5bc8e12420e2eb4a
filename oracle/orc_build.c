/*
 * orc_build.c — CPU oracle: BVH2 / BVH8 / TLAS builders, instance and camera set-up.
 *
 * TEST INFRASTRUCTURE ONLY (see nexus_oracle.h).  A serial, recursion-for-recursion restatement of
 *   /root/reference/Nexus/src/Geometry/BVH/BVH.cpp          (BVH2 binned SAH, 1 triangle per leaf)
 *   /root/reference/Nexus/src/Geometry/BVH/BVH8Builder.cpp  (Ylitie 2017 SAH-DP collapse to 80-byte nodes)
 *   /root/reference/Nexus/src/Geometry/BVH/TLAS.cpp, TLASBuilder.cpp, BVHInstance.cpp
 *   /root/reference/Nexus/src/Scene/Camera.cpp:142-168
 * The product's builder (nexus_amd/csrc/host) is a different, iterative/parallel implementation of
 * the same algorithm; tests compare the two byte for byte.
 */
#include <stdlib.h>
#include <float.h>
#include "nexus_oracle.h"
#include "orc_math.h"

/* ------------------------------------------------------------------------------------------------ */
/* AABB helpers — Geometry/AABB.h:5-36                                                             */

typedef struct { f3 bMin, bMax; } aabb;

static aabb aabb_empty(void) { aabb a; a.bMin = mk3s(1e30f); a.bMax = mk3s(-1e30f); return a; }
static void aabb_grow_point(aabb *a, f3 p) { a->bMin = min3v(a->bMin, p); a->bMax = max3v(a->bMax, p); }
static void aabb_grow(aabb *a, const aabb *o)
{
    if (o->bMin.x != 1e30f) { a->bMin = min3v(a->bMin, o->bMin); a->bMax = max3v(a->bMax, o->bMax); }
}
/* half area: Geometry/AABB.h:27-31 */
static float aabb_area(const aabb *a)
{
    const f3 d = sub3(a->bMax, a->bMin);
    return d.x * d.y + d.y * d.z + d.x * d.z;
}

/* ------------------------------------------------------------------------------------------------ */
/* BVH2 — Geometry/BVH/BVH.cpp                                                                      */

#define BINS 8

typedef struct {
    const nx_triangle *tris;
    f3 *centroid;  /* Triangle::centroid, Geometry/Triangle.h:30 */
    aabb *triBox;  /* BVH2::trianglesAABB */
    uint32_t *triIdx;
    orc_bvh2_node *nodes;
    uint32_t nodeCount, nodeCap;
} bvh2_ctx;

static uint32_t bvh2_push(bvh2_ctx *c, orc_bvh2_node n)
{
    if (c->nodeCount == c->nodeCap) {
        c->nodeCap = c->nodeCap ? c->nodeCap * 2 : 1024;
        c->nodes = (orc_bvh2_node *)realloc(c->nodes, (size_t)c->nodeCap * sizeof(orc_bvh2_node));
    }
    c->nodes[c->nodeCount] = n;
    return c->nodeCount++;
}

/* BVH.cpp:136-148 */
static void bvh2_update_bounds(bvh2_ctx *c, uint32_t nodeIdx)
{
    orc_bvh2_node *node = &c->nodes[nodeIdx];
    f3 mn = mk3s(1e30f), mx = mk3s(-1e30f);
    for (uint32_t i = 0; i < node->triCount; i++) {
        const aabb *b = &c->triBox[c->triIdx[node->leftFirst + i]];
        mn = min3v(mn, b->bMin);
        mx = max3v(mx, b->bMax);
    }
    st3(node->aabbMin, mn);
    st3(node->aabbMax, mx);
}

static float comp(f3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

/* BVH.cpp:150-210 */
static float bvh2_find_split(bvh2_ctx *c, const orc_bvh2_node *node, int *axis, double *splitPos)
{
    float bestCost = 1e30f;
    for (int a = 0; a < 3; a++) {
        float boundsMin = 1e30f, boundsMax = -1e30f;
        for (uint32_t i = 0; i < node->triCount; i++) {
            const float cc = comp(c->centroid[c->triIdx[node->leftFirst + i]], a);
            boundsMin = fminf(boundsMin, cc);
            boundsMax = fmaxf(boundsMax, cc);
        }
        if (boundsMin == boundsMax) continue;

        struct { aabb bounds; int triCount; } bins[BINS];
        for (int i = 0; i < BINS; i++) { bins[i].bounds = aabb_empty(); bins[i].triCount = 0; }
        double scale = (float)BINS / (boundsMax - boundsMin); /* float division stored in a double, BVH.cpp:166 */

        for (uint32_t i = 0; i < node->triCount; i++) {
            const uint32_t t = c->triIdx[node->leftFirst + i];
            const float cc = comp(c->centroid[t], a);
            int binIdx = (int)((cc - boundsMin) * scale);
            if (binIdx > BINS - 1) binIdx = BINS - 1;
            bins[binIdx].triCount++;
            bins[binIdx].bounds.bMin = min3v(bins[binIdx].bounds.bMin, c->triBox[t].bMin);
            bins[binIdx].bounds.bMax = max3v(bins[binIdx].bounds.bMax, c->triBox[t].bMax);
        }

        float leftArea[BINS - 1], rightArea[BINS - 1];
        int leftCount[BINS - 1], rightCount[BINS - 1];
        aabb leftBox = aabb_empty(), rightBox = aabb_empty();
        int leftSum = 0, rightSum = 0;
        for (int i = 0; i < BINS - 1; i++) {
            leftSum += bins[i].triCount;
            leftCount[i] = leftSum;
            aabb_grow(&leftBox, &bins[i].bounds);
            leftArea[i] = aabb_area(&leftBox);

            rightSum += bins[BINS - 1 - i].triCount;
            rightCount[BINS - 2 - i] = rightSum;
            aabb_grow(&rightBox, &bins[BINS - 1 - i].bounds);
            rightArea[BINS - 2 - i] = aabb_area(&rightBox);
        }

        scale = (boundsMax - boundsMin) / (float)BINS;
        for (int i = 0; i < BINS - 1; i++) {
            const float planeCost = (float)leftCount[i] * leftArea[i] + (float)rightCount[i] * rightArea[i];
            if (planeCost < bestCost) {
                *axis = a;
                *splitPos = boundsMin + scale * (i + 1);
                bestCost = planeCost;
            }
        }
    }
    return bestCost;
}

static void bvh2_subdivide(bvh2_ctx *c, uint32_t nodeIdx);

/* BVH.cpp:40-63 */
static void bvh2_split_in_half(bvh2_ctx *c, uint32_t nodeIdx)
{
    orc_bvh2_node left, right;
    memset(&left, 0, sizeof left);
    memset(&right, 0, sizeof right);
    const uint32_t first = c->nodes[nodeIdx].leftFirst, count = c->nodes[nodeIdx].triCount;
    left.leftFirst = first;
    left.triCount = count / 2;
    right.leftFirst = first + count / 2;
    right.triCount = count - count / 2;
    const uint32_t l = bvh2_push(c, left);
    const uint32_t r = bvh2_push(c, right);
    c->nodes[nodeIdx].leftFirst = l;
    c->nodes[nodeIdx].triCount = 0;
    bvh2_update_bounds(c, l);
    bvh2_update_bounds(c, r);
    bvh2_subdivide(c, l);
    bvh2_subdivide(c, r);
}

/* BVH.cpp:65-134 */
static void bvh2_subdivide(bvh2_ctx *c, uint32_t nodeIdx)
{
    orc_bvh2_node node = c->nodes[nodeIdx];
    int axis = -1;
    double splitPos = 0.0;
    (void)bvh2_find_split(c, &node, &axis, &splitPos);

    if (node.triCount == 1) return;
    if (axis == -1) { bvh2_split_in_half(c, nodeIdx); return; }

    int i = (int)node.leftFirst;
    int j = i + (int)node.triCount - 1;
    while (i <= j) {
        const float cc = comp(c->centroid[c->triIdx[i]], axis);
        if ((double)cc < splitPos) i++;
        else { const uint32_t t = c->triIdx[i]; c->triIdx[i] = c->triIdx[j]; c->triIdx[j] = t; j--; }
    }
    const int leftCount = i - (int)node.leftFirst;
    if (leftCount == 0 || leftCount == (int)node.triCount) { bvh2_split_in_half(c, nodeIdx); return; }

    orc_bvh2_node left, right;
    memset(&left, 0, sizeof left);
    memset(&right, 0, sizeof right);
    left.leftFirst = node.leftFirst;
    left.triCount = (uint32_t)leftCount;
    right.leftFirst = (uint32_t)i;
    right.triCount = node.triCount - (uint32_t)leftCount;
    const uint32_t l = bvh2_push(c, left);
    const uint32_t r = bvh2_push(c, right);
    c->nodes[nodeIdx].leftFirst = l;
    c->nodes[nodeIdx].triCount = 0;
    bvh2_update_bounds(c, l);
    bvh2_update_bounds(c, r);
    bvh2_subdivide(c, l);
    bvh2_subdivide(c, r);
}

int orc_bvh2_build(const nx_triangle *tris, uint32_t n, orc_bvh2 *out)
{
    if (!tris || n == 0 || !out) return -1;
    bvh2_ctx c;
    memset(&c, 0, sizeof c);
    c.tris = tris;
    c.centroid = (f3 *)malloc((size_t)n * sizeof(f3));
    c.triBox = (aabb *)malloc((size_t)n * sizeof(aabb));
    c.triIdx = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; i++) {
        const f3 p0 = ld3(tris[i].pos0), p1 = ld3(tris[i].pos1), p2 = ld3(tris[i].pos2);
        c.centroid[i] = div3s(add3(add3(p0, p1), p2), 3.0f); /* Geometry/Triangle.h:30 */
        aabb b = aabb_empty();                                /* BVH.cpp:28-38 */
        aabb_grow_point(&b, p0);
        aabb_grow_point(&b, p1);
        aabb_grow_point(&b, p2);
        c.triBox[i] = b;
        c.triIdx[i] = i;
    }
    orc_bvh2_node root;
    memset(&root, 0, sizeof root);
    root.leftFirst = 0;
    root.triCount = n;
    bvh2_push(&c, root);
    bvh2_update_bounds(&c, 0);
    bvh2_subdivide(&c, 0);

    out->nodes = c.nodes;
    out->nodeCount = c.nodeCount;
    out->triIdx = c.triIdx;
    out->triCount = n;
    free(c.centroid);
    free(c.triBox);
    return 0;
}

void orc_bvh2_free(orc_bvh2 *b)
{
    if (!b) return;
    free(b->nodes);
    free(b->triIdx);
    memset(b, 0, sizeof *b);
}

/* ------------------------------------------------------------------------------------------------ */
/* BVH2 -> BVH8 collapse — BVH8Builder.cpp / TLASBuilder.cpp (one generic implementation over an     */
/* abstract binary tree; the two reference files differ only in node type, leaf payload and the qhi */
/* clamp, see the diff cited in DESIGN.md)                                                          */

#define C_PRIM 0.3f /* BVH8.h:18 */
#define C_NODE 1.0f /* BVH8.h:19 */
#define P_MAX 3     /* BVH8.h:20 */
#define N_Q 8       /* BVH8.h:21 */

enum { DEC_UNDEFINED = -1, DEC_LEAF = 0, DEC_INTERNAL = 1, DEC_DISTRIBUTE = 2 };

typedef struct { float cost; int decision; int leftCount, rightCount; } node_eval;

/* generic binary tree view */
typedef struct {
    uint32_t nodeCount;
    const float (*bmin)[3]; /* via accessor below */
    int isTlas;
    const orc_bvh2_node *b2;      /* BLAS */
    const uint32_t *b2TriIdx;
    /* TLAS */
    const struct tlas_node *tn;
} tree_view;

typedef struct tlas_node { /* Geometry/BVH/TLAS.h:8-17 */
    float aabbMin[3], aabbMax[3];
    uint32_t left, right, blasCount, blasIdx;
} tlas_node;

typedef struct {
    tree_view t;
    node_eval *evals; /* nodeCount x 7 */
    int *triCount;    /* leaves below each node */
    uint32_t usedNodes, usedIndices;
    nx_bvh8_node *nodes8;
    uint32_t nodes8Cap;
    uint32_t *primIdx;
    int clampQhi;
} collapse_ctx;

static int tv_is_leaf(const tree_view *t, uint32_t n) { return t->isTlas ? (t->tn[n].left == 0) : (t->b2[n].triCount > 0); }
static uint32_t tv_left(const tree_view *t, uint32_t n) { return t->isTlas ? t->tn[n].left : t->b2[n].leftFirst; }
static uint32_t tv_right(const tree_view *t, uint32_t n) { return t->isTlas ? t->tn[n].right : t->b2[n].leftFirst + 1; }
static aabb tv_box(const tree_view *t, uint32_t n)
{
    aabb a;
    if (t->isTlas) { a.bMin = ld3(t->tn[n].aabbMin); a.bMax = ld3(t->tn[n].aabbMax); }
    else { a.bMin = ld3(t->b2[n].aabbMin); a.bMax = ld3(t->b2[n].aabbMax); }
    return a;
}
/* primitive count used by Cleaf at i == 0: BVH8Builder.cpp:83 uses m_TriCount, TLASBuilder uses blasCount */
static int tv_prim_count(const collapse_ctx *c, uint32_t n) { return c->t.isTlas ? (int)c->t.tn[n].blasCount : c->triCount[n]; }
static int tv_leaf_prims(const tree_view *t, uint32_t n) { return t->isTlas ? (int)t->tn[n].blasCount : (int)t->b2[n].triCount; }

/* BVH8Builder.cpp:119-136 */
static int compute_tri_count(collapse_ctx *c, uint32_t n)
{
    if (tv_is_leaf(&c->t, n)) c->triCount[n] = tv_leaf_prims(&c->t, n);
    else c->triCount[n] = compute_tri_count(c, tv_left(&c->t, n)) + compute_tri_count(c, tv_right(&c->t, n));
    return c->triCount[n];
}

/* BVH8Builder.cpp:28-35 */
static float c_leaf(const collapse_ctx *c, uint32_t n, int primCount)
{
    if (primCount > P_MAX) return 1.0e30f;
    aabb b = tv_box(&c->t, n);
    return aabb_area(&b) * (float)primCount * C_PRIM;
}

static float compute_node_cost(collapse_ctx *c, uint32_t n, int i);

/* BVH8Builder.cpp:37-55 */
static float c_distribute(collapse_ctx *c, uint32_t n, int j, int *leftCount, int *rightCount)
{
    float best = 1.0e30f;
    for (int k = 0; k < j; k++) {
        const float cl = compute_node_cost(c, tv_left(&c->t, n), k);
        const float cr = compute_node_cost(c, tv_right(&c->t, n), j - 1 - k);
        if (cl + cr < best) { best = cl + cr; *leftCount = k; *rightCount = j - 1 - k; }
    }
    return best;
}

/* BVH8Builder.cpp:63-117 */
static float compute_node_cost(collapse_ctx *c, uint32_t n, int i)
{
    node_eval *e = &c->evals[(size_t)n * 7 + i];
    if (e->decision != DEC_UNDEFINED) return e->cost;

    if (tv_is_leaf(&c->t, n)) {
        e->decision = DEC_LEAF;
        e->cost = c_leaf(c, n, tv_leaf_prims(&c->t, n));
        return e->cost;
    }
    if (i == 0) {
        int lc = 0, rc = 0;
        const float cLeaf = c_leaf(c, n, tv_prim_count(c, n));
        aabb b = tv_box(&c->t, n);
        const float cInternal = c_distribute(c, n, 7, &lc, &rc) + aabb_area(&b) * C_NODE;
        e = &c->evals[(size_t)n * 7 + i];
        if (cLeaf < cInternal) { e->decision = DEC_LEAF; e->cost = cLeaf; }
        else { e->decision = DEC_INTERNAL; e->cost = cInternal; e->leftCount = lc; e->rightCount = rc; }
        return e->cost;
    }
    int lc = 0, rc = 0;
    const float cDist = c_distribute(c, n, i, &lc, &rc);
    const float cFewer = compute_node_cost(c, n, i - 1);
    e = &c->evals[(size_t)n * 7 + i];
    if (cDist < cFewer) { e->decision = DEC_DISTRIBUTE; e->cost = cDist; e->leftCount = lc; e->rightCount = rc; }
    else *e = c->evals[(size_t)n * 7 + i - 1];
    return e->cost;
}

/* BVH8Builder.cpp:138-168 */
static void get_children(const collapse_ctx *c, uint32_t n, int *indices, int i, int *count)
{
    const node_eval *e = &c->evals[(size_t)n * 7 + i];
    if (e->decision == DEC_LEAF) { indices[(*count)++] = (int)n; return; }
    const uint32_t l = tv_left(&c->t, n), r = tv_right(&c->t, n);
    const node_eval *le = &c->evals[(size_t)l * 7 + e->leftCount];
    const node_eval *re = &c->evals[(size_t)r * 7 + e->rightCount];
    if (le->decision == DEC_DISTRIBUTE) get_children(c, l, indices, e->leftCount, count);
    else indices[(*count)++] = (int)l;
    if (re->decision == DEC_DISTRIBUTE) get_children(c, r, indices, e->rightCount, count);
    else indices[(*count)++] = (int)r;
}

/* BVH8Builder.cpp:170-252 */
static void order_children(const collapse_ctx *c, uint32_t parent, int *children)
{
    aabb pb = tv_box(&c->t, parent);
    const f3 pc = scale3(add3(pb.bMax, pb.bMin), 0.5f);
    float cost[8][8];
    int childCount = 0;
    for (int ch = 0; ch < 8; ch++) {
        if (children[ch] == -1) break;
        aabb cb = tv_box(&c->t, (uint32_t)children[ch]);
        const f3 cen = scale3(add3(cb.bMin, cb.bMax), 0.5f);
        const f3 d = sub3(cen, pc);
        for (int s = 0; s < 8; s++) {
            const f3 ds = mk3((s & 4) ? -1.0f : 1.0f, (s & 2) ? -1.0f : 1.0f, (s & 1) ? -1.0f : 1.0f);
            cost[ch][s] = d.x * ds.x + d.y * ds.y + d.z * ds.z; /* helper_math dot: plain products and sums */
        }
        childCount++;
    }
    int slotAssigned[8] = {0};
    int assignment[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    for (;;) {
        float minCost = FLT_MAX;
        int an = -1, as = -1;
        for (int ch = 0; ch < childCount; ch++) {
            if (assignment[ch] != -1) continue;
            for (int s = 0; s < 8; s++) {
                if (slotAssigned[s]) continue;
                if (cost[ch][s] < minCost) { minCost = cost[ch][s]; an = ch; as = s; }
            }
        }
        if (an == -1) break;
        assignment[an] = as;
        slotAssigned[as] = 1;
    }
    int cpy[8];
    memcpy(cpy, children, sizeof cpy);
    for (int i = 0; i < 8; i++) children[i] = -1;
    for (int i = 0; i < childCount; i++) children[assignment[i]] = cpy[i];
}

/* BVH8Builder.cpp:254-270 / TLASBuilder.cpp:235-246 */
static int count_prims(collapse_ctx *c, uint32_t n)
{
    if (tv_is_leaf(&c->t, n)) {
        if (c->t.isTlas) { c->primIdx[c->usedIndices++] = c->t.tn[n].blasIdx; return 1; }
        const orc_bvh2_node *b = &c->t.b2[n];
        for (uint32_t i = 0; i < b->triCount; i++) c->primIdx[c->usedIndices++] = c->t.b2TriIdx[b->leftFirst + i];
        return (int)b->triCount;
    }
    return count_prims(c, tv_left(&c->t, n)) + count_prims(c, tv_right(&c->t, n));
}

/* float -> u8 as the x86 build of the reference resolves it (cvttss2si then truncation to a byte):
 * NaN and out-of-range produce INT_MIN whose low byte is 0; 256 wraps to 0. */
static uint8_t to_byte_x86(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f || v < -2147483648.0f) return 0;
    return (uint8_t)((int32_t)v & 0xff);
}

static void ensure_nodes8(collapse_ctx *c, uint32_t n)
{
    if (n > c->nodes8Cap) {
        uint32_t cap = c->nodes8Cap ? c->nodes8Cap : 64;
        while (cap < n) cap *= 2;
        c->nodes8 = (nx_bvh8_node *)realloc(c->nodes8, (size_t)cap * sizeof(nx_bvh8_node));
        memset(c->nodes8 + c->nodes8Cap, 0, (size_t)(cap - c->nodes8Cap) * sizeof(nx_bvh8_node));
        c->nodes8Cap = cap;
    }
}

/* BVH8Builder.cpp:273-393 */
static void collapse_node(collapse_ctx *c, uint32_t n2, uint32_t n8)
{
    aabb nb = tv_box(&c->t, n2);
    const float denom = 1.0f / (float)((1 << N_Q) - 1);
    const float ex = ceilf(log2f((nb.bMax.x - nb.bMin.x) * denom));
    const float ey = ceilf(log2f((nb.bMax.y - nb.bMin.y) * denom));
    const float ez = ceilf(log2f((nb.bMax.z - nb.bMin.z) * denom));
    const float exe = exp2f(ex), eye = exp2f(ey), eze = exp2f(ez);

    nx_bvh8_node node;
    memset(&node, 0, sizeof node);
    node.e[0] = (uint8_t)(f2u(exe) >> 23);
    node.e[1] = (uint8_t)(f2u(eye) >> 23);
    node.e[2] = (uint8_t)(f2u(eze) >> 23);
    node.childBaseIdx = c->usedNodes;
    node.triangleBaseIdx = c->usedIndices;
    st3(node.p, nb.bMin);
    node.imask = 0;

    int children[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int count = 0;
    get_children(c, n2, children, 0, &count);
    order_children(c, n2, children);

    int nTrianglesTotal = 0;
    const float scaleX = 1.0f / powf(2.0f, ex);
    const float scaleY = 1.0f / powf(2.0f, ey);
    const float scaleZ = 1.0f / powf(2.0f, ez);

    for (int i = 0; i < 8; i++) {
        if (children[i] == -1) { node.meta[i] = 0; continue; }
        aabb cb = tv_box(&c->t, (uint32_t)children[i]);
        const node_eval *ev = &c->evals[(size_t)children[i] * 7 + 0];

        node.qlox[i] = to_byte_x86(floorf((cb.bMin.x - node.p[0]) * scaleX));
        node.qloy[i] = to_byte_x86(floorf((cb.bMin.y - node.p[1]) * scaleY));
        node.qloz[i] = to_byte_x86(floorf((cb.bMin.z - node.p[2]) * scaleZ));
        float hx = ceilf((cb.bMax.x - node.p[0]) * scaleX);
        float hy = ceilf((cb.bMax.y - node.p[1]) * scaleY);
        float hz = ceilf((cb.bMax.z - node.p[2]) * scaleZ);
        if (c->clampQhi) { /* TLASBuilder.cpp:314-316: std::min(x, 255.0f) (NaN stays NaN) */
            hx = (255.0f < hx) ? 255.0f : hx;
            hy = (255.0f < hy) ? 255.0f : hy;
            hz = (255.0f < hz) ? 255.0f : hz;
        }
        node.qhix[i] = to_byte_x86(hx);
        node.qhiy[i] = to_byte_x86(hy);
        node.qhiz[i] = to_byte_x86(hz);

        if (ev->decision == DEC_INTERNAL) {
            c->usedNodes++;
            node.meta[i] = (uint8_t)(0x20 | (24 + i));
            node.imask |= (uint8_t)(1u << i);
        } else if (ev->decision == DEC_LEAF) {
            const int nTri = count_prims(c, (uint32_t)children[i]);
            node.meta[i] = 0;
            for (int j = 0; j < nTri; j++) node.meta[i] |= (uint8_t)(1u << (j + 5));
            node.meta[i] |= (uint8_t)nTrianglesTotal;
            nTrianglesTotal += nTri;
        }
    }

    const uint32_t childBase = node.childBaseIdx;
    ensure_nodes8(c, c->usedNodes);
    c->nodes8[n8] = node;

    int childCount = 0;
    for (int i = 0; i < 8; i++) {
        if (children[i] == -1) continue;
        if (c->evals[(size_t)children[i] * 7 + 0].decision == DEC_INTERNAL) {
            collapse_node(c, (uint32_t)children[i], childBase + (uint32_t)childCount);
            childCount++;
        }
    }
}

static int collapse_tree(collapse_ctx *c, uint32_t primCount, orc_bvh8 *out)
{
    const uint32_t nc = c->t.nodeCount;
    c->evals = (node_eval *)malloc((size_t)nc * 7 * sizeof(node_eval));
    for (size_t i = 0; i < (size_t)nc * 7; i++) { c->evals[i].decision = DEC_UNDEFINED; c->evals[i].cost = 0; c->evals[i].leftCount = c->evals[i].rightCount = 0; }
    c->triCount = (int *)calloc(nc, sizeof(int));
    if (!c->t.isTlas) compute_tri_count(c, 0);
    (void)compute_node_cost(c, 0, 0);

    c->usedNodes = 1;
    c->usedIndices = 0;
    c->primIdx = (uint32_t *)malloc((size_t)primCount * sizeof(uint32_t));
    ensure_nodes8(c, 1);
    collapse_node(c, 0, 0);

    out->nodes = c->nodes8;
    out->nodeCount = c->usedNodes;
    out->primIdx = c->primIdx;
    out->primCount = primCount;
    free(c->evals);
    free(c->triCount);
    return 0;
}

int orc_bvh8_build(const nx_triangle *tris, uint32_t n, int clamp_qhi, orc_bvh8 *out)
{
    orc_bvh2 b2;
    if (orc_bvh2_build(tris, n, &b2) != 0) return -1;
    collapse_ctx c;
    memset(&c, 0, sizeof c);
    c.t.nodeCount = b2.nodeCount;
    c.t.isTlas = 0;
    c.t.b2 = b2.nodes;
    c.t.b2TriIdx = b2.triIdx;
    c.clampQhi = clamp_qhi;
    const int rc = collapse_tree(&c, n, out);
    orc_bvh2_free(&b2);
    return rc;
}

void orc_bvh8_free(orc_bvh8 *b)
{
    if (!b) return;
    free(b->nodes);
    free(b->primIdx);
    memset(b, 0, sizeof *b);
}

/* ------------------------------------------------------------------------------------------------ */
/* Mat4 — Math/Mat4.h, Math/Mat4.cpp (host-side, plain products and sums, no fma)                   */

void orc_mat4_identity(nx_mat4 *m)
{
    memset(m, 0, sizeof *m);
    m->cell[0] = m->cell[5] = m->cell[10] = m->cell[15] = 1.0f;
}

/* Mat4.cpp:3-16 */
void orc_mat4_mul(const nx_mat4 *a, const nx_mat4 *b, nx_mat4 *out)
{
    nx_mat4 r;
    for (int i = 0; i < 16; i += 4)
        for (int j = 0; j < 4; ++j)
            r.cell[i + j] = (a->cell[i + 0] * b->cell[j + 0]) + (a->cell[i + 1] * b->cell[j + 4]) +
                            (a->cell[i + 2] * b->cell[j + 8]) + (a->cell[i + 3] * b->cell[j + 12]);
    *out = r;
}

/* Mat4.h:151-194 — cofactor expansion ("from MESA"), row-major cells */
void orc_mat4_invert(const nx_mat4 *mm, nx_mat4 *out)
{
    const float *m = mm->cell;
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    nx_mat4 r;
    orc_mat4_identity(&r);
    if (det != 0) {
        const float invdet = 1.0f / det;
        for (int i = 0; i < 16; i++) r.cell[i] = inv[i] * invdet;
    }
    *out = r;
}

static float to_radians(float deg) { return (float)(deg * ORC_PI / 180.0f); } /* Utils/Utils.h:30-33 */

/* BVHInstance.cpp:24-29: Translate * RotateZ * RotateY * RotateX * Scale, Mat4.h:59-69,138-140 */
void orc_mat4_from_trs(const float pos[3], const float rotDeg[3], const float scale[3], nx_mat4 *out)
{
    nx_mat4 T, Rz, Ry, Rx, S, t0, t1, t2;
    orc_mat4_identity(&T); orc_mat4_identity(&Rz); orc_mat4_identity(&Ry); orc_mat4_identity(&Rx); orc_mat4_identity(&S);
    T.cell[3] = pos[0]; T.cell[7] = pos[1]; T.cell[11] = pos[2];
    const float az = to_radians(rotDeg[2]), ay = to_radians(rotDeg[1]), ax = to_radians(rotDeg[0]);
    Rz.cell[0] = cosf(az); Rz.cell[1] = -sinf(az); Rz.cell[4] = sinf(az); Rz.cell[5] = cosf(az);
    Ry.cell[0] = cosf(ay); Ry.cell[2] = sinf(ay); Ry.cell[8] = -sinf(ay); Ry.cell[10] = cosf(ay);
    Rx.cell[5] = cosf(ax); Rx.cell[6] = -sinf(ax); Rx.cell[9] = sinf(ax); Rx.cell[10] = cosf(ax);
    S.cell[0] = scale[0]; S.cell[5] = scale[1]; S.cell[10] = scale[2];
    orc_mat4_mul(&T, &Rz, &t0);
    orc_mat4_mul(&t0, &Ry, &t1);
    orc_mat4_mul(&t1, &Rx, &t2);
    orc_mat4_mul(&t2, &S, out);
}

/* BVHInstance.cpp:4-22, BVHInstance::ToDevice :36-45 */
void orc_instance_init(nx_bvh_instance *inst, uint32_t bvhIdx, int32_t materialId, const nx_mat4 *transform,
                       const nx_bvh8_node *root)
{
    memset(inst, 0, sizeof *inst);
    inst->bvhIdx = bvhIdx;
    inst->materialId = materialId;
    inst->transform = *transform;
    orc_mat4_invert(transform, &inst->invTransform);
    const f3 bMin = ld3(root->p);
    const float k = exp2f(8.0f) - 1.0f;
    const f3 bMax = add3(bMin, scale3(mk3(exp2f((float)(root->e[0] - 127)), exp2f((float)(root->e[1] - 127)),
                                          exp2f((float)(root->e[2] - 127))), k));
    aabb b = aabb_empty();
    const float *c = transform->cell;
    for (int i = 0; i < 8; i++) {
        const f3 p = mk3((i & 1) ? bMax.x : bMin.x, (i & 2) ? bMax.y : bMin.y, (i & 4) ? bMax.z : bMin.z);
        /* TransformPosition = float4(a,1) * M, Mat4.cpp:58-64,66-69 */
        const f3 q = mk3(c[0] * p.x + c[1] * p.y + c[2] * p.z + c[3] * 1.0f,
                         c[4] * p.x + c[5] * p.y + c[6] * p.z + c[7] * 1.0f,
                         c[8] * p.x + c[9] * p.y + c[10] * p.z + c[11] * 1.0f);
        aabb_grow_point(&b, q);
    }
    st3(inst->boundsMin, b.bMin);
    st3(inst->boundsMax, b.bMax);
}

/* ------------------------------------------------------------------------------------------------ */
/* TLAS — Geometry/BVH/TLAS.cpp:13-91                                                               */

static int tlas_find_best_match(const tlas_node *nodes, const uint32_t *idx, int N, int A)
{
    float smallest = 1e30f;
    int bestB = -1;
    for (int B = 0; B < N; B++) {
        if (B == A) continue;
        const f3 bMax = max3v(ld3(nodes[idx[A]].aabbMax), ld3(nodes[idx[B]].aabbMax));
        const f3 bMin = min3v(ld3(nodes[idx[A]].aabbMin), ld3(nodes[idx[B]].aabbMin));
        const f3 e = sub3(bMax, bMin);
        const float area = e.x * e.y + e.y * e.z + e.x * e.z;
        if (area < smallest) { smallest = area; bestB = B; }
    }
    return bestB;
}

int orc_tlas_build(const nx_bvh_instance *instances, uint32_t n, orc_bvh8 *out)
{
    if (!instances || n == 0 || !out) return -1;
    tlas_node *nodes = (tlas_node *)calloc((size_t)2 * n + 1, sizeof(tlas_node));
    uint32_t *idx = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    uint32_t nodeCount = 1;
    for (uint32_t i = 0; i < n; i++) {
        idx[i] = i + 1;
        tlas_node nd;
        memset(&nd, 0, sizeof nd);
        memcpy(nd.aabbMin, instances[i].boundsMin, 12);
        memcpy(nd.aabbMax, instances[i].boundsMax, 12);
        nd.blasIdx = i;
        nd.blasCount = 1;
        nodes[nodeCount++] = nd;
    }
    int nodeIndices = (int)n;
    int A = 0, B = tlas_find_best_match(nodes, idx, nodeIndices, A);
    while (nodeIndices > 1) {
        const int C = tlas_find_best_match(nodes, idx, nodeIndices, B);
        if (A == C) {
            const uint32_t ia = idx[A], ib = idx[B];
            tlas_node nn;
            memset(&nn, 0, sizeof nn);
            nn.left = ib;
            nn.right = ia;
            nn.blasCount = nodes[ia].blasCount + nodes[ib].blasCount;
            st3(nn.aabbMin, min3v(ld3(nodes[ia].aabbMin), ld3(nodes[ib].aabbMin)));
            st3(nn.aabbMax, max3v(ld3(nodes[ia].aabbMax), ld3(nodes[ib].aabbMax)));
            idx[A] = nodeCount;
            idx[B] = idx[nodeIndices - 1];
            nodes[nodeCount++] = nn;
            B = tlas_find_best_match(nodes, idx, --nodeIndices, A);
        } else { A = B; B = C; }
    }
    nodes[0] = nodes[idx[A]];

    collapse_ctx c;
    memset(&c, 0, sizeof c);
    c.t.nodeCount = nodeCount;
    c.t.isTlas = 1;
    c.t.tn = nodes;
    c.clampQhi = 1;
    const int rc = collapse_tree(&c, n, out);
    free(nodes);
    free(idx);
    return rc;
}

/* ------------------------------------------------------------------------------------------------ */
/* Camera::ToDevice — Scene/Camera.cpp:30-35 (right = cross(forward, +Y)), :142-168                 */

static f3 cross_plain(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

void orc_camera_init(nx_camera *cam, const float position[3], const float forward[3], float hfov, uint32_t width,
                     uint32_t height, float focusDist, float defocusAngle)
{
    memset(cam, 0, sizeof *cam);
    const f3 pos = ld3(position), fwd = ld3(forward);
    const f3 right = cross_plain(fwd, mk3(0.0f, 1.0f, 0.0f));
    const f3 up = cross_plain(right, fwd);
    const float aspect = (float)width / (float)height;
    const float halfWidth = focusDist * tanf((float)(hfov / 2.0f * ORC_PI / 180.0f));
    const float halfHeight = halfWidth / aspect;
    const f3 vx = scale3(right, 2 * halfWidth);
    const f3 vy = scale3(up, 2 * halfHeight);
    const f3 llc = add3(sub3(sub3(pos, div3s(vx, 2.0f)), div3s(vy, 2.0f)), scale3(fwd, focusDist));
    const float lensRadius = focusDist * tanf((float)(defocusAngle / 2.0f * ORC_PI / 180.0f));
    st3(cam->position, pos);
    st3(cam->right, right);
    st3(cam->up, up);
    cam->lensRadius = lensRadius;
    st3(cam->lowerLeftCorner, llc);
    st3(cam->viewportX, vx);
    st3(cam->viewportY, vy);
    cam->resolution[0] = width;
    cam->resolution[1] = height;
}
