/*
 * orc_trace.c — CPU oracle: compressed BVH8 two-level traversal (closest hit / any hit), brute-force
 * ground truth and a BVH2 traversal.
 *
 * TEST INFRASTRUCTURE ONLY (see nexus_oracle.h).  Restates, with serial semantics,
 *   /root/reference/Nexus/src/Cuda/BVH/BVH8Traversal.cuh:55-146   ChildTrace
 *   /root/reference/Nexus/src/Cuda/BVH/BVH8Traversal.cuh:148-322  BVH8Trace
 *   /root/reference/Nexus/src/Cuda/BVH/BVH8Traversal.cuh:326-518  BVH8TraceShadow
 *   /root/reference/Nexus/src/Cuda/Geometry/Triangle.cuh:53-118   Trace / ShadowTrace (Moeller-Trumbore)
 *   /root/reference/Nexus/src/Cuda/BVH/BVH2Traversal.cuh:7-52     IntersectBVH2 (dead code in the reference)
 * "Serial" = one lane: __activemask() is all lanes, so the triangle-postponing and lost-work
 * heuristics (which only reorder work between lanes) never trigger; neither can change a result.
 */
#include <stdlib.h>
#include <pthread.h>
#include "nexus_oracle.h"
#include "orc_math.h"

#define ORC_STACK 128 /* reference: TRAVERSAL_STACK_SIZE 32 (BVH8Traversal.cuh:17); maxStack is reported */

typedef struct { f3 origin, direction, invDirection; } ray_t;

static ray_t make_ray(f3 o, f3 d)
{
    ray_t r;
    r.origin = o;
    r.direction = d;
    r.invDirection = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); /* Cuda/Geometry/Ray.cuh:55-56 */
    return r;
}

static uint32_t octant(f3 a) { return ((a.x < 0 ? 1u : 0u) << 2) | ((a.y < 0 ? 1u : 0u) << 1) | (a.z < 0 ? 1u : 0u); }

static uint32_t extract_byte(uint32_t x, uint32_t i) { return (x >> (i * 8)) & 0xff; }

/* prmt.b32 v, i, 0, 0xBA98: replicate the sign bit of each byte (Cuda/Utils.cuh:10-13) */
static uint32_t sign_extend_s8x4(uint32_t i)
{
    uint32_t v = 0;
    for (int b = 0; b < 4; b++)
        if (i & (0x80u << (8 * b))) v |= 0xffu << (8 * b);
    return v;
}

/* vmax.s32.s32.s32.max / vmin...min on float bit patterns (Cuda/Utils.cuh:16-33) */
static float vmaxmax(float a, float b, float c)
{
    int32_t x = f2i(a), y = f2i(b), z = f2i(c);
    int32_t m = x > y ? x : y;
    m = m > z ? m : z;
    return i2f(m);
}
static float vminmin(float a, float b, float c)
{
    int32_t x = f2i(a), y = f2i(b), z = f2i(c);
    int32_t m = x < y ? x : y;
    m = m < z ? m : z;
    return i2f(m);
}

/* NaNs from 0 * inf are canonicalised to the positive quiet NaN the GPU produces (x86 produces the
 * negative one); the signed-int ordering above depends on the NaN's sign bit. */
static float canon(float x) { return (x != x) ? u2f(0x7fc00000u) : x; }

static uint32_t load_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

/* ChildTrace — BVH8Traversal.cuh:55-146 */
static void child_trace(const nx_bvh8_node *n, const ray_t *ray, uint32_t invOctant4, float hitDistance,
                        uint32_t internalEntry[2], uint32_t triangleEntry[2])
{
    const f3 p = ld3(n->p);
    const f3 tdir = mk3(u2f((uint32_t)n->e[0] << 23) * ray->invDirection.x,
                        u2f((uint32_t)n->e[1] << 23) * ray->invDirection.y,
                        u2f((uint32_t)n->e[2] << 23) * ray->invDirection.z);
    const f3 torg = mul3(sub3(p, ray->origin), ray->invDirection);
    uint32_t hitMask = 0;

    for (int i = 0; i < 2; i++) {
        const uint32_t meta4 = load_u32(n->meta + 4 * i);
        const uint32_t isInner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
        const uint32_t innerMask4 = sign_extend_s8x4(isInner4 << 3);
        const uint32_t bitIndex4 = (meta4 ^ (invOctant4 & innerMask4)) & 0x1f1f1f1fu;
        const uint32_t childBits4 = (meta4 >> 5) & 0x07070707u;

        const uint32_t qlox = load_u32(n->qlox + 4 * i), qhix = load_u32(n->qhix + 4 * i);
        const uint32_t qloy = load_u32(n->qloy + 4 * i), qhiy = load_u32(n->qhiy + 4 * i);
        const uint32_t qloz = load_u32(n->qloz + 4 * i), qhiz = load_u32(n->qhiz + 4 * i);

        const uint32_t xMin = ray->direction.x < 0.0f ? qhix : qlox, xMax = ray->direction.x < 0.0f ? qlox : qhix;
        const uint32_t yMin = ray->direction.y < 0.0f ? qhiy : qloy, yMax = ray->direction.y < 0.0f ? qloy : qhiy;
        const uint32_t zMin = ray->direction.z < 0.0f ? qhiz : qloz, zMax = ray->direction.z < 0.0f ? qloz : qhiz;

        for (uint32_t j = 0; j < 4; j++) {
            const float tminx = canon(fmaf((float)extract_byte(xMin, j), tdir.x, torg.x));
            const float tminy = canon(fmaf((float)extract_byte(yMin, j), tdir.y, torg.y));
            const float tminz = canon(fmaf((float)extract_byte(zMin, j), tdir.z, torg.z));
            const float tmaxx = canon(fmaf((float)extract_byte(xMax, j), tdir.x, torg.x));
            const float tmaxy = canon(fmaf((float)extract_byte(yMax, j), tdir.y, torg.y));
            const float tmaxz = canon(fmaf((float)extract_byte(zMax, j), tdir.z, torg.z));
            const float tmin = vmaxmax(tminx, tminy, fmaxf(tminz, 0.0f));
            const float tmax = vminmin(tmaxx, tmaxy, fminf(tmaxz, hitDistance));
            if (tmin <= tmax) {
                const uint32_t childBits = extract_byte(childBits4, j);
                const uint32_t bitIndex = extract_byte(bitIndex4, j);
                hitMask |= childBits << bitIndex;
            }
        }
    }
    internalEntry[0] = n->childBaseIdx;
    internalEntry[1] = (hitMask & 0xff000000u) | n->imask;
    triangleEntry[0] = n->triangleBaseIdx;
    triangleEntry[1] = hitMask & 0x00ffffffu;
}

void orc_child_trace(const nx_bvh8_node *node, const float origin[3], const float direction[3], float tmax,
                     uint32_t out_entries[4])
{
    const ray_t r = make_ray(ld3(origin), ld3(direction));
    const uint32_t invOctant = 7 - octant(r.direction);
    child_trace(node, &r, invOctant * 0x01010101u, tmax, out_entries, out_entries + 2);
}

/* D_Triangle::Trace — Cuda/Geometry/Triangle.cuh:53-86.  Returns 1 and updates *t,*u,*v on a closer hit. */
static int tri_trace(const nx_triangle *tri, const ray_t *r, float *tBest, float *uOut, float *vOut)
{
    const f3 p0 = ld3(tri->pos0);
    const f3 edge0 = sub3(ld3(tri->pos1), p0);
    const f3 edge1 = sub3(ld3(tri->pos2), p0);
    const f3 rayCrossEdge1 = cross3(r->direction, edge1);
    const float det = dot3(edge0, rayCrossEdge1);
    const float invDet = 1.0f / det;
    const f3 s = sub3(r->origin, p0);
    const float u = invDet * dot3(s, rayCrossEdge1);
    if (u < 0.0f || u > 1.0f) return 0;
    const f3 sCrossEdge0 = cross3(s, edge0);
    const float v = invDet * dot3(r->direction, sCrossEdge0);
    if (v < 0.0f || u + v > 1.0f) return 0;
    const float t = invDet * dot3(edge1, sCrossEdge0);
    if (t > 0.0f && t < *tBest) { *tBest = t; *uOut = u; *vOut = v; return 1; }
    return 0;
}

/* Optional step log (tools/lane_sim.py: what a wave's lanes would be doing under different loop policies).  Per traced ray the
 * kinds of the records it visits, in visiting order — 1 node, 2 triangle, 3 instance entry — closed by a 0.  Off unless set. */
static __thread uint8_t *g_stepLog = NULL;
static __thread uint64_t g_stepLogCap = 0, g_stepLogLen = 0;
void orc_trace_set_step_log(uint8_t *buf, uint64_t cap) { g_stepLog = buf; g_stepLogCap = cap; g_stepLogLen = 0; }
uint64_t orc_trace_step_log_length(void) { return g_stepLogLen; }
static inline void log_step(uint8_t kind) { if (g_stepLog && g_stepLogLen < g_stepLogCap) g_stepLog[g_stepLogLen++] = kind; }
/* Optional second log, for tools/entry_point_probe.py: per NODE step of the closest-hit traversal three words — which node
 * (instance + 1 in the high half, 0 = TLAS), and the two hit masks ChildTrace produced for this ray (inner children, leaf
 * primitives) — and per ray a terminating triple of zeros.  Two rays whose triples agree up to some step have, up to there,
 * visited the same nodes and hold the same traversal stack. */
static __thread uint64_t *g_nodeLog = NULL;
static __thread uint64_t g_nodeLogCap = 0, g_nodeLogLen = 0;
void orc_trace_set_node_log(uint64_t *buf, uint64_t capWords) { g_nodeLog = buf; g_nodeLogCap = capWords; g_nodeLogLen = 0; }
uint64_t orc_trace_node_log_length(void) { return g_nodeLogLen; }
static inline void log_node(uint64_t id, uint32_t inner, uint32_t leaf)
{
    if (g_nodeLog && g_nodeLogLen + 3 <= g_nodeLogCap) { g_nodeLog[g_nodeLogLen++] = id; g_nodeLog[g_nodeLogLen++] = inner; g_nodeLog[g_nodeLogLen++] = leaf; }
}

static int clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; }
static int popc32(uint32_t x) { return __builtin_popcount(x); }

typedef struct { uint32_t x, y; } u2;

/* One ray, closest hit (anyHit == 0) or any hit within tmaxIn (anyHit == 1; returns 1 if occluded). */
int orc_trace_one(const orc_scene *s, const float org[3], const float dir[3], int anyHit, float tmaxIn, nx_hit *hit,
                  orc_trace_stats *st);
static int trace_one(const orc_scene *s, f3 org, f3 dir, int anyHit, float tmaxIn, nx_hit *hit, orc_trace_stats *st)
{
    u2 stack[ORC_STACK];
    int stackPtr = 0;
    ray_t ray = make_ray(org, dir);
    const ray_t backup = ray;
    uint32_t invOctant = 7 - octant(ray.direction);
    uint32_t invOctant4 = invOctant * 0x01010101u;
    float hitDistance = anyHit ? tmaxIn : 1e30f;
    float hu = 0.0f, hv = 0.0f;
    uint32_t hTri = 0xffffffffu, hInst = 0xffffffffu;
    int instanceStackDepth = -1;
    uint32_t instanceIdx = 0;
    const nx_bvh8_node *nodes = s->tlasNodes;
    const orc_blas *bvh = NULL;
    u2 nodeEntry = {0, 0x80000000u};
    u2 triangleEntry = {0, 0};
    uint64_t nNodes = 0, nTris = 0, nInst = 0, maxStack = 0;
    int occluded = 0;

    for (;;) {
        if (nodeEntry.y & 0xff000000u) {
            const int nodeOffset = 31 - clz32(nodeEntry.y);
            nodeEntry.y &= ~(1u << nodeOffset);
            if (nodeEntry.y & 0xff000000u) stack[stackPtr++] = nodeEntry;
            const int nodeSlot = (nodeOffset - 24) ^ (int)invOctant;
            const int relativeNodeIdx = popc32(nodeEntry.y & ~(0xffffffffu << nodeSlot));
            uint32_t ie[2], te[2];
            const uint32_t nodeLogIndex = nodeEntry.x + (uint32_t)relativeNodeIdx;
            child_trace(&nodes[nodeLogIndex], &ray, invOctant4, hitDistance, ie, te);
            nodeEntry.x = ie[0]; nodeEntry.y = ie[1];
            triangleEntry.x = te[0]; triangleEntry.y = te[1];
            nNodes++;
            log_step(1);
            if (g_nodeLog && !anyHit)
                log_node(((uint64_t)(instanceStackDepth == -1 ? 0u : instanceIdx + 1u) << 32) | (uint64_t)nodeLogIndex, ie[1], te[1]);
        } else {
            triangleEntry = nodeEntry;
            nodeEntry.x = 0; nodeEntry.y = 0;
        }

        while (triangleEntry.y) {
            if (instanceStackDepth == -1) {
                const int triangleOffset = 31 - clz32(triangleEntry.y);
                triangleEntry.y &= ~(1u << triangleOffset);
                instanceIdx = s->tlasInstIdx[triangleEntry.x + (uint32_t)triangleOffset];
                if (triangleEntry.y) stack[stackPtr++] = triangleEntry;
                if (nodeEntry.y & 0xff000000u) stack[stackPtr++] = nodeEntry;
                instanceStackDepth = stackPtr;
                const nx_bvh_instance *inst = &s->instances[instanceIdx];
                bvh = &s->blas[inst->bvhIdx];
                nodes = bvh->nodes;
                nodeEntry.x = 0; nodeEntry.y = 0x80000000u;
                /* octant from the untransformed direction, then transform (BVH8Traversal.cuh:259-264) */
                invOctant = 7 - octant(ray.direction);
                invOctant4 = invOctant * 0x01010101u;
                ray.origin = mat_point(&inst->invTransform, ray.origin);
                ray.direction = mat_vec(&inst->invTransform, ray.direction);
                ray.invDirection = mk3(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
                nInst++;
                log_step(3);
                break;
            }
            const int triangleOffset = 31 - clz32(triangleEntry.y);
            triangleEntry.y &= ~(1u << triangleOffset);
            const uint32_t triangleIdx = bvh->triIdx[triangleEntry.x + (uint32_t)triangleOffset];
            nTris++;
            log_step(2);
            if (anyHit) {
                float t = hitDistance, u, v; /* ShadowTrace: t > 0 && t < hitDistance, Triangle.cuh:89-118 */
                if (tri_trace(&bvh->tris[triangleIdx], &ray, &t, &u, &v)) { occluded = 1; break; }
            } else if (tri_trace(&bvh->tris[triangleIdx], &ray, &hitDistance, &hu, &hv)) {
                hTri = triangleIdx;
                hInst = instanceIdx;
            }
        }
        if ((uint64_t)stackPtr > maxStack) maxStack = (uint64_t)stackPtr;
        if (occluded) break;

        if ((nodeEntry.y & 0xff000000u) == 0) {
            if (stackPtr == 0) break;
            if (stackPtr == instanceStackDepth) {
                ray = backup;
                invOctant = 7 - octant(ray.direction);
                invOctant4 = invOctant * 0x01010101u;
                nodes = s->tlasNodes;
                instanceStackDepth = -1;
            }
            nodeEntry = stack[--stackPtr];
        }
    }
    log_step(0);
    if (g_nodeLog && !anyHit) log_node(0, 0, 0);
    if (hit) {
        hit->hitDistance = hitDistance;
        hit->u = hu; hit->v = hv;
        hit->triIdx = hTri; hit->instanceIdx = hInst;
    }
    if (st) {
        st->rays++; st->nodes += nNodes; st->tris += nTris; st->instances += nInst;
        if (maxStack > st->maxStack) st->maxStack = maxStack;
    }
    return occluded;
}

int orc_trace_one(const orc_scene *s, const float org[3], const float dir[3], int anyHit, float tmaxIn, nx_hit *hit,
                  orc_trace_stats *st)
{
    return trace_one(s, ld3(org), ld3(dir), anyHit, tmaxIn, hit, st);
}

void orc_trace_closest(const orc_scene *s, const nx_ray *rays, uint32_t n, nx_hit *hits, orc_trace_stats *stats)
{
    for (uint32_t i = 0; i < n; i++) trace_one(s, ld3(rays[i].origin), ld3(rays[i].direction), 0, 0.0f, &hits[i], stats);
}

void orc_trace_any(const orc_scene *s, const nx_ray *rays, const float *tmax, uint32_t n, uint8_t *occluded,
                   orc_trace_stats *stats)
{
    for (uint32_t i = 0; i < n; i++)
        occluded[i] = (uint8_t)trace_one(s, ld3(rays[i].origin), ld3(rays[i].direction), 1, tmax[i], NULL, stats);
}

typedef struct { const orc_scene *s; const nx_ray *rays; nx_hit *hits; uint32_t begin, end; } mt_job;

static void *mt_worker(void *arg)
{
    mt_job *j = (mt_job *)arg;
    for (uint32_t i = j->begin; i < j->end; i++)
        trace_one(j->s, ld3(j->rays[i].origin), ld3(j->rays[i].direction), 0, 0.0f, &j->hits[i], NULL);
    return NULL;
}

void orc_trace_closest_mt(const orc_scene *s, const nx_ray *rays, uint32_t n, nx_hit *hits, int nthreads)
{
    if (nthreads <= 1 || n < 1024) { orc_trace_closest(s, rays, n, hits, NULL); return; }
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    mt_job jobs[256];
    const uint32_t chunk = (n + (uint32_t)nthreads - 1) / (uint32_t)nthreads;
    int started = 0;
    for (int t = 0; t < nthreads; t++) {
        const uint32_t b = (uint32_t)t * chunk;
        if (b >= n) break;
        jobs[t].s = s; jobs[t].rays = rays; jobs[t].hits = hits;
        jobs[t].begin = b; jobs[t].end = (b + chunk < n) ? b + chunk : n;
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------------------------------------ */
/* brute force ground truth                                                                         */

void orc_brute_closest(const orc_scene *s, const nx_ray *rays, uint32_t n, nx_hit *hits)
{
    for (uint32_t i = 0; i < n; i++) {
        const f3 o = ld3(rays[i].origin), d = ld3(rays[i].direction);
        nx_hit h = {1e30f, 0.0f, 0.0f, 0xffffffffu, 0xffffffffu};
        for (uint32_t k = 0; k < s->instanceCount; k++) {
            const nx_bvh_instance *inst = &s->instances[k];
            const orc_blas *b = &s->blas[inst->bvhIdx];
            const ray_t r = make_ray(mat_point(&inst->invTransform, o), mat_vec(&inst->invTransform, d));
            for (uint32_t t = 0; t < b->triCount; t++)
                if (tri_trace(&b->tris[t], &r, &h.hitDistance, &h.u, &h.v)) { h.triIdx = t; h.instanceIdx = k; }
        }
        hits[i] = h;
    }
}

void orc_brute_any(const orc_scene *s, const nx_ray *rays, const float *tmax, uint32_t n, uint8_t *occluded)
{
    for (uint32_t i = 0; i < n; i++) {
        const f3 o = ld3(rays[i].origin), d = ld3(rays[i].direction);
        uint8_t occ = 0;
        for (uint32_t k = 0; k < s->instanceCount && !occ; k++) {
            const nx_bvh_instance *inst = &s->instances[k];
            const orc_blas *b = &s->blas[inst->bvhIdx];
            const ray_t r = make_ray(mat_point(&inst->invTransform, o), mat_vec(&inst->invTransform, d));
            for (uint32_t t = 0; t < b->triCount; t++) {
                float tt = tmax[i], u, v;
                if (tri_trace(&b->tris[t], &r, &tt, &u, &v)) { occ = 1; break; }
            }
        }
        occluded[i] = occ;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* BVH2 traversal — algorithm of BVH2Traversal.cuh:7-52 with D_AABB::IntersectionAABB (AABB.cuh:11-20) */

static float intersect_aabb(const ray_t *r, float hitDistance, const float *bmin, const float *bmax)
{
    const float tx1 = (bmin[0] - r->origin.x) * r->invDirection.x, tx2 = (bmax[0] - r->origin.x) * r->invDirection.x;
    float tmin = fminf(tx1, tx2), tmax = fmaxf(tx1, tx2);
    const float ty1 = (bmin[1] - r->origin.y) * r->invDirection.y, ty2 = (bmax[1] - r->origin.y) * r->invDirection.y;
    tmin = fmaxf(tmin, fminf(ty1, ty2)); tmax = fminf(tmax, fmaxf(ty1, ty2));
    const float tz1 = (bmin[2] - r->origin.z) * r->invDirection.z, tz2 = (bmax[2] - r->origin.z) * r->invDirection.z;
    tmin = fmaxf(tmin, fminf(tz1, tz2)); tmax = fminf(tmax, fmaxf(tz1, tz2));
    if (tmax >= tmin && tmin < hitDistance && tmax > 0) return tmin;
    return 1e30f;
}

void orc_bvh2_trace_closest(const orc_bvh2 *b, const nx_triangle *tris, const nx_ray *rays, uint32_t n, nx_hit *hits)
{
    for (uint32_t i = 0; i < n; i++) {
        const ray_t r = make_ray(ld3(rays[i].origin), ld3(rays[i].direction));
        nx_hit h = {1e30f, 0.0f, 0.0f, 0xffffffffu, 0xffffffffu};
        const orc_bvh2_node *node = &b->nodes[0];
        const orc_bvh2_node *stack[64];
        uint32_t sp = 0;
        for (;;) {
            if (node->triCount > 0) {
                for (uint32_t k = 0; k < node->triCount; k++) {
                    const uint32_t t = b->triIdx[node->leftFirst + k];
                    if (tri_trace(&tris[t], &r, &h.hitDistance, &h.u, &h.v)) { h.triIdx = t; h.instanceIdx = 0; }
                }
                if (sp == 0) break;
                node = stack[--sp];
                continue;
            }
            const orc_bvh2_node *c1 = &b->nodes[node->leftFirst], *c2 = &b->nodes[node->leftFirst + 1];
            float d1 = intersect_aabb(&r, h.hitDistance, c1->aabbMin, c1->aabbMax);
            float d2 = intersect_aabb(&r, h.hitDistance, c2->aabbMin, c2->aabbMax);
            if (d1 > d2) { const float td = d1; d1 = d2; d2 = td; const orc_bvh2_node *tc = c1; c1 = c2; c2 = tc; }
            if (d1 == 1e30f) {
                if (sp == 0) break;
                node = stack[--sp];
            } else {
                node = c1;
                if (d2 != 1e30f) stack[sp++] = c2;
            }
        }
        hits[i] = h;
    }
}
