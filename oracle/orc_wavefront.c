/*
 * orc_wavefront.c — CPU oracle: the wavefront frame (Generate, Trace, Logic, Shade x4 + NEE, Shadow,
 * Accumulate) with SERIAL queue semantics.
 *
 * TEST INFRASTRUCTURE ONLY (see nexus_oracle.h).  Restates
 *   /root/reference/Nexus/src/Renderer/PathTracer.cpp:248-288          frame sequence
 *   /root/reference/Nexus/src/Cuda/PathTracer/PathTracer.cu:37-83      ToColorUInt, Tonemap, SampleBackground
 *   /root/reference/Nexus/src/Cuda/PathTracer/PathTracer.cu:85-122     GenerateKernel
 *   /root/reference/Nexus/src/Cuda/PathTracer/PathTracer.cu:136-210    LogicKernel
 *   /root/reference/Nexus/src/Cuda/PathTracer/PathTracer.cu:213-308    NextEventEstimation
 *   /root/reference/Nexus/src/Cuda/PathTracer/PathTracer.cu:311-458    Shade
 *   /root/reference/Nexus/src/Cuda/PathTracer/PathTracer.cu:480-496    AccumulateKernel
 * Serial queue semantics = threads of a kernel run in ascending index order, the material kernels in
 * graph insertion order Diffuse, Plastic, Dielectric, Conductor (PathTracer.cpp:116-120), every
 * atomicAdd slot is therefore handed out in that order.  The CUDA run itself is racy; this ordering is
 * the only reproducible one and is what NX_COMPACT_ORDERED reproduces on the GPU.
 */
#include <stdlib.h>
#include <pthread.h>
#include "orc_shade.h"

/* what one shaded path asks for: the shadow ray of its light sample, its continuation ray */
struct sh_out { int valid; f3 origin, direction, radiance; float distance; uint32_t pixel; };
struct tr_out { int valid; f3 origin, direction; uint32_t pixel; };

struct orc_wavefront {
    const orc_scene *scene;
    uint32_t n; /* local pixel count */
    uint32_t *pixelMap;
    int rngMode, conductorMode;
    /* D_PathStateSOA — PathTracer.cuh:19-30 */
    f3 *throughput, *radiance, *rayOrigin;
    float *lastPdf;
    /* D_TraceRequestSOA */
    f3 *trOrigin, *trDirection;
    nx_hit *trHit;
    uint32_t *trPixel;
    /* D_ShadowTraceRequestSOA */
    f3 *shOrigin, *shDirection, *shRadiance;
    float *shDistance;
    uint32_t *shPixel;
    /* D_MaterialRequestSOA x4 in enum order DIFFUSE, DIELECTRIC, PLASTIC, CONDUCTOR */
    f3 *mqDirection[4];
    nx_hit *mqHit[4];
    uint32_t *mqPixel[4];
    orc_queue_sizes q;
    f3 *accumulation;
    uint32_t *rgba8;
    orc_trace_stats closestStats, shadowStats;
    uint32_t frameNumber;
    /* threaded logic / shade (orc_wavefront_render with nthreads > 1): what item k of a queue produces is parked in slot k of
     * these and appended to the output queues afterwards, in item order — the serial slot order exactly */
    struct sh_out *tmpShadow;
    struct tr_out *tmpTrace;
    int8_t *tmpType;
};

static int32_t *mat_queue_size(orc_queue_sizes *q, int type)
{
    switch (type) {
    case NX_MAT_DIFFUSE: return q->diffuseSize;
    case NX_MAT_DIELECTRIC: return q->dielectricSize;
    case NX_MAT_PLASTIC: return q->plasticSize;
    default: return q->conductorSize;
    }
}

orc_wavefront *orc_wavefront_create(const orc_scene *scene, uint32_t localCount, const uint32_t *pixelMap, int rngMode,
                                    int conductorMode)
{
    orc_wavefront *w = (orc_wavefront *)calloc(1, sizeof *w);
    const size_t n = localCount;
    w->scene = scene;
    w->n = localCount;
    w->rngMode = rngMode;
    w->conductorMode = conductorMode;
    w->pixelMap = (uint32_t *)malloc(n * 4);
    for (uint32_t i = 0; i < localCount; i++) w->pixelMap[i] = pixelMap ? pixelMap[i] : i;
#define A3(x) w->x = (f3 *)calloc(n, sizeof(f3))
    A3(throughput); A3(radiance); A3(rayOrigin); A3(trOrigin); A3(trDirection); A3(shOrigin); A3(shDirection);
    A3(shRadiance); A3(accumulation);
#undef A3
    w->lastPdf = (float *)calloc(n, 4);
    w->trHit = (nx_hit *)calloc(n, sizeof(nx_hit));
    w->trPixel = (uint32_t *)calloc(n, 4);
    w->shDistance = (float *)calloc(n, 4);
    w->shPixel = (uint32_t *)calloc(n, 4);
    for (int m = 0; m < 4; m++) {
        w->mqDirection[m] = (f3 *)calloc(n, sizeof(f3));
        w->mqHit[m] = (nx_hit *)calloc(n, sizeof(nx_hit));
        w->mqPixel[m] = (uint32_t *)calloc(n, 4);
    }
    w->rgba8 = (uint32_t *)calloc(n, 4);
    return w;
}

void orc_wavefront_destroy(orc_wavefront *w)
{
    if (!w) return;
    free(w->pixelMap); free(w->throughput); free(w->radiance); free(w->rayOrigin); free(w->lastPdf);
    free(w->trOrigin); free(w->trDirection); free(w->trHit); free(w->trPixel);
    free(w->shOrigin); free(w->shDirection); free(w->shRadiance); free(w->shDistance); free(w->shPixel);
    for (int m = 0; m < 4; m++) { free(w->mqDirection[m]); free(w->mqHit[m]); free(w->mqPixel[m]); }
    free(w->accumulation); free(w->rgba8);
    free(w->tmpShadow); free(w->tmpTrace); free(w->tmpType);
    free(w);
}

/* ------------------------------------------------------------------------------------------------ */

/* SampleBackground — PathTracer.cu:65-83 */
static f3 sample_background(const orc_scene *s, f3 d)
{
    if (s->hdrMap) {
        const float theta = nxf_atan2f(d.z, d.x);
        const float phi = nxf_asinf(d.y);
        const float u = (float)((theta + ORC_PI) * ORC_INV_PI * 0.5);
        const float v = (float)(1.0f - (phi + ORC_PI * 0.5f) * ORC_INV_PI);
        const f4 c = orc_tex2d_f4(s->hdrMap, u, v);
        return mk3(c.x, c.y, c.z);
    }
    return scale3(ld3(s->settings.backgroundColor), s->settings.backgroundIntensity);
}

/* ---- environment importance sampling (extension; see nexus_oracle.h) -------------------------------------------- */

void orc_env_distribution(const nx_texture_desc *hdr, float *marginalCdf, float *rowCdf, float *density)
{
    const int W = (int)hdr->width, H = (int)hdr->height;
    double *rowSum = (double *)malloc(sizeof(double) * (size_t)H);
    double total = 0.0;
    for (int y = 0; y < H; y++) {
        const double sinTheta = sin(3.14159265358979323846 * ((double)y + 0.5) / (double)H);
        double run = 0.0;
        for (int x = 0; x < W; x++) {
            const uint8_t *t = hdr->rgba8 + 4 * ((size_t)y * (size_t)W + (size_t)x);
            const double lum = 0.2126 * (double)orc_srgb_to_linear(t[0]) + 0.7152 * (double)orc_srgb_to_linear(t[1]) + 0.0722 * (double)orc_srgb_to_linear(t[2]);
            const double wgt = lum * sinTheta + 1e-6;
            density[(size_t)y * (size_t)W + (size_t)x] = (float)wgt; /* scaled below */
            run += wgt;
            rowCdf[(size_t)y * (size_t)W + (size_t)x] = (float)run; /* normalised below */
        }
        rowSum[y] = run;
        total += run;
    }
    double run = 0.0;
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const size_t i = (size_t)y * (size_t)W + (size_t)x;
            rowCdf[i] = x == W - 1 ? 1.0f : (float)((double)rowCdf[i] / rowSum[y]);
            density[i] = (float)((double)density[i] / total * (double)W * (double)H / (2.0 * 3.14159265358979323846 * 3.14159265358979323846));
        }
        run += rowSum[y];
        marginalCdf[y] = y == H - 1 ? 1.0f : (float)(run / total);
    }
    free(rowSum);
}

/* (u, v) of a direction exactly as SampleBackground computes them, and the texel they fall in */
static void env_texel(const orc_scene *s, f3 d, int *x, int *y)
{
    const float theta = nxf_atan2f(d.z, d.x);
    const float phi = nxf_asinf(d.y);
    const float u = (float)((theta + ORC_PI) * ORC_INV_PI * 0.5);
    const float v = (float)(1.0f - (phi + ORC_PI * 0.5f) * ORC_INV_PI);
    const int W = (int)s->hdrMap->width, H = (int)s->hdrMap->height;
    int xi = (int)(u * (float)W), yi = (int)(v * (float)H);
    *x = xi < 0 ? 0 : (xi > W - 1 ? W - 1 : xi);
    *y = yi < 0 ? 0 : (yi > H - 1 ? H - 1 : yi);
}

/* pdf per solid angle of the environment sampler for direction d (unit length), light-selection probability excluded */
static float env_pdf(const orc_scene *s, f3 d)
{
    int x, y;
    env_texel(s, d, &x, &y);
    const float cosLat = sqrtf(fmaxf(1.0f - d.y * d.y, 1.0e-12f));
    return s->envDensity[(size_t)y * s->hdrMap->width + (size_t)x] / cosLat;
}

static int cdf_find(const float *cdf, int n, float r)  /* first index whose cdf exceeds r */
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] > r) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

static f3 env_sample(const orc_scene *s, float r1, float r2)
{
    const int W = (int)s->hdrMap->width, H = (int)s->hdrMap->height;
    const int y = cdf_find(s->envMarginalCdf, H, r1);
    const float ylo = y ? s->envMarginalCdf[y - 1] : 0.0f;
    const float fy = (r1 - ylo) / (s->envMarginalCdf[y] - ylo);
    const float *row = s->envRowCdf + (size_t)y * (size_t)W;
    const int x = cdf_find(row, W, r2);
    const float xlo = x ? row[x - 1] : 0.0f;
    const float fx = (r2 - xlo) / (row[x] - xlo);
    const float u = ((float)x + fx) / (float)W, v = ((float)y + fy) / (float)H;
    const float phi = (1.0f - v) * 3.14159265f - 1.57079633f, theta = u * 6.28318531f - 3.14159265f;
    const float c = nxf_cosf(phi);
    return mk3(c * nxf_cosf(theta), nxf_sinf(phi), c * nxf_sinf(theta));
}

/* lights the NEE chooses among: the mesh lights, plus the environment when it is importance sampled */
static uint32_t nee_light_count(const orc_scene *s) { return s->lightCount + (s->envSampling && s->hdrMap ? 1u : 0u); }

/* GenerateKernel — PathTracer.cu:85-122 */
static void generate(orc_wavefront *w)
{
    const nx_camera *cam = &w->scene->camera;
    const uint32_t resX = cam->resolution[0], resY = cam->resolution[1];
    for (uint32_t index = 0; index < w->n; index++) {
        const uint32_t g = w->pixelMap[index];
        const uint32_t j = g / resX;
        const uint32_t i = g - j * resX;
        uint32_t rng = orc_rng_init_pixel(i, j, resX, w->frameNumber);
        const float x = ((float)i + orc_rand(&rng)) / (float)resX;
        const float y = ((float)j + orc_rand(&rng)) / (float)resY;
        const f2 disk = orc_unit_disk(&rng);
        const f2 rd = {cam->lensRadius * disk.x, cam->lensRadius * disk.y};
        const f3 offset = add3(scale3(ld3(cam->right), rd.x), scale3(ld3(cam->up), rd.y));
        const f3 origin = add3(ld3(cam->position), offset);
        const f3 target = sub3(sub3(add3(add3(ld3(cam->lowerLeftCorner), scale3(ld3(cam->viewportX), x)),
                                         scale3(ld3(cam->viewportY), y)), ld3(cam->position)), offset);
        const f3 direction = normalize3(target);
        w->rayOrigin[index] = origin;
        w->lastPdf[index] = 1.0e10f;
        /* A deliberate definition where the reference leaves a value undefined: pathState.radiance[pixel] is only ever WRITTEN by
         * the bounce-1 logic (miss) or material kernel (hit) — PathTracer.cu:155-158, 387-390 — so a path whose first hit lands in
         * a queue without a kernel (a CONDUCTOR: the kernel body is commented out, PathTracer.cu:475-478) keeps whatever the
         * previous frame left there (uninitialised memory in frame 1) and AccumulateKernel averages that in.  Device and oracle
         * both start every frame's path radiance at zero: such a path contributes nothing.  (Found in round 4 when frames became
         * comparable bit for bit: 0.17 % of the material zoo's pixels in the reference-conductor mode.) */
        w->radiance[index] = mk3s(0.0f);
        w->trOrigin[index] = origin;
        w->trDirection[index] = direction;
        w->trPixel[index] = index;
    }
    w->q.traceSize[0] = (int32_t)w->n;
}

typedef struct { orc_wavefront *w; uint32_t begin, end; int shadow; orc_trace_stats st; } tr_job;

static void trace_range(tr_job *j)
{
    orc_wavefront *w = j->w;
    for (uint32_t i = j->begin; i < j->end; i++) {
        if (!j->shadow) {
            orc_trace_one(w->scene, (const float *)&w->trOrigin[i], (const float *)&w->trDirection[i], 0, 0.0f, &w->trHit[i], &j->st);
        } else {
            /* BVH8TraceShadow tail: unoccluded => pathRadiance[pixelIdx] += radiance (BVH8Traversal.cuh:515-516) */
            if (!orc_trace_one(w->scene, (const float *)&w->shOrigin[i], (const float *)&w->shDirection[i], 1, w->shDistance[i], NULL, &j->st)) {
                const uint32_t p = w->shPixel[i];
                w->radiance[p] = add3(w->radiance[p], w->shRadiance[i]);
            }
        }
    }
}
static void *trace_worker(void *a) { trace_range((tr_job *)a); return NULL; }

static void merge_stats(orc_trace_stats *dst, const orc_trace_stats *s)
{
    dst->rays += s->rays; dst->nodes += s->nodes; dst->tris += s->tris; dst->instances += s->instances;
    if (s->maxStack > dst->maxStack) dst->maxStack = s->maxStack;
}

static void trace_pass(orc_wavefront *w, uint32_t count, int shadow, int nthreads)
{
    orc_trace_stats *dst = shadow ? &w->shadowStats : &w->closestStats;
    if (nthreads <= 1 || count < 4096) {
        tr_job j;
        memset(&j, 0, sizeof j);
        j.w = w; j.begin = 0; j.end = count; j.shadow = shadow;
        trace_range(&j);
        merge_stats(dst, &j.st);
        return;
    }
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    tr_job jobs[256];
    const uint32_t chunk = (count + (uint32_t)nthreads - 1) / (uint32_t)nthreads;
    int started = 0;
    for (int t = 0; t < nthreads; t++) {
        const uint32_t b = (uint32_t)t * chunk;
        if (b >= count) break;
        memset(&jobs[t], 0, sizeof jobs[t]);
        jobs[t].w = w; jobs[t].begin = b; jobs[t].end = (b + chunk < count) ? b + chunk : count; jobs[t].shadow = shadow;
        pthread_create(&th[t], NULL, trace_worker, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; t++) { pthread_join(th[t], NULL); merge_stats(dst, &jobs[t].st); }
}

static uint32_t seed_for(const orc_wavefront *w, uint32_t slot, uint32_t pixelIdx, uint32_t bounce, uint32_t stage)
{
    if (w->rngMode == NX_RNG_PIXEL_KEYED) return orc_rng_init_keyed(w->pixelMap[pixelIdx], bounce, w->frameNumber, stage);
    return orc_rng_init_index(slot, w->scene->camera.resolution[0], w->frameNumber);
}

/* Ranges of a queue on worker threads (pthread per range, as trace_pass).  fn(ctx, begin, end). */
typedef struct { void (*fn)(void *, uint32_t, uint32_t); void *ctx; uint32_t begin, end; } range_job;
static void *range_worker(void *a) { range_job *j = (range_job *)a; j->fn(j->ctx, j->begin, j->end); return NULL; }
static void parallel_ranges(uint32_t count, int nthreads, void (*fn)(void *, uint32_t, uint32_t), void *ctx)
{
    if (nthreads <= 1 || count < 4096) { fn(ctx, 0, count); return; }
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    range_job jobs[256];
    const uint32_t chunk = (count + (uint32_t)nthreads - 1) / (uint32_t)nthreads;
    int started = 0;
    for (int t = 0; t < nthreads; t++) {
        const uint32_t b = (uint32_t)t * chunk;
        if (b >= count) break;
        jobs[t].fn = fn; jobs[t].ctx = ctx; jobs[t].begin = b; jobs[t].end = (b + chunk < count) ? b + chunk : count;
        pthread_create(&th[t], NULL, range_worker, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}

/* LogicKernel — PathTracer.cu:136-210: one thread (queue item).  Returns the material queue the path goes to, -1: none. */
static int logic_item(orc_wavefront *w, uint32_t bounce, uint32_t index)
{
    const orc_scene *s = w->scene;
    const nx_hit hit = w->trHit[index];
    const f3 dir = w->trDirection[index];
    const uint32_t pixelIdx = w->trPixel[index];
    uint32_t rng = seed_for(w, index, pixelIdx, bounce, 0);
    const f3 throughput = bounce == 1 ? mk3s(1.0f) : w->throughput[pixelIdx];

    if (hit.hitDistance == 1e30f) {
        f3 bg = mul3(throughput, sample_background(s, dir));
        if (s->envSampling && s->hdrMap && bounce > 1 && s->settings.useMIS) {
            /* the NEE samples the environment too: weight the BSDF-sampled miss against it (extension) */
            const float envPdf = env_pdf(s, dir) / (float)nee_light_count(s);
            if (orc_pdf_valid(envPdf)) bg = scale3(bg, orc_power_heuristic(w->lastPdf[pixelIdx], envPdf));
        }
        if (bounce == 1) w->radiance[pixelIdx] = bg;
        else w->radiance[pixelIdx] = add3(w->radiance[pixelIdx], bg);
        return -1;
    }
    /* Russian roulette */
    const float probability = maxcomp3(throughput);
    if (orc_rand(&rng) < probability) w->throughput[pixelIdx] = div3s(throughput, probability);
    else return -1;

    const nx_bvh_instance *inst = &s->instances[hit.instanceIdx];
    const int type = s->materials[inst->materialId].type;
    if (type < 0 || type > 3) return -1;
    return type;
}

typedef struct { orc_wavefront *w; uint32_t bounce; int type; } item_ctx;
static void logic_range(void *a, uint32_t begin, uint32_t end)
{
    item_ctx *c = (item_ctx *)a;
    for (uint32_t i = begin; i < end; i++) c->w->tmpType[i] = (int8_t)logic_item(c->w, c->bounce, i);
}

/* The kernel: items in ascending order claim their queue slots (serial semantics).  With threads the per-item work — which
 * touches only the item's own pixel — runs on ranges first and the slots are handed out afterwards, in the same order. */
static void logic(orc_wavefront *w, uint32_t bounce, int nthreads)
{
    const uint32_t count = (uint32_t)w->q.traceSize[bounce - 1];
    const int threaded = nthreads > 1 && count >= 4096;
    if (threaded) {
        if (!w->tmpType) w->tmpType = (int8_t *)malloc(w->n);
        item_ctx c = {w, bounce, 0};
        parallel_ranges(count, nthreads, logic_range, &c);
    }
    for (uint32_t index = 0; index < count; index++) {
        const int type = threaded ? w->tmpType[index] : logic_item(w, bounce, index);
        if (type < 0) continue;
        int32_t *size = mat_queue_size(&w->q, type);
        const int32_t slot = size[bounce]++;
        w->mqHit[type][slot] = w->trHit[index];
        w->mqDirection[type][slot] = w->trDirection[index];
        w->mqPixel[type][slot] = w->trPixel[index];
    }
}

static float tri_area(f3 p0, f3 p1, f3 p2) { return 0.5f * length3(cross3(sub3(p1, p0), sub3(p2, p0))); }
static f3 tri_normal(const nx_triangle *t) { return cross3(sub3(ld3(t->pos1), ld3(t->pos0)), sub3(ld3(t->pos2), ld3(t->pos0))); }

/* NextEventEstimation — PathTracer.cu:213-308 */
static void nee(orc_wavefront *w, f3 wi, const nx_material *material, f3 hitPoint, f3 normal,
                f3 hitGNormal, f3 throughput, uint32_t pixelIdx, uint32_t *rng, struct sh_out *out)
{
    const orc_scene *s = w->scene;
    /* no lights: the reference indexes an empty array here (undefined); defined as "no light sample, no random numbers
     * drawn", identically in the device code */
    const uint32_t nLights = nee_light_count(s);
    if (nLights == 0u) return;
    const uint32_t pick = orc_uniform(nLights, rng);
    if (pick >= s->lightCount) {
        /* the environment (extension): direction from the map's luminance distribution, shadow ray to infinity */
        const float r1 = orc_rand(rng), r2 = orc_rand(rng);
        const f3 shDir = env_sample(s, r1, r2);
        const float lightPdf = env_pdf(s, shDir) / (float)nLights;
        if (!orc_pdf_valid(lightPdf)) return;
        const f4 q = rotation_to_z(normal);
        const f3 wo = rotate_point(q, shDir);
        f3 sampleThroughput;
        float bsdfPdf;
        if (!orc_bsdf_eval_f3(material, wi, wo, &sampleThroughput, &bsdfPdf)) return;
        const float weight = orc_power_heuristic(lightPdf, bsdfPdf);
        const f3 radiance = div3s(mul3(mul3(scale3(throughput, weight), sampleThroughput), sample_background(s, shDir)), lightPdf);
        out->valid = 1;
        out->distance = 1e30f;
        out->radiance = radiance;
        out->origin = offset_ray(hitPoint, scale3(hitGNormal, sgnE(dot3(shDir, normal))));
        out->direction = shDir;
        out->pixel = pixelIdx;
        return;
    }
    const nx_light light = s->lights[pick];
    if (light.type != NX_LIGHT_MESH) return;

    const nx_bvh_instance *inst = &s->instances[light.mesh.meshId];
    const orc_blas *bvh = &s->blas[inst->bvhIdx];
    const uint32_t triangleIdx = orc_uniform(bvh->triCount, rng);
    const f2 uv = orc_uniform_triangle(rng);
    const nx_triangle *tri = &bvh->tris[triangleIdx];

    f3 p = bary3(ld3(tri->pos0), ld3(tri->pos1), ld3(tri->pos2), uv.x, uv.y);
    p = mat_point(&inst->transform, p);
    const f3 lightGNormal = normalize3(mat_vec_transposed(&inst->invTransform, tri_normal(tri)));
    f3 lightNormal = bary3(ld3(tri->normal0), ld3(tri->normal1), ld3(tri->normal2), uv.x, uv.y);
    lightNormal = normalize3(mat_vec_transposed(&inst->invTransform, lightNormal));

    f3 toLight = sub3(p, hitPoint);
    float offsetDirection = sgnE(dot3(toLight, normal));
    const f3 shOrigin = offset_ray(hitPoint, scale3(hitGNormal, offsetDirection));
    offsetDirection = sgnE(dot3(neg3(toLight), lightNormal));
    p = offset_ray(p, scale3(lightGNormal, offsetDirection));

    toLight = sub3(p, shOrigin);
    const float distance = length3(toLight);
    const f3 shDir = div3s(toLight, distance);

    const f4 q = rotation_to_z(normal);
    const f3 wo = rotate_point(q, shDir);
    const float cosThetaO = fabsf(dot3(lightNormal, shDir));
    const float dSquared = dot3(toLight, toLight);
    const float area = tri_area(mat_point(&inst->transform, ld3(tri->pos0)), mat_point(&inst->transform, ld3(tri->pos1)),
                                mat_point(&inst->transform, ld3(tri->pos2)));
    float lightPdf = 1.0f / ((float)(nLights * bvh->triCount) * area);
    lightPdf *= dSquared / cosThetaO;
    if (!orc_pdf_valid(lightPdf)) return;

    const nx_material *lightMaterial = &s->materials[inst->materialId];
    f3 sampleThroughput;
    float bsdfPdf;
    if (!orc_bsdf_eval_f3(material, wi, wo, &sampleThroughput, &bsdfPdf)) return;
    const float weight = orc_power_heuristic(lightPdf, bsdfPdf);

    f3 emissive;
    if (lightMaterial->emissiveMapId != -1) {
        const f2 t = bary2(tri->texCoord0, tri->texCoord1, tri->texCoord2, uv.x, uv.y);
        const f4 c = orc_tex2d_f4(&s->emissiveMaps[lightMaterial->emissiveMapId], t.x, t.y);
        emissive = mk3(c.x, c.y, c.z);
    } else emissive = ld3(lightMaterial->emissive);

    const f3 radiance = div3s(scale3(mul3(mul3(scale3(throughput, weight), sampleThroughput), emissive), lightMaterial->intensity), lightPdf);
    out->valid = 1;
    out->distance = distance;
    out->radiance = radiance;
    out->origin = shOrigin;
    out->direction = shDir;
    out->pixel = pixelIdx;
}

/* Shade<BSDF> — PathTracer.cu:311-458.  `type` selects the queue and the BSDF. */
static void shade_item(orc_wavefront *w, uint32_t bounce, int type, int32_t requestIdx, struct sh_out *shOut, struct tr_out *trOut)
{
    const orc_scene *s = w->scene;
    shOut->valid = 0;
    trOut->valid = 0;
    {
        const nx_hit hit = w->mqHit[type][requestIdx];
        const f3 rayDirection = w->mqDirection[type][requestIdx];
        const uint32_t pixelIdx = w->mqPixel[type][requestIdx];
        f3 throughput = bounce == 1 ? mk3s(1.0f) : w->throughput[pixelIdx];
        uint32_t rng = seed_for(w, (uint32_t)requestIdx, pixelIdx, bounce, 1);

        const nx_bvh_instance *inst = &s->instances[hit.instanceIdx];
        const orc_blas *bvh = &s->blas[inst->bvhIdx];
        const nx_triangle *tri = &bvh->tris[hit.triIdx];
        nx_material material = s->materials[inst->materialId];
        material.type = (int8_t)type;

        f3 p = bary3(ld3(tri->pos0), ld3(tri->pos1), ld3(tri->pos2), hit.u, hit.v);
        p = mat_point(&inst->transform, p);
        f3 normal = bary3(ld3(tri->normal0), ld3(tri->normal1), ld3(tri->normal2), hit.u, hit.v);
        const f2 texUv = bary2(tri->texCoord0, tri->texCoord1, tri->texCoord2, hit.u, hit.v);
        normal = normalize3(mat_vec_transposed(&inst->invTransform, normal));
        f3 gNormal = normalize3(mat_vec_transposed(&inst->invTransform, tri_normal(tri)));

        if (material.emissiveMapId != -1) {
            const f4 c = orc_tex2d_f4(&s->emissiveMaps[material.emissiveMapId], texUv.x, texUv.y);
            material.emissive[0] = c.x; material.emissive[1] = c.y; material.emissive[2] = c.z;
        }
        const int allowMIS = bounce > 1 && s->settings.useMIS;
        f3 radiance = mk3s(0.0f);
        const f3 emissive = ld3(material.emissive);
        if (maxcomp3(scale3(emissive, material.intensity)) > 0.0f) {
            float weight = 1.0f;
            if (allowMIS) {
                const float lastPdf = w->lastPdf[pixelIdx];
                const float cosThetaO = fabsf(dot3(normal, rayDirection));
                const float dSquared = squaref(length3(sub3(p, w->rayOrigin[pixelIdx])));
                const float area = tri_area(mat_point(&inst->transform, ld3(tri->pos0)), mat_point(&inst->transform, ld3(tri->pos1)),
                                            mat_point(&inst->transform, ld3(tri->pos2)));
                float lightPdf = 1.0f / ((float)(nee_light_count(s) * bvh->triCount) * area);
                lightPdf *= dSquared / cosThetaO;
                if (!orc_pdf_valid(lightPdf)) weight = 0.0f;
                else weight = orc_power_heuristic(lastPdf, lightPdf);
            }
            radiance = mul3(scale3(scale3(emissive, weight), material.intensity), throughput);
        }
        if (bounce == 1) w->radiance[pixelIdx] = radiance;
        else w->radiance[pixelIdx] = add3(w->radiance[pixelIdx], radiance);

        if (bounce == s->settings.pathLength) return;

        f4 color = {1.0f, 1.0f, 1.0f, 1.0f};
        if (material.diffuseMapId != -1) {
            color = orc_tex2d_f4(&s->diffuseMaps[material.diffuseMapId], texUv.x, texUv.y);
            material.diffuse.albedo[0] = color.x; material.diffuse.albedo[1] = color.y; material.diffuse.albedo[2] = color.z;
        }
        if (dot3(gNormal, rayDirection) > 0.0f && type != NX_MAT_DIELECTRIC) { normal = neg3(normal); gNormal = neg3(gNormal); }

        const f4 q = rotation_to_z(normal);
        const f3 wi = rotate_point(q, neg3(rayDirection));
        f3 wo;

        if (orc_rand(&rng) > material.opacity || (material.diffuseMapId != -1 && orc_rand(&rng) > color.w)) {
            wo = normalize3(rotate_point(invert_rotation(q), neg3(wi)));
            const float od = sgnE(dot3(wo, normal));
            const f3 origin = offset_ray(p, scale3(gNormal, od));
            trOut->valid = 1;
            trOut->origin = origin;
            trOut->direction = wo;
            trOut->pixel = pixelIdx;
        } else {
            if (s->settings.useMIS) nee(w, wi, &material, p, normal, gNormal, throughput, pixelIdx, &rng, shOut);
            float pdf;
            f3 sampleThroughput;
            if (!orc_bsdf_sample_f3(&material, wi, &rng, &wo, &sampleThroughput, &pdf)) return;
            wo = normalize3(rotate_point(invert_rotation(q), wo));
            const float od = sgnE(dot3(wo, normal));
            const f3 origin = offset_ray(p, scale3(gNormal, od));
            throughput = mul3(throughput, sampleThroughput);
            trOut->valid = 1;
            trOut->origin = origin;
            trOut->direction = wo;
            trOut->pixel = pixelIdx;
            w->rayOrigin[pixelIdx] = origin;
            w->throughput[pixelIdx] = throughput;
            w->lastPdf[pixelIdx] = pdf;
        }
    }
}

static void shade_range(void *a, uint32_t begin, uint32_t end)
{
    item_ctx *c = (item_ctx *)a;
    for (uint32_t i = begin; i < end; i++) shade_item(c->w, c->bounce, c->type, (int32_t)i, &c->w->tmpShadow[i], &c->w->tmpTrace[i]);
}

/* The kernel: a path's shadow request is appended before its continuation ray, paths in ascending order (see logic) */
static void shade(orc_wavefront *w, uint32_t bounce, int type, int nthreads)
{
    const int32_t size = mat_queue_size(&w->q, type)[bounce];
    const int threaded = nthreads > 1 && size >= 4096;
    if (threaded) {
        if (!w->tmpShadow) w->tmpShadow = (struct sh_out *)malloc((size_t)w->n * sizeof(struct sh_out));
        if (!w->tmpTrace) w->tmpTrace = (struct tr_out *)malloc((size_t)w->n * sizeof(struct tr_out));
        item_ctx c = {w, bounce, type};
        parallel_ranges((uint32_t)size, nthreads, shade_range, &c);
    }
    for (int32_t requestIdx = 0; requestIdx < size; requestIdx++) {
        struct sh_out shLocal;
        struct tr_out trLocal;
        const struct sh_out *sh = &shLocal;
        const struct tr_out *tr = &trLocal;
        if (threaded) { sh = &w->tmpShadow[requestIdx]; tr = &w->tmpTrace[requestIdx]; }
        else shade_item(w, bounce, type, requestIdx, &shLocal, &trLocal);
        if (sh->valid) {
            const int32_t slot = w->q.traceShadowSize[bounce]++;
            w->shDistance[slot] = sh->distance;
            w->shRadiance[slot] = sh->radiance;
            w->shOrigin[slot] = sh->origin;
            w->shDirection[slot] = sh->direction;
            w->shPixel[slot] = sh->pixel;
        }
        if (tr->valid) {
            const int32_t slot = w->q.traceSize[bounce]++;
            w->trOrigin[slot] = tr->origin;
            w->trDirection[slot] = tr->direction;
            w->trPixel[slot] = tr->pixel;
        }
    }
}

void orc_wavefront_render(orc_wavefront *w, uint32_t frameNumber, int nthreads)
{
    const uint32_t pathLength = w->scene->settings.pathLength;
    w->frameNumber = frameNumber;
    memset(&w->q, 0, sizeof w->q);
    generate(w);
    trace_pass(w, (uint32_t)w->q.traceSize[0], 0, nthreads);
    for (uint32_t bounce = 1; bounce <= pathLength && bounce < NX_PATH_MAX_LENGTH; bounce++) {
        logic(w, bounce, nthreads);
        /* graph insertion order: Diffuse, Plastic, Dielectric, Conductor (PathTracer.cpp:116-120) */
        shade(w, bounce, NX_MAT_DIFFUSE, nthreads);
        shade(w, bounce, NX_MAT_PLASTIC, nthreads);
        shade(w, bounce, NX_MAT_DIELECTRIC, nthreads);
        if (w->conductorMode == NX_CONDUCTOR_EXTENDED) shade(w, bounce, NX_MAT_CONDUCTOR, nthreads);
        trace_pass(w, (uint32_t)w->q.traceSize[bounce], 0, nthreads);
        trace_pass(w, (uint32_t)w->q.traceShadowSize[bounce], 1, nthreads);
    }
}

/* Tonemap + LinearToGamma + ToColorUInt — PathTracer.cu:37-62, Utils/Utils.h:51-54 */
uint32_t orc_tonemap_rgba8(const float rgb[3])
{
    uint32_t out = 0;
    for (int c = 0; c < 3; c++) {
        float x = rgb[c] * 0.6f;
        x = clampf((x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f), 0.0f, 1.0f);
        x = (float)nxf_pow((double)x, 0.45454545454);
        x = clampf(x, 0.0f, 1.0f);
        out |= (uint32_t)(uint8_t)(x * 255.0f) << (8 * c);
    }
    return out | (255u << 24);
}

/* AccumulateKernel — PathTracer.cu:480-496 */
void orc_wavefront_accumulate(orc_wavefront *w, uint32_t frameNumber)
{
    for (uint32_t i = 0; i < w->n; i++) {
        if (frameNumber == 1) w->accumulation[i] = w->radiance[i];
        else w->accumulation[i] = add3(w->accumulation[i], div3s(sub3(w->radiance[i], w->accumulation[i]), (float)frameNumber));
        w->rgba8[i] = orc_tonemap_rgba8((const float *)&w->accumulation[i]);
    }
}

const float *orc_wavefront_radiance(const orc_wavefront *w) { return &w->radiance[0].x; }
const float *orc_wavefront_accumulation(const orc_wavefront *w) { return &w->accumulation[0].x; }
const uint32_t *orc_wavefront_rgba8(const orc_wavefront *w) { return w->rgba8; }
const orc_queue_sizes *orc_wavefront_queue_sizes(const orc_wavefront *w) { return &w->q; }
void orc_wavefront_trace_stats(const orc_wavefront *w, orc_trace_stats *closest, orc_trace_stats *shadow)
{
    if (closest) *closest = w->closestStats;
    if (shadow) *shadow = w->shadowStats;
}
