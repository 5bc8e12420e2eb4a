/*
 * nexus_oracle.h — public interface of the CPU oracle (liboracle.so, loaded with ctypes by tests/).
 *
 * TEST INFRASTRUCTURE ONLY — see oracle/README.md.  Nothing under nexus_amd/ may include, link, import
 * or execute anything from oracle/.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / reported baseline.
 *
 * PARITY STATUS: "parity unpinned" against reference *outputs*.  The reference
 * (/root/reference, Patoche692/Nexus @ 2024-10-08) ships no tests, golden vectors or fixtures for
 * this path, and it cannot be compiled in this image without stand-ins (it needs the CUDA toolkit
 * headers, glm and assimp, all absent; writing stand-in headers is not allowed).  The oracle is
 * therefore a line-by-line restatement of the reference's algorithm, each function citing the
 * reference file:line it follows, and is pinned only by algorithm-independent ground truth:
 * brute-force ray/triangle intersection, analytic BSDF identities, structural BVH invariants and
 * known-answer RNG values computed by hand from the published Jenkins / xorshift definitions.
 */
#ifndef NEXUS_ORACLE_H
#define NEXUS_ORACLE_H

#include <stdint.h>
#include "../include/nexus_pod.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- builders (orc_build.c) */

/* BVH2 node, same layout as the reference's BVH2Node (Geometry/BVH/BVH.h:17-37), 32 bytes. */
typedef struct orc_bvh2_node {
    float aabbMin[3], aabbMax[3];
    uint32_t leftFirst; /* left child index, or first index into triIdx for a leaf */
    uint32_t triCount;  /* 0 for inner nodes */
} orc_bvh2_node;

typedef struct orc_bvh2 {
    orc_bvh2_node *nodes;
    uint32_t nodeCount;
    uint32_t *triIdx;
    uint32_t triCount;
} orc_bvh2;

typedef struct orc_bvh8 {
    nx_bvh8_node *nodes;
    uint32_t nodeCount;
    uint32_t *primIdx; /* triangle indices (BLAS) or instance indices (TLAS) in leaf order */
    uint32_t primCount;
} orc_bvh8;

/* Binned-SAH BVH2, one triangle per leaf.  Geometry/BVH/BVH.cpp:13-210. */
int orc_bvh2_build(const nx_triangle *tris, uint32_t n, orc_bvh2 *out);
void orc_bvh2_free(orc_bvh2 *b);

/* BVH2 -> BVH8 collapse (Ylitie 2017 DP).  Geometry/BVH/BVH8Builder.cpp:7-393.
 * clamp_qhi = 0 mirrors the reference BLAS path (u8 cast wraps), 1 clamps ceil() to 255 like the
 * reference TLAS path (TLASBuilder.cpp:314-316). */
int orc_bvh8_build(const nx_triangle *tris, uint32_t n, int clamp_qhi, orc_bvh8 *out);
void orc_bvh8_free(orc_bvh8 *b);

/* Instance: transform / inverse / world bounds from the BLAS root quantisation frame.
 * Geometry/BVH/BVHInstance.cpp:4-29, Math/Mat4.h:151-194. */
void orc_mat4_identity(nx_mat4 *m);
void orc_mat4_mul(const nx_mat4 *a, const nx_mat4 *b, nx_mat4 *out);
void orc_mat4_invert(const nx_mat4 *m, nx_mat4 *out);
void orc_mat4_from_trs(const float pos[3], const float rotDeg[3], const float scale[3], nx_mat4 *out);
void orc_instance_init(nx_bvh_instance *inst, uint32_t bvhIdx, int32_t materialId, const nx_mat4 *transform,
                       const nx_bvh8_node *blasRoot);

/* TLAS: agglomerative BVH2 over instance bounds, then the same DP collapse.
 * Geometry/BVH/TLAS.cpp:13-91, TLASBuilder.cpp:5-370. */
int orc_tlas_build(const nx_bvh_instance *instances, uint32_t n, orc_bvh8 *out);

/* Camera::ToDevice, Scene/Camera.cpp:142-168. */
void orc_camera_init(nx_camera *cam, const float position[3], const float forward[3], float horizontalFovDeg,
                     uint32_t width, uint32_t height, float focusDist, float defocusAngleDeg);

/* ---------------------------------------------------------------- scene + traversal (orc_trace.c) */

typedef struct orc_blas {
    const nx_bvh8_node *nodes;
    const nx_triangle *tris;
    const uint32_t *triIdx;
    uint32_t nodeCount, triCount;
} orc_blas;

typedef struct orc_scene {
    const nx_bvh8_node *tlasNodes;
    const uint32_t *tlasInstIdx;
    uint32_t tlasNodeCount;
    const nx_bvh_instance *instances;
    uint32_t instanceCount;
    const orc_blas *blas; /* indexed by nx_bvh_instance.bvhIdx */
    uint32_t blasCount;
    const nx_material *materials;
    uint32_t materialCount;
    const nx_light *lights;
    uint32_t lightCount;
    const nx_texture_desc *diffuseMaps;
    const nx_texture_desc *emissiveMaps;
    const nx_texture_desc *hdrMap; /* NULL: flat background */
    nx_camera camera;
    nx_render_settings settings;
    /* Extension (no counterpart in the reference, which adds the environment on a miss only, PathTracer.cu:152-164):
     * importance sampling of the environment map in NEE + MIS on a miss — include/nexus_hip.h nxhip_set_env_sampling.
     * The three tables are filled by orc_env_distribution. */
    int32_t envSampling;
    const float *envMarginalCdf; /* [height] */
    const float *envRowCdf;      /* [height][width] */
    const float *envDensity;     /* [height][width]: pdf per solid angle x cos(latitude) */
} orc_scene;

/* Piecewise-constant sampling distribution of an equirectangular map: texel weight = luminance of the sRGB-decoded texel
 * x sin(polar angle of the row) + 1e-6, accumulated in double; cdfs as float with a final 1; density = weight / total
 * x width x height / (2 pi^2). */
void orc_env_distribution(const nx_texture_desc *hdr, float *marginalCdf, float *rowCdf, float *density);

/* Visit counters for the roofline's algorithmic bytes. */
typedef struct orc_trace_stats {
    uint64_t rays, nodes, tris, instances, maxStack;
} orc_trace_stats;

/* Closest hit through TLAS -> BLAS.  Cuda/BVH/BVH8Traversal.cuh:55-322 with serial semantics
 * (__activemask() == all lanes, so triangle postponing never triggers). */
void orc_trace_closest(const orc_scene *s, const nx_ray *rays, uint32_t n, nx_hit *hits, orc_trace_stats *stats);
/* Any hit within tmax[i].  occluded[i] = 1 if something was hit.  BVH8Traversal.cuh:326-518. */
void orc_trace_any(const orc_scene *s, const nx_ray *rays, const float *tmax, uint32_t n, uint8_t *occluded,
                   orc_trace_stats *stats);
/* Same as orc_trace_closest, rays split over nthreads pthreads (cpu_baseline leg). */
/* Step log for the lane-scheduling simulator (tools/lane_sim.py): while set (calling thread only), every traced ray appends the
 * kinds of the records it visits — 1 node, 2 triangle, 3 instance entry — and a closing 0. */
void orc_trace_set_step_log(uint8_t *buf, uint64_t cap);
uint64_t orc_trace_step_log_length(void);
void orc_trace_closest_mt(const orc_scene *s, const nx_ray *rays, uint32_t n, nx_hit *hits, int nthreads);

/* Ground truth: every instance x every triangle with the reference's Moeller-Trumbore
 * (Cuda/Geometry/Triangle.cuh:53-86), instance order ascending, first-found wins ties. */
void orc_brute_closest(const orc_scene *s, const nx_ray *rays, uint32_t n, nx_hit *hits);
void orc_brute_any(const orc_scene *s, const nx_ray *rays, const float *tmax, uint32_t n, uint8_t *occluded);

/* "CPU BVH2 intersect reference path" of BASELINE.json configs[0]: ordered two-child descent over a
 * BVH2 of ONE mesh in object space (the reference's BVH2Traversal.cuh:7-52 is dead code that does
 * not compile; this follows its algorithm). */
/* diagnostic logs of the closest-hit traversal (tools/lane_sim.py, tools/entry_point_probe.py): see orc_trace.c */
void orc_trace_set_node_log(uint64_t *buf, uint64_t capWords);
uint64_t orc_trace_node_log_length(void);
void orc_bvh2_trace_closest(const orc_bvh2 *b, const nx_triangle *tris, const nx_ray *rays, uint32_t n, nx_hit *hits);

/* Decode one node against one ray: returns the two stack entries of ChildTrace. */
void orc_child_trace(const nx_bvh8_node *node, const float origin[3], const float direction[3], float tmax,
                     uint32_t out_entries[4]);

/* ---------------------------------------------------------------- RNG / sampling / BSDF (orc_shade.c) */

uint32_t orc_jenkins(uint32_t x);
uint32_t orc_rng_init_pixel(uint32_t px, uint32_t py, uint32_t resX, uint32_t frame);
uint32_t orc_rng_init_index(uint32_t index, uint32_t resX, uint32_t frame);
uint32_t orc_rng_init_keyed(uint32_t globalPixel, uint32_t bounce, uint32_t frame, uint32_t stage);
float orc_rand(uint32_t *state);

/* BSDF sample / eval in the local frame (z = normal).  type = NX_MAT_*.  Return 1 if valid.
 * Cuda/BSDF/ (all .cuh files).  conductor Eval exists only as an extension. */
int orc_bsdf_sample(const nx_material *m, const float wi[3], uint32_t *rng, float wo[3], float throughput[3], float *pdf);
int orc_bsdf_eval(const nx_material *m, const float wi[3], const float wo[3], float throughput[3], float *pdf);
/* ... over arrays of the C-ABI hooks' records (include/nexus_pod.h nx_bsdf_query / nx_bsdf_result) */
void orc_bsdf_sample_batch(const nx_material *m, const nx_bsdf_query *q, uint32_t n, nx_bsdf_result *r);
void orc_bsdf_eval_batch(const nx_material *m, const nx_bsdf_query *q, uint32_t n, nx_bsdf_result *r);

/* Software stand-in for tex2D<float4> on an sRGB, wrap, bilinear, normalised-coordinate texture. */
void orc_tex2d(const nx_texture_desc *t, float u, float v, float out[4]);

/* The shared transcendental functions (include/nexus_fmath.h) on arrays: out[i] = nxf_apply(op, a[i], b[i]); op = NXF_OP_*;
 * b may be NULL for the one-argument functions.  tests/test_fmath.py holds them against a 50-digit reference (CPU) and
 * against the device's nxhip_fmath_batch bit for bit (GPU). */
void orc_fmath_batch(int op, const double *a, const double *b, uint32_t n, double *out);

/* ---------------------------------------------------------------- wavefront (orc_wavefront.c) */

typedef struct orc_queue_sizes {
    int32_t traceSize[NX_PATH_MAX_LENGTH];
    int32_t traceShadowSize[NX_PATH_MAX_LENGTH];
    int32_t diffuseSize[NX_PATH_MAX_LENGTH];
    int32_t plasticSize[NX_PATH_MAX_LENGTH];
    int32_t dielectricSize[NX_PATH_MAX_LENGTH];
    int32_t conductorSize[NX_PATH_MAX_LENGTH];
} orc_queue_sizes;

typedef struct orc_wavefront orc_wavefront;

/* localCount pixels are rendered; pixelMap[i] = global pixel index of local pixel i (NULL: identity,
 * localCount must then be width*height).  rngMode / conductorMode as in nexus_pod.h. */
orc_wavefront *orc_wavefront_create(const orc_scene *scene, uint32_t localCount, const uint32_t *pixelMap,
                                    int rngMode, int conductorMode);
void orc_wavefront_destroy(orc_wavefront *w);
/* One frame: Generate, Trace, (Logic, Shade x4, Trace, Shadow) x pathLength — serial queue semantics
 * (Renderer/PathTracer.cpp:248-288, Cuda/PathTracer/PathTracer.cu:85-478).  frameNumber >= 1.
 * nthreads > 1 parallelises only the two trace passes (results identical). */
void orc_wavefront_render(orc_wavefront *w, uint32_t frameNumber, int nthreads);
/* AccumulateKernel, PathTracer.cu:480-496: running mean + tonemap into RGBA8. */
void orc_wavefront_accumulate(orc_wavefront *w, uint32_t frameNumber);
const float *orc_wavefront_radiance(const orc_wavefront *w);     /* localCount x 3 */
const float *orc_wavefront_accumulation(const orc_wavefront *w); /* localCount x 3 */
const uint32_t *orc_wavefront_rgba8(const orc_wavefront *w);     /* localCount */
const orc_queue_sizes *orc_wavefront_queue_sizes(const orc_wavefront *w);
void orc_wavefront_trace_stats(const orc_wavefront *w, orc_trace_stats *closest, orc_trace_stats *shadow);

uint32_t orc_tonemap_rgba8(const float rgb[3]);

#ifdef __cplusplus
}
#endif
#endif
