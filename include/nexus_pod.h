/*
 * nexus_pod.h — plain-old-data contracts shared by the host classes, the HIP device layer and the
 * test oracle.  Plain C (also valid C++ / HIP).  Byte layouts are identical to the reference's device
 * PODs so that buffers built by either side are interchangeable and fixtures are raw bytes.
 *
 * Reference layouts followed (all paths relative to /root/reference/Nexus/src):
 *   nx_bvh8_node     80 B  Cuda/BVH/BVH8.cuh:47-63, Geometry/BVH/BVH8.h:24-49
 *   nx_triangle      96 B  Cuda/Geometry/Triangle.cuh:25-49
 *   nx_mat4          64 B  Math/Mat4.h:24 (row major)
 *   nx_bvh_instance 160 B  Cuda/BVH/BVHInstance.cuh:7-14
 *   nx_material      60 B  Cuda/Scene/Material.cuh:5-51
 *   nx_light         12 B  Cuda/Scene/Light.cuh:4-33
 *   nx_camera        88 B  Cuda/Scene/Camera.cuh:5-15
 *   nx_render_settings 20 B Cuda/Scene/Scene.cuh:10-17, Renderer/RenderSettings.h:4-10
 */
#ifndef NEXUS_POD_H
#define NEXUS_POD_H

#include <stdint.h>

#ifdef __cplusplus
#define NX_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define NX_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

#define NX_ALIGN(n) __attribute__((aligned(n)))

/* Ylitie et al. 2017 compressed wide BVH node. */
typedef struct NX_ALIGN(16) nx_bvh8_node {
    float p[3];               /* origin of the local quantisation grid */
    uint8_t e[3];             /* biased exponents of the grid scale per axis */
    uint8_t imask;            /* bit i set: child slot i is an internal node */
    uint32_t childBaseIdx;    /* index of the first internal child */
    uint32_t triangleBaseIdx; /* index of the first leaf primitive in the index list */
    uint8_t meta[8];          /* per slot: inner 001|24+slot, leaf unary count|offset, empty 0 */
    uint8_t qlox[8], qloy[8], qloz[8];
    uint8_t qhix[8], qhiy[8], qhiz[8];
} nx_bvh8_node;
NX_STATIC_ASSERT(sizeof(nx_bvh8_node) == 80, "nx_bvh8_node must be 80 bytes");

typedef struct NX_ALIGN(8) nx_triangle {
    float pos0[3], pos1[3], pos2[3];
    float normal0[3], normal1[3], normal2[3];
    float texCoord0[2], texCoord1[2], texCoord2[2];
} nx_triangle;
NX_STATIC_ASSERT(sizeof(nx_triangle) == 96, "nx_triangle must be 96 bytes");

typedef struct nx_mat4 {
    float cell[16]; /* row major */
} nx_mat4;

typedef struct nx_bvh_instance {
    uint32_t bvhIdx;
    nx_mat4 invTransform;
    nx_mat4 transform;
    float boundsMin[3], boundsMax[3];
    int32_t materialId;
} nx_bvh_instance;
NX_STATIC_ASSERT(sizeof(nx_bvh_instance) == 160, "nx_bvh_instance must be 160 bytes");

enum { NX_MAT_DIFFUSE = 0, NX_MAT_DIELECTRIC = 1, NX_MAT_PLASTIC = 2, NX_MAT_CONDUCTOR = 3 };

typedef struct nx_material {
    union {
        struct { float albedo[3]; } diffuse;
        struct { float albedo[3]; float roughness; float ior; } dielectric;
        struct { float albedo[3]; float roughness; float ior; } plastic;
        struct { float ior[3]; float k[3]; float roughness; } conductor;
    };
    float emissive[3];
    float intensity;
    float opacity;
    int32_t diffuseMapId;
    int32_t emissiveMapId;
    int8_t type;
} nx_material;
NX_STATIC_ASSERT(sizeof(nx_material) == 60, "nx_material must be 60 bytes");

enum { NX_LIGHT_POINT = 0, NX_LIGHT_AREA = 1, NX_LIGHT_MESH = 2 };

typedef struct nx_light {
    union {
        struct { uint32_t radius; uint32_t intensity; } point;
        struct { uint32_t intensity; } area;
        struct { uint32_t meshId; } mesh; /* index of the mesh instance == index into the instance array */
    };
    int8_t type;
} nx_light;
NX_STATIC_ASSERT(sizeof(nx_light) == 12, "nx_light must be 12 bytes");

typedef struct NX_ALIGN(8) nx_camera {
    float position[3];
    float right[3];
    float up[3];
    float lensRadius;
    float lowerLeftCorner[3];
    float viewportX[3];
    float viewportY[3];
    uint32_t pad_;
    uint32_t resolution[2];
} nx_camera;
NX_STATIC_ASSERT(sizeof(nx_camera) == 88, "nx_camera must be 88 bytes");

typedef struct nx_render_settings {
    uint8_t useMIS;
    uint8_t pathLength;
    uint8_t pad_[2];
    float backgroundColor[3];
    float backgroundIntensity;
} nx_render_settings;
NX_STATIC_ASSERT(sizeof(nx_render_settings) == 20, "nx_render_settings must be 20 bytes");

/* A ray as handed to the batch-trace test hook: origin, direction (not necessarily unit). */
typedef struct nx_ray {
    float origin[3];
    float direction[3];
} nx_ray;

/* Hit record == the reference's D_Intersection (Cuda/Geometry/Ray.cuh:5-15). Miss: hitDistance == 1e30f. */
typedef struct nx_hit {
    float hitDistance;
    float u, v;
    uint32_t triIdx;
    uint32_t instanceIdx;
} nx_hit;

/* RGBA8 image, row 0 first, as uploaded to a reference texture (Assets/Texture.cpp:10-39). */
typedef struct nx_texture_desc {
    uint32_t width, height;
    const uint8_t *rgba8;
} nx_texture_desc;

/* query / result of the BSDF test hooks (nxhip_bsdf_sample_batch, nxhip_bsdf_eval_batch) */
typedef struct nx_bsdf_query {
    float wi[3];
    uint32_t rng;
    float wo[3];
    uint32_t pad_;
} nx_bsdf_query;
typedef struct nx_bsdf_result {
    float wo[3];
    float pdf;
    float throughput[3];
    uint32_t ok;      /* the BSDF's return value (false: sample rejected / invalid pdf) */
    uint32_t rngOut;  /* xorshift state after the call (sample only) */
    uint32_t pad_[3];
} nx_bsdf_result;
NX_STATIC_ASSERT(sizeof(nx_bsdf_query) == 32, "nx_bsdf_query must be 32 bytes");
NX_STATIC_ASSERT(sizeof(nx_bsdf_result) == 48, "nx_bsdf_result must be 48 bytes");


/* How Logic/Shade seed their RNG.  REFERENCE_SLOT mirrors Cuda/Random.cuh:79-82 + PathTracer.cu:143,326
 * (seed by queue slot, no bounce term).  PIXEL_KEYED seeds by (global pixel, bounce, frame): the image
 * then does not depend on queue slot order, so it is reproducible under racing compaction and under a
 * multi-GPU tile split. */
enum { NX_RNG_REFERENCE_SLOT = 0, NX_RNG_PIXEL_KEYED = 1 };

/* Queue compaction: ORDERED reproduces the reference's serial slot order (ascending thread index,
 * material kernels in graph order); FAST uses wave-aggregated atomics. */
enum { NX_COMPACT_FAST = 0, NX_COMPACT_ORDERED = 1 };

/* Conductor handling.  REFERENCE: the reference's ConductorMaterialKernel body is commented out
 * (PathTracer.cu:475-478) so conductor hits end the path.  EXTENDED: shade them (beyond the reference). */
enum { NX_CONDUCTOR_REFERENCE = 0, NX_CONDUCTOR_EXTENDED = 1 };

#define NX_PATH_MAX_LENGTH 100 /* Cuda/PathTracer/PathTracer.cuh:15 */
#define NX_MISS_DISTANCE 1e30f

#endif /* NEXUS_POD_H */
