/*
 * nexus_host.h — flat C view of the host-side data producers (BVH2 / BVH8 / TLAS builders, instance and
 * camera set-up) for callers that cannot use the C++ classes in include/nexus/ (the Python tests and bench).
 * The C++ classes mirror the reference's host API; these functions only wrap them.
 *   nxh_bvh8_build      BVH8Builder(tris).Init(); Build()     /root/reference/Nexus/src/Assets/AssetManager.cpp:23-37
 *   nxh_bvh2_build      BVH2(tris).Build()                     Geometry/BVH/BVH.cpp:13-26
 *   nxh_instance_init   BVHInstance::SetTransform + ToDevice   Geometry/BVH/BVHInstance.cpp:4-45
 *   nxh_tlas_build      TLAS::Build(); TLAS::Convert()         Geometry/BVH/TLAS.cpp:13-68
 *   nxh_camera_init     Camera::ToDevice                       Scene/Camera.cpp:142-168
 * All functions return 0 on success.
 */
#ifndef NEXUS_HOST_H
#define NEXUS_HOST_H

#include <stdint.h>

#include "nexus_pod.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nxh_bvh8 nxh_bvh8; /* owns nodes + primitive indices */

/* threads = 0: all hardware threads.  The result does not depend on the thread count. */
int nxh_bvh8_build(const nx_triangle *tris, uint32_t triCount, uint32_t threads, nxh_bvh8 **out);
int nxh_tlas_build(const nx_bvh_instance *instances, uint32_t instanceCount, nxh_bvh8 **out);
/* TLAS refit after instance transforms changed (same instances, same topology): recomputes, in place, every node's
 * frame and its children's quantised boxes bottom-up from instances[].boundsMin/Max with the builder's formulas; imask,
 * meta and indices stay.  O(n), against the O(n^2) agglomerative rebuild (Geometry/BVH/TLAS.cpp:13-91, 2.8 s for 16 000
 * instances).  The refitted tree bounds the same instances, so closest hits are those of a rebuilt tree; only the order
 * of visits (and which of two equidistant hits wins) may differ.  Returns 0, or 1 on malformed input. */
int nxh_tlas_refit(nx_bvh8_node *nodes, uint32_t nodeCount, const uint32_t *instanceIdx, const nx_bvh_instance *instances, uint32_t instanceCount);
uint32_t nxh_bvh8_node_count(const nxh_bvh8 *b);
uint32_t nxh_bvh8_prim_count(const nxh_bvh8 *b);
const nx_bvh8_node *nxh_bvh8_nodes(const nxh_bvh8 *b);
const uint32_t *nxh_bvh8_prim_indices(const nxh_bvh8 *b);
void nxh_bvh8_free(nxh_bvh8 *b);

/* BVH2 only (tests): nodes32 receives 2*triCount-1 32-byte nodes, triIdx triCount indices. */
int nxh_bvh2_build(const nx_triangle *tris, uint32_t triCount, uint32_t threads, void *nodes32, uint32_t *triIdx);

void nxh_mat4_from_trs(const float pos[3], const float rotDeg[3], const float scale[3], float out16[16]);
void nxh_mat4_invert(const float in16[16], float out16[16]);
int nxh_instance_init(nx_bvh_instance *out, uint32_t bvhIdx, int32_t materialId, const float transform16[16],
                      const nx_bvh8_node *blasRoot);
int nxh_camera_init(nx_camera *out, const float position[3], const float forward[3], float horizontalFovDeg,
                    uint32_t width, uint32_t height, float focusDist, float defocusAngleDeg);

/* ---- flat view of nexus::Scene / nexus::AssetManager / nexus::PathTracer (include/nexus/Scene.h, PathTracer.h) ------
 * The call sequence is the reference's: load meshes -> AssetManager::CreateBVH + AddMesh, Scene::CreateMeshInstance,
 * Scene::Update (TLAS build), PathTracer::UpdateDeviceScene, PathTracer::Render (Renderer/Renderer.cpp:41-77). */
/* ---- scene files (nexus::OBJLoader, include/nexus/OBJLoader.h): binary glTF 2.0 (.glb) and Wavefront .obj without Assimp.
 * Replaces what Assets/OBJLoader.cpp:8-239 obtains through Assimp: one mesh per glTF primitive, one instance per
 * (node, primitive) with the node transform decomposed to position / Euler degrees / scale, the reference's material
 * heuristics.  Returns 0 on success; the message of a failure is in nxs_last_error(). */
typedef struct nxh_loaded_scene nxh_loaded_scene;
typedef struct nx_loaded_instance {
    int32_t mesh, material;
    float position[3], rotation[3] /* Euler XYZ, degrees */, scale[3];
} nx_loaded_instance;
int nxh_load_scene_file(const char *file, nxh_loaded_scene **out);
void nxh_loaded_scene_free(nxh_loaded_scene *s);
uint32_t nxh_loaded_mesh_count(const nxh_loaded_scene *s);
uint32_t nxh_loaded_mesh_triangle_count(const nxh_loaded_scene *s, uint32_t mesh);
int nxh_loaded_mesh_triangles(const nxh_loaded_scene *s, uint32_t mesh, nx_triangle *dst);
uint32_t nxh_loaded_material_count(const nxh_loaded_scene *s);
int nxh_loaded_materials(const nxh_loaded_scene *s, nx_material *dst);
uint32_t nxh_loaded_instance_count(const nxh_loaded_scene *s);
int nxh_loaded_instances(const nxh_loaded_scene *s, nx_loaded_instance *dst);
/* Images the file's materials refer to (glTF baseColorTexture -> kind 0 diffuse, emissiveTexture -> kind 1 emissive; what
 * OBJLoader.cpp:115-160 creates through Assimp + stb_image), decoded to RGBA8 with row 0 = top row.  Per material the index
 * of its diffuse / emissive texture in this list, -1 for none.  An image that cannot be decoded (PNG only) is dropped with
 * a warning, as the reference prints and carries on (IMGLoader.cpp:24-25). */
uint32_t nxh_loaded_texture_count(const nxh_loaded_scene *s);
int nxh_loaded_texture_info(const nxh_loaded_scene *s, uint32_t index, uint32_t *width, uint32_t *height, int32_t *kind);
int nxh_loaded_texture_pixels(const nxh_loaded_scene *s, uint32_t index, uint8_t *dstRgba8);
int nxh_loaded_material_textures(const nxh_loaded_scene *s, int32_t *diffuseTexture, int32_t *emissiveTexture);
uint32_t nxh_loaded_warning_count(const nxh_loaded_scene *s);
const char *nxh_loaded_warning(const nxh_loaded_scene *s, uint32_t index);
/* IMGLoader::LoadIMG (Assets/IMGLoader.cpp:17-41: stbi_load with 4 channels) for PNG data: RGBA8, row 0 first.
 * dstRgba8 may be NULL to query the size first; *channels = channels of the file (1-4). */
int nxh_decode_png(const uint8_t *data, size_t size, uint32_t *width, uint32_t *height, uint32_t *channels, uint8_t *dstRgba8, size_t dstCapacity);

typedef struct nxs_scene nxs_scene;
typedef struct nxs_pathtracer nxs_pathtracer;
struct nxhip_ctx;

const char *nxs_last_error(void);
int nxs_scene_create(uint32_t width, uint32_t height, nxs_scene **out);
void nxs_scene_destroy(nxs_scene *s);
int nxs_scene_add_material(nxs_scene *s, const nx_material *m, int32_t *materialId);
int nxs_scene_add_texture(nxs_scene *s, int kind /*0 diffuse, 1 emissive*/, const uint8_t *rgba8, uint32_t w, uint32_t h, int32_t *texId);
int nxs_scene_set_hdr_map(nxs_scene *s, const uint8_t *rgba8, uint32_t w, uint32_t h);
int nxs_scene_add_mesh(nxs_scene *s, const nx_triangle *tris, uint32_t triCount, int32_t materialId, int32_t *meshId);
int nxs_scene_create_instance(nxs_scene *s, uint32_t meshId, int32_t materialId, const float pos[3], const float rotDeg[3],
                              const float scale[3], int32_t *instanceId);
/* MeshInstance::AssignMaterial + Scene::InvalidateMeshInstance (the viewer's material picker): applied by the next nxs_scene_update. */
int nxs_scene_assign_material(nxs_scene *s, uint32_t instanceId, int32_t materialId);
/* MeshInstance::SetTransform + Scene::InvalidateMeshInstance: move an existing instance; applied by the next nxs_scene_update. */
int nxs_scene_set_instance_transform(nxs_scene *s, uint32_t instanceId, const float pos[3], const float rotDeg[3], const float scale[3]);
/* Extension (off by default): refit the TLAS in nxs_scene_update when only existing instances changed, instead of the
 * reference's full rebuild. */
int nxs_scene_set_tlas_refit(nxs_scene *s, int enable);
/* Extension (off by default): the TLAS is built and refitted on the device (Scene::SetDeviceTlasBuild -> nxhip_rebuild_tlas /
 * nxhip_set_instance_transforms); nxs_scene_update then runs no TLAS builder on the host. */
int nxs_scene_set_device_tlas(nxs_scene *s, int enable);
/* Scene::CreateMeshInstanceFromFile — Scene.cpp:83-91: adds the file's materials, meshes (one BVH8 each) and instances. */
int nxs_scene_load_file(nxs_scene *s, const char *path, const char *fileName);
int nxs_scene_set_camera(nxs_scene *s, const float pos[3], const float forward[3], float horizontalFovDeg, float focusDist,
                         float defocusAngleDeg);
int nxs_scene_set_render_settings(nxs_scene *s, const nx_render_settings *settings);
int nxs_scene_update(nxs_scene *s);
uint32_t nxs_scene_light_count(const nxs_scene *s);
uint32_t nxs_scene_instance_count(const nxs_scene *s);

int nxs_pathtracer_create(uint32_t width, uint32_t height, int device, nxs_pathtracer **out);
void nxs_pathtracer_destroy(nxs_pathtracer *p);
int nxs_pathtracer_set_modes(nxs_pathtracer *p, int rngMode, int compactMode, int conductorMode);
/* PathTracer::SetFramesPerPass / SetPassesInFlight (extensions, include/nexus/PathTracer.h): frames per Render() call,
 * Render() calls in flight */
int nxs_pathtracer_set_frames_per_pass(nxs_pathtracer *p, uint32_t frames);
int nxs_pathtracer_set_passes_in_flight(nxs_pathtracer *p, uint32_t passes);
/* PathTracer::SetPixelOrder / SetEntryPoints (extensions): NXHIP_ORDER_* of the frame's paths; primary rays from their run's entry state */
int nxs_pathtracer_set_pixel_order(nxs_pathtracer *p, int order);
int nxs_pathtracer_set_entry_points(nxs_pathtracer *p, int on);
/* PathTracer::SetDeviceBlasBuild (extension, off by default): meshes added to `s` from now on get their BVH8 from the device
 * builder (nxhip_build_blas) instead of the host's; switch it off, or destroy the scene, before `p` goes away. */
int nxs_pathtracer_set_device_blas_build(nxs_pathtracer *p, nxs_scene *s, int enable);
int nxs_pathtracer_update_device_scene(nxs_pathtracer *p, nxs_scene *s);
int nxs_pathtracer_render(nxs_pathtracer *p, nxs_scene *s);
int nxs_pathtracer_reset_frame_number(nxs_pathtracer *p);
uint32_t nxs_pathtracer_frame_number(const nxs_pathtracer *p);
int nxs_pathtracer_read_pixels(nxs_pathtracer *p, uint32_t *rgba8);
struct nxhip_ctx *nxs_pathtracer_device_context(nxs_pathtracer *p);

/* Scene::AddHDRMap(filePath, fileName) — Scene/Scene.cpp:93-97: environment map from a Radiance .hdr (or .png) file. */
int nxs_scene_add_hdr_map_file(nxs_scene *s, const char *path, const char *fileName);

/* ---- nexus::Renderer (include/nexus/Renderer.h): the reference's frame driver without its window ---------------------
 * Renderer::Render (Renderer/Renderer.cpp:41-77): scene.Update() + ResetFrameNumber when the scene is invalid, then
 * UpdateDeviceScene + PathTracer::Render; SaveScreenshot (Renderer.cpp:183-215): PNG with rows flipped. */
typedef struct nxs_renderer nxs_renderer;
int nxs_renderer_create(uint32_t width, uint32_t height, nxs_scene *scene, int device, nxs_renderer **out);
void nxs_renderer_destroy(nxs_renderer *r);
int nxs_renderer_render(nxs_renderer *r, nxs_scene *scene, float deltaTime);
int nxs_renderer_reset(nxs_renderer *r);
int nxs_renderer_on_resize(nxs_renderer *r, uint32_t width, uint32_t height);
int nxs_renderer_save_screenshot(nxs_renderer *r, const char *path);
int nxs_renderer_save_exr(nxs_renderer *r, const char *path); /* extension: float accumulation as OpenEXR */
uint32_t nxs_renderer_frame_number(const nxs_renderer *r);
double nxs_renderer_megasamples_per_second(const nxs_renderer *r); /* MetricsPanel.cpp:28-56 */
struct nxhip_ctx *nxs_renderer_device_context(nxs_renderer *r);
int nxs_renderer_set_modes(nxs_renderer *r, int rngMode, int compactMode, int conductorMode);
/* Image writers without stb: PNG (RGBA8; the reference's stbi_write_png) and OpenEXR (scanline, uncompressed, float32 B G R).
 * flipVertically != 0 writes the last row first (the render buffer's row 0 is the bottom of the viewport). */
int nxh_write_png(const char *path, const uint32_t *rgba8, uint32_t width, uint32_t height, int flipVertically);
int nxh_write_exr(const char *path, const float *rgb, uint32_t width, uint32_t height, int flipVertically);

#ifdef __cplusplus
}
#endif
#endif
