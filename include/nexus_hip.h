/*
 * nexus_hip.h — C-ABI of the MI355X (gfx950) device layer: the wavefront path-tracing hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI layer: its host classes reach the device
 * through CUDA symbols and bare kernel pointers (17 GetDevice*Address() getters,
 * /root/reference/Nexus/src/Cuda/PathTracer/PathTracer.cuh:86-103, and (void*)XxxKernel pointers,
 * Renderer/PathTracer.cpp:99-107).  Each entry point below names the reference interface it replaces.
 * Plain C: opaque context, plain pointers and sizes, int status (0 = ok, message via
 * nxhip_last_error()); no HIP, torch or C++ types in any signature.  Host arrays passed in are copied
 * before the call returns.  A context belongs to one host thread at a time; one context per GPU.
 */
#ifndef NEXUS_HIP_H
#define NEXUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "nexus_fmath.h"
#include "nexus_pod.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nxhip_ctx nxhip_ctx;

enum {
    NXHIP_OK = 0,
    NXHIP_ERR_INVALID = 1, /* bad argument / state */
    NXHIP_ERR_HIP = 2,     /* a HIP runtime call failed */
    NXHIP_ERR_NO_DEVICE = 3,
    NXHIP_ERR_TRAVERSAL = 4, /* a trace kernel gave up on rays that made no progress (malformed BVH): reported by nxhip_sync and the read-backs */
    NXHIP_ERR_ABI = 5,       /* caller and library (or the library's own translation units) disagree about a struct layout / the API version */
    NXHIP_ERR_TIMEOUT = 6    /* nxhip_sync_timeout gave up waiting: the context is DEAD from then on (every later call returns this) */
};

/* Bumped whenever an entry point changes its signature or meaning, or a struct of this header / nexus_pod.h its layout. */
#define NXHIP_API_VERSION 7

/* Thread-local message of the last failing call (replaces CheckCudaErrors -> exit(99), Utils/Utils.cpp:3-12). */
const char *nxhip_last_error(void);

/* Number of visible HIP devices (0 if none / runtime unavailable). */
int nxhip_device_count(void);

/* PathTracer::PathTracer(width, height) + Reset() — Renderer/PathTracer.cpp:5-28, 92-241: allocates every
 * queue for `localPixels` paths.  `stream` is a hipStream_t passed as void* (NULL: the context creates its own). */
int nxhip_create(int device, uint32_t width, uint32_t height, void *stream, nxhip_ctx **out);
/* PathTracer::~PathTracer / FreeDeviceBuffers — PathTracer.cpp:30-90 */
void nxhip_destroy(nxhip_ctx *ctx);
/* PathTracer::OnResize — PathTracer.cpp:290-303 (frees and re-allocates queues, resets the frame number) */
int nxhip_resize(nxhip_ctx *ctx, uint32_t width, uint32_t height);
int nxhip_sync(nxhip_ctx *ctx);
/* nxhip_sync with a wall-clock limit (milliseconds).  The reference waits for the device without one and a kernel that never ends hangs
 * the viewer (CheckCudaErrors only sees launches that RETURN: Utils/Utils.cpp:3-12, Renderer/PathTracer.cpp:280-284).  Polls the
 * context's streams; work done in time: as nxhip_sync.  Otherwise NXHIP_ERR_TIMEOUT and the context is marked DEAD: nothing can be
 * said about the device's state, every later call on it returns NXHIP_ERR_TIMEOUT without touching the device (nxhip_destroy frees
 * the host side only); the caller reports and exits, or starts a fresh process — there is no in-process recovery from a hung GPU. */
int nxhip_sync_timeout(nxhip_ctx *ctx, uint32_t timeoutMs);

/* ---- scene upload -------------------------------------------------------------------------------- */

/* BVH8::InitDeviceData + AssetManager::InitDeviceData — Geometry/BVH/BVH8.cpp:28-33, Assets/AssetManager.cpp:45-49
 * (symbol `bvhs`).  Returns the BLAS id == index instances refer to as bvhIdx.  ids are dense from 0. */
int nxhip_upload_blas(nxhip_ctx *ctx, const nx_bvh8_node *nodes, uint32_t nodeCount, const nx_triangle *tris,
                      uint32_t triCount, const uint32_t *triIdx, int32_t *blasId);
/* BLAS built ON THE DEVICE from triangles alone.  Replaces BVH2::Build + BVH8Builder::Init /
 * Build + BVH8::InitDeviceData (Geometry/BVH/BVH.cpp:13-210, BVH8Builder.cpp:10-393, BVH8.cpp:28-33) with the same two
 * steps in HBM: a binary tree by the reference's rule — top-down, binned surface-area heuristic over the centroid bounds,
 * one primitive per leaf, "split in half" when nothing separates (level-synchronous: atomics into 16 bins per axis and node,
 * a scan partition per level, nodes of up to 8 primitives finished by an exact sweep) — and the reference's SAH dynamic
 * programme for the collapse into 80-byte 8-wide nodes (cost table bottom-up, decisions followed top-down, the reference's
 * octant slot assignment and quantisation).  29 ms per million triangles (host builder of this repo: 0.4 s on 16 threads; the
 * reference: ~10 s on one) and fewer node visits per ray than the host build on every mesh of
 * profiles/r03_builder_quality.txt.  A valid, conservative CWBVH whose node bytes differ from the host builder's (another
 * tree); hit records are the same up to equidistant ties.  Returns the BLAS id like nxhip_upload_blas. */
int nxhip_build_blas(nxhip_ctx *ctx, const nx_triangle *tris, uint32_t triCount, int32_t *blasId);
/* The BLASes of `meshCount` meshes in ONE device build.  The reference creates one BVH8 per aiMesh of a file
 * (Assets/OBJLoader.cpp:213-239 -> AssetManager::AddMesh -> CreateBVH, Assets/AssetManager.cpp:23-37: a glTF scene is hundreds
 * or thousands of small meshes); built one by one on the device each of them pays the build's ~25 level synchronisations and
 * its allocations.  Here the meshes are a forest over the concatenated triangles — per-mesh Morton order, one root segment per
 * mesh, then the level loops of nxhip_build_blas (binning, split, partition; cost table; collapse) over all meshes at once — and
 * the BLASes share four pooled allocations.  Each tree is the tree nxhip_build_blas(tris[m], triCounts[m]) builds (same
 * decisions from the same counts, boxes and orders; node numbering differs as between two single builds).  blasIds[m] (may be
 * NULL) = the id of mesh m; ids are consecutive.  With another builder selected (nxhip_set_device_builder) the meshes are built
 * one by one.  1 000 meshes of 1 000 triangles: see profiles/r04_blas_batch.txt. */
int nxhip_build_blas_batch(nxhip_ctx *ctx, const nx_triangle *const *tris, const uint32_t *triCounts, uint32_t meshCount, int32_t *blasIds);
/* nxhip_read_blas for `count` consecutive BLAS ids: nodes and primitive indices concatenated in id order (either may be NULL),
 * nodeCounts[k] = nodes of BLAS firstBlasId + k.  One transfer each when the range comes from one nxhip_build_blas_batch call. */
int nxhip_read_blas_batch(nxhip_ctx *ctx, int32_t firstBlasId, uint32_t count, nx_bvh8_node *nodes, uint32_t nodeCapacity, uint32_t *nodeCounts,
                          uint32_t *primIdx, uint32_t primCapacity);
/* Which binary tree the device builders (nxhip_build_blas, nxhip_rebuild_tlas) collapse into 8-wide nodes.
 * NXHIP_BUILDER_SAH (default): the top-down binned SAH build described above.  0: the binary radix tree of the 63-bit
 * Morton codes (LBVH: sort + one launch; 15 ms per million triangles).  clusteringRadius > 0: parallel locally-ordered
 * clustering — the Morton-sorted primitives are merged bottom-up, every cluster pairing with the neighbour within `radius`
 * places whose union has the smallest surface area; a few dozen rounds.  All three go through the same SAH collapse.
 * Measured (profiles/r03_builder_quality.txt, node visits per ray against the host SAH build): top-down SAH -1 ... -14 %,
 * radix tree +1 ... +25 %, clustering +2 ... +15 %.  Either way the result is a valid conservative CWBVH. */
#define NXHIP_BUILDER_SAH (-1)
int nxhip_set_device_builder(nxhip_ctx *ctx, int clusteringRadius);
/* Read a BLAS's nodes / primitive index list back (either may be NULL; *nodeCount = nodes it has). */
int nxhip_read_blas(nxhip_ctx *ctx, int32_t blasId, nx_bvh8_node *nodes, uint32_t nodeCapacity, uint32_t *primIdx, uint32_t primCapacity,
                    uint32_t *nodeCount);
int nxhip_clear_blas(nxhip_ctx *ctx);
/* TLAS::UpdateDeviceData — Geometry/BVH/TLAS.cpp:93-100 (symbols `tlas`, `blas`). */
int nxhip_set_tlas(nxhip_ctx *ctx, const nx_bvh8_node *nodes, uint32_t nodeCount, const uint32_t *instanceIdx,
                   const nx_bvh_instance *instances, uint32_t instanceCount);
/* TLAS built ON THE DEVICE from the instances alone ("refit / rebuild on device"): the builder
 * of nxhip_build_blas (whichever nxhip_set_device_builder selected; default: top-down binned SAH) run over the instances'
 * world-space boxes, collapsed into 80-byte nodes with up to three instances per leaf slot, then
 * installed exactly as nxhip_set_tlas installs a host-built tree (instances[i].boundsMin / boundsMax must be filled in, as
 * BVHInstance::SetTransform does).  Replaces TLAS::Build + TLAS::Convert — the reference's O(n^2) agglomerative clustering
 * on the CPU and BVH8 conversion, re-run on every edit (Geometry/BVH/TLAS.cpp:13-91, Scene/Scene.cpp:29-55) — when instances are
 * added or removed: 16 000 instances take the host clustering 0.8 s and this build about a millisecond.  A different tree
 * than the host's (hits identical up to equidistant ties).  Afterwards nxhip_set_instance_transforms refits it in place. */
int nxhip_rebuild_tlas(nxhip_ctx *ctx, const nx_bvh_instance *instances, uint32_t instanceCount);
/* The installed TLAS's instance index list (leaf order; instanceCount entries; NULL: only the node count). */
int nxhip_read_tlas_index(nxhip_ctx *ctx, uint32_t *instanceIdx, uint32_t capacity, uint32_t *nodeCount);
/* Dynamic transforms without a host round trip of the scene.  Replaces, for instances that
 * already exist, MeshInstance::SetTransform -> BVHInstance::SetTransform -> TLAS::Build -> TLAS::UpdateDeviceData
 * (Geometry/BVH/BVHInstance.cpp:4-29, Scene/Scene.cpp:29-55, Geometry/BVH/TLAS.cpp:13-100: an O(n^2) agglomerative rebuild
 * on the CPU per edit).  transforms16: count row-major object-to-world matrices.  On the device: inverse matrix and world
 * bounds of every listed instance (bit-identical to nexus::BVHInstance::SetTransform), its traversal record, then a
 * bottom-up refit of the TLAS BVH8 with unchanged topology (bit-identical to nexus::collapse::Refit / nxh_tlas_refit).
 * Runs on the context's stream after the frames already issued. */
int nxhip_set_instance_transforms(nxhip_ctx *ctx, const uint32_t *instanceIds, const float *transforms16, uint32_t count);
/* Read the device's TLAS nodes / instance table back (tests; either destination may be NULL). */
int nxhip_read_tlas(nxhip_ctx *ctx, nx_bvh8_node *nodes, uint32_t nodeCapacity, nx_bvh_instance *instances, uint32_t instanceCapacity);
/* AssetManager device materials — Assets/AssetManager.cpp:57-62,106-116 */
int nxhip_set_materials(nxhip_ctx *ctx, const nx_material *materials, uint32_t count);
/* Scene::m_DeviceLights — Scene/Scene.cpp:142-176 */
int nxhip_set_lights(nxhip_ctx *ctx, const nx_light *lights, uint32_t count);
/* Texture::ToDevice — Assets/Texture.cpp:10-39 (RGBA8, sRGB, wrap, bilinear).  kind: 0 diffuse, 1 emissive, 2 hdr map.
 * Diffuse/emissive maps get ids in upload order; the hdr map replaces the previous one. */
int nxhip_upload_texture(nxhip_ctx *ctx, int kind, const uint8_t *rgba8, uint32_t width, uint32_t height, int32_t *texId);
int nxhip_clear_textures(nxhip_ctx *ctx);
/* Extension (BASELINE.json configs[3] asks for "HDR envmap NEE/MIS"; the reference only adds the environment when a ray
 * misses, PathTracer.cu:152-164, and its NEE knows mesh lights only, :227): with enable != 0 the next-event estimation
 * treats the environment map as one more light — picked with probability 1 / (lightCount + 1), direction drawn from the
 * map's luminance x sin(theta) distribution, shadow ray to infinity — and a BSDF-sampled ray that misses is weighted
 * against that sampler with the power heuristic.  Unbiased either way; the expectation of a frame is unchanged.  Needs an
 * uploaded environment map (kind 2); off by default. */
int nxhip_set_env_sampling(nxhip_ctx *ctx, int enable);
/* PathTracer::UpdateDeviceScene / Scene::ToDevice — Renderer/PathTracer.cpp:305-308, Scene/Scene.cpp:115-140 */
int nxhip_set_camera(nxhip_ctx *ctx, const nx_camera *camera);
int nxhip_set_render_settings(nxhip_ctx *ctx, const nx_render_settings *settings);
/* Extensions that do not exist in the reference (see nexus_pod.h): RNG keying, compaction order, conductor. */
int nxhip_set_modes(nxhip_ctx *ctx, int rngMode, int compactMode, int conductorMode);
/* Multi-GPU tile split: this context renders `localCount` pixels; pixelMap[i] = global pixel index of local
 * pixel i (NULL: identity over width*height).  Re-allocates the queues for localCount paths. */
int nxhip_set_pixel_map(nxhip_ctx *ctx, const uint32_t *pixelMap, uint32_t localCount);
/* The order of the context's paths over the FULL frame (no tile split): NXHIP_ORDER_ROWS = image rows, the reference's (thread k of
 * GenerateKernel is pixel k: PathTracer.cu:85-100) and the default; NXHIP_ORDER_TILES = 8 x 8 pixel tiles, row-major inside a
 * tile, tiles left to right in bands of eight rows — the 64 primary rays a wave fetches together are then a compact block of the
 * image instead of a 64 x 1 strip (coherent node fetches, and what entry points need: nxhip_set_entry_points).  With the
 * pixel-keyed RNG the image does not depend on it.  Equivalent to nxhip_set_pixel_map with the map nxhip_tile_pixel_map(width,
 * height, 1, 0, 1, order, ...) returns; unlike a caller's map the ORDER is kept across nxhip_resize.  A later nxhip_set_pixel_map /
 * nxhip_mgpu_init replaces it. */
enum { NXHIP_ORDER_ROWS = 0, NXHIP_ORDER_TILES = 1 };
int nxhip_set_pixel_order(nxhip_ctx *ctx, int order);

/* Batch `frames` consecutive frames into one pass of the wavefront (default 1 = the reference's one frame per
 * Render()).  Every queue then holds localCount * frames paths, so each kernel launch carries `frames` times the work:
 * the latency tail of a trace launch (its slowest ray) and the per-launch overheads are amortised, at the price of
 * HBM capacity (about 0.3 KB per path).  Each frame keeps its own frame number / RNG streams; the accumulate step
 * applies the frames' running-mean updates in order.  The queues grow when needed and are kept when `frames` shrinks
 * (a shorter last pass of a frame budget costs no allocation); the frame number and the accumulation are untouched. */
int nxhip_set_frames_per_pass(nxhip_ctx *ctx, uint32_t frames);

/* Passes in flight (default 1 = the reference's behaviour: a pass starts when the previous one has finished).  With
 * R > 1 consecutive nxhip_render_frame calls go round robin to R slots, each with its own queues (R times the queue memory),
 * stream and graph instance, so that the drain phase of a pass — a few long rays, most of the GPU idle, at one frame per
 * pass more than half of the pass — overlaps with the bulk of the next passes.  nxhip_accumulate folds every finished pass
 * into the one accumulation, oldest first: the image is bit-identical to R = 1.  Worth it for small passes (one frame per
 * pass: 2.2x at R = 6; 20 frames per pass: +20 % at R = 4; the persistent trace launches of the slots are sized to share the
 * CUs).  Needs enough hardware queues: export GPU_MAX_HW_QUEUES=24
 * before the process touches HIP (the runtime's default of 4 serialises the slots).  Kernel timing, the counting variant and
 * a bound radiance buffer fall back to one pass at a time. */
int nxhip_set_passes_in_flight(nxhip_ctx *ctx, uint32_t passes);

/* Tail kernel (no counterpart in the reference, whose graph has one node per kernel and bounce, PathTracer.cpp:114-124).
 * From bounce `bounce` on — 2 .. pathLength, 0 = off — the rest of every path is run by ONE launch after the trace of
 * bounce - 1: each wave takes 64 paths and loops logic -> shade -> shadow ray -> continuation ray per lane.  Late bounces
 * carry a few per cent of a pass's rays but each costs a trace level as long as its slowest ray plus five more launches.
 * Default NXHIP_TAIL_AUTO: for passes of up to four 1080p frames' worth of paths, bounce 5 (bounce 3 for up to 2.5 frames with
 * at most three passes in flight) — one frame per pass +21 %, with six passes in flight +16 % — off for larger ones (where it
 * loses).  Same functions and the same order of additions per pixel:
 * the image is bit-identical.  Only with NX_RNG_PIXEL_KEYED and NX_COMPACT_FAST, and not while kernel timing or the counting
 * variant is enabled (those passes use the level-by-level graph). */
#define NXHIP_TAIL_AUTO 0xffffffffu
int nxhip_set_tail_bounce(nxhip_ctx *ctx, uint32_t bounce);
/* Entry points of the primary rays (off by default).  on != 0: every pass first walks, once per run of 64 consecutive paths, the
 * node steps whose outcome is provably the same for every primary ray the run can contain (conservative bundle-against-box tests:
 * nx_entry.hip), and the closest-hit launch of the primary rays starts each ray from that state instead of the TLAS root — the
 * reference starts every ray at the root (Cuda/BVH/BVH8Traversal.cuh:165-192).  Hit records are unchanged bit for bit; the node
 * visit counts of nxhip_read_trace_stats drop by the steps saved.  Takes effect for a pinhole camera (lens radius 0). */
int nxhip_set_entry_points(nxhip_ctx *ctx, int on);
/* The entry states of the last rendered pass, 80 bytes each (nx_device.h EntryState: six stack entries, node group, leaf group,
 * then int32 sp, instSp, leafSlot, steps) — a test hook: how many node steps the walk saved per run.  *count = number of runs. */
int nxhip_read_entry_states(nxhip_ctx *ctx, void *out, uint32_t capacityRuns, uint32_t *count);
/* Test hook for the thin kernel (nx_trace.hip): the hand-over rule — at most `lanes` busy lanes of a dry wave for at least `iters`
 * iterations (product: 16 / 16; 64 / 0 makes every wave hand over the first rays it takes, after one iteration) — and whether the ray-batch
 * hooks (nxhip_trace_batch, nxhip_trace_shadow_batch) use the hand-over + thin launch too, so that a test can put arbitrary rays
 * through the cooperative search and compare the records with the oracle's (inHooks bit 0).  inHooks bit 1: hand over after `iters`
 * iterations of EVERY stretch between two refill points, dry queue or not — the rays then reach the thin kernel with the traversal
 * state of exactly that many steps (round 6: the hand-over carries the state).  nxhip_debug_thin_counts: the rays the last hook call
 * handed over (closest-hit, any-hit). */
int nxhip_debug_set_thin(nxhip_ctx *ctx, uint32_t lanes, uint32_t iters, int inHooks);
/* ... and how many items a thin wave's pool may hold (0 = the product's 960 of 1 024): a round whose children do not fit puts items back and
 * goes on with fewer per round — a test lowers the limit so that ordinary scenes drive that path; results must not change. */
int nxhip_debug_set_thin_pool(nxhip_ctx *ctx, uint32_t slots);
int nxhip_debug_thin_counts(nxhip_ctx *ctx, int32_t counts[2]);
/* ... and of level `bounce` (0 = the primary rays) of the pass rendered last: what its trace launches handed over (closest-hit, any-hit). */
int nxhip_debug_thin_counts_of_pass(nxhip_ctx *ctx, uint32_t bounce, int32_t counts[2]);

/* ---- rendering ----------------------------------------------------------------------------------- */

/* PathTracer::ResetFrameNumber — PathTracer.cpp:243-246 */
int nxhip_reset_frame_number(nxhip_ctx *ctx);
int nxhip_set_frame_number(nxhip_ctx *ctx, uint32_t frameNumber); /* next render uses frameNumber + 1 */
uint32_t nxhip_frame_number(nxhip_ctx *ctx);
/* PathTracer::Render minus AccumulateKernel — PathTracer.cpp:248-276: frameNumber++, Generate, Trace, then
 * pathLength x (Logic, 4 x Shade, Trace || TraceShadow), replayed as one hipGraph.  Asynchronous.  Renders one pass
 * = frames-per-pass frames (1 unless nxhip_set_frames_per_pass was called). */
int nxhip_render_frame(nxhip_ctx *ctx);
/* AccumulateKernel — Cuda/PathTracer/PathTracer.cu:480-496, PathTracer.cpp:278.  Asynchronous. */
int nxhip_accumulate(nxhip_ctx *ctx);
/* `frames` frames: passes of frames-per-pass frames (render + accumulate), the last one shorter if `frames` is not a
 * multiple. */
int nxhip_render(nxhip_ctx *ctx, uint32_t frames);

/* Read-back (synchronises).  radiance: localCount * framesPerPass x 3 floats (frame slices one after the other);
 * accumulation: localCount x 3 floats; rgba8: localCount uint32
 * (the reference's GL pixel buffer, OpenGL/PixelBuffer.cpp:4-41). */
int nxhip_read_radiance(nxhip_ctx *ctx, float *dst);
int nxhip_read_accumulation(nxhip_ctx *ctx, float *dst);
int nxhip_read_rgba8(nxhip_ctx *ctx, uint32_t *dst);
/* Resume: load a previously read accumulation (localCount x 3 floats) and continue as if `frameNumber` frames had been
 * rendered; the next frame is frameNumber + 1 and the running mean continues bit for bit.  (The reference keeps the
 * accumulation on the device only and restarts on every change, PathTracer.cpp:243-246.) */
int nxhip_write_accumulation(nxhip_ctx *ctx, const float *src, uint32_t frameNumber);
/* Device pointers for zero-copy consumers on the same GPU (e.g. an RCCL gather of radiance tiles):
 * float4 per local pixel (xyz = radiance, w unused).  Valid until the next resize / set_pixel_map. */
/* PathTracer::FreeDeviceBuffers (Renderer/PathTracer.cpp:35-92): give the path-state and queue buffers of every pass slot back
 * (about 0.28 KB per path and slot: 36 GB at 1080p x 64 frames per pass).  The accumulated image, the scene and the
 * configuration stay; the next nxhip_render_frame (or ray-batch hook) allocates what it needs again. */
int nxhip_release_queues(nxhip_ctx *ctx);
void *nxhip_radiance_device_ptr(nxhip_ctx *ctx);
void *nxhip_accumulation_device_ptr(nxhip_ctx *ctx);
/* Make the context write its per-frame radiance into caller-owned device memory (float4[capacity], capacity >=
 * localCount * framesPerPass) — e.g. a torch tensor that is then handed to an RCCL gather without a copy.  NULL: back to the
 * context's own buffer.  The binding is dropped by nxhip_resize / nxhip_set_pixel_map. */
int nxhip_bind_radiance(nxhip_ctx *ctx, void *radianceDevice, uint32_t capacity);
/* Root-side accumulate + tonemap of externally gathered radiance: src = device float4 laid out [slices][sliceStride];
 * element k (< count) of slice s is the radiance of frame firstFrame + s at full-image pixel srcPixelMapDevice[k]
 * (device uint32[count]; NULL: k).  Writes the context's accumulation / RGBA8 buffers, which always cover
 * width*height pixels. */
int nxhip_accumulate_external(nxhip_ctx *ctx, const void *srcRadianceDevice, uint32_t count, uint32_t slices, uint32_t sliceStride,
                              uint32_t firstFrame, const void *srcPixelMapDevice);
/* Root-side image assembly when every rank accumulates its own tiles (nxhip_accumulate with a pixel map) and only the
 * accumulated tiles travel: element k (< count) of srcAccumulationDevice (device float4[count], a rank's local-order
 * accumulation) is copied to dstAccumulationDevice[srcPixelMapDevice[k]] (device float4[width*height], caller-owned) and,
 * if dstRgba8Device is not NULL, tonemapped into dstRgba8Device[...] (device uint32[width*height]).  No arithmetic is
 * applied to the accumulated values, so the assembled image is bit-identical to a single-GPU accumulation. */
int nxhip_compose_tiles(nxhip_ctx *ctx, const void *srcAccumulationDevice, uint32_t count, const void *srcPixelMapDevice,
                        void *dstAccumulationDevice, void *dstRgba8Device);
/* Read-back of the full width*height accumulation / RGBA8 image (after nxhip_accumulate_external). */
int nxhip_read_full_accumulation(nxhip_ctx *ctx, float *dst);
int nxhip_read_full_rgba8(nxhip_ctx *ctx, uint32_t *dst);

/* ---- multi-GPU: interleaved row tiles + one RCCL gather per pass ---------------------------------------
 * No counterpart in the reference (single GPU; SURVEY.md section 8e defines the layer).  One context per GPU — one process
 * or one host thread each.  RCCL (librccl.so.1, or $NX_RCCL_LIB) is loaded on first use.  Call sequence per rank:
 *   rank 0: nxhip_mgpu_unique_id(id) -> hand the 128 bytes to every rank (MPI, a file, a socket: the caller's choice)
 *   all   : nxhip_mgpu_init(ctx, world, rank, id, tileRows)   [ncclCommInitRank; installs the rank's pixel map]
 *   loop  : nxhip_render_frame, nxhip_accumulate, nxhip_mgpu_gather   [asynchronous, one stream]
 *   rank 0: nxhip_mgpu_read_rgba8 / nxhip_mgpu_read_accumulation      [full width x height image]
 * With the pixel-keyed RNG (nxhip_set_modes) the assembled image equals the single-GPU image bit for bit. */
/* Row r belongs to rank (r / tileRows) % worldSize; height must be a multiple of tileRows * worldSize.  Writes the global
 * pixel index of every local pixel in the context's path order (tiledOrder != 0: 8x8 pixel tiles; 0: rows);
 * out == NULL: only *localCount. */
int nxhip_tile_pixel_map(uint32_t width, uint32_t height, int worldSize, int rank, uint32_t tileRows, int tiledOrder, uint32_t *out,
                         uint32_t *localCount);
int nxhip_mgpu_unique_id(void *id128);
int nxhip_mgpu_init(nxhip_ctx *ctx, int worldSize, int rank, const void *id128, uint32_t tileRows);
/* The same with a communicator the caller already owns (an ncclComm_t passed as void*; it is not destroyed). */
int nxhip_mgpu_attach(nxhip_ctx *ctx, void *ncclComm, int worldSize, int rank, uint32_t tileRows);
/* ncclGather of every rank's accumulated tile (16 B per local pixel) to rank 0 on the context's stream; rank 0 then
 * scatters the tiles into the full image and tonemaps (a copy: no arithmetic on the accumulated values). */
int nxhip_mgpu_gather(nxhip_ctx *ctx);
int nxhip_mgpu_read_rgba8(nxhip_ctx *ctx, uint32_t *dst);        /* rank 0: width*height uint32 */
int nxhip_mgpu_read_accumulation(nxhip_ctx *ctx, float *dst);    /* rank 0: width*height x 3 floats */
int nxhip_mgpu_shutdown(nxhip_ctx *ctx);

/* D_QueueSize after the last rendered frame — Cuda/PathTracer/PathTracer.cuh:61-73.  Each array NX_PATH_MAX_LENGTH ints. */
typedef struct nxhip_queue_sizes {
    int32_t traceSize[NX_PATH_MAX_LENGTH];
    int32_t traceShadowSize[NX_PATH_MAX_LENGTH];
    int32_t diffuseSize[NX_PATH_MAX_LENGTH];
    int32_t plasticSize[NX_PATH_MAX_LENGTH];
    int32_t dielectricSize[NX_PATH_MAX_LENGTH];
    int32_t conductorSize[NX_PATH_MAX_LENGTH];
} nxhip_queue_sizes;
int nxhip_read_queue_sizes(nxhip_ctx *ctx, nxhip_queue_sizes *out);

/* PathTracer::SetPixelQuery / GetSelectedInstance — PathTracer.cpp:310-317, PathTracer.h:25 */
int nxhip_set_pixel_query(nxhip_ctx *ctx, uint32_t x, uint32_t y);
int nxhip_get_selected_instance(nxhip_ctx *ctx, int32_t *instanceIdx);

/* ---- kernel-level hooks (tests, bench) ----------------------------------------------------------- */

/* TraceKernel on a caller-supplied ray batch — Cuda/BVH/BVH8Traversal.cuh:148-322.  Host buffers. */
int nxhip_trace_batch(nxhip_ctx *ctx, const nx_ray *rays, uint32_t count, nx_hit *hits);
/* TraceShadowKernel's any-hit test — BVH8Traversal.cuh:326-518.  occluded[i] = 1 if blocked within tmax[i]. */
int nxhip_trace_shadow_batch(nxhip_ctx *ctx, const nx_ray *rays, const float *tmax, uint32_t count, uint8_t *occluded);

/* Test hook: overwrite ONE node of an uploaded BLAS in device memory WITHOUT the checks of nxhip_upload_blas.  Exists so that
 * the trace kernels' behaviour on a BVH that is not a tree (a child that points back at its parent) can be tested: they must
 * abandon such rays and nxhip_sync must report NXHIP_ERR_TRAVERSAL (nx_device.h kStallLimit) instead of never returning.
 * The node's child / leaf ranges must still lie inside the BLAS (that part is checked: a wild index is a wild device read). */
int nxhip_debug_write_blas_node(nxhip_ctx *ctx, int32_t blasId, uint32_t nodeIdx, const nx_bvh8_node *node);
/* Test hook: the number of passes every slot has begun since its ordered-compaction status words were last cleared.  The words
 * are tagged with that number instead of being cleared per launch (NX_COMPACT_ORDERED); it wraps after 2^20 - 1 passes — a
 * viewer at a thousand one-frame passes per second gets there in 17 minutes — where the words are cleared in stream order and
 * the count starts over.  The test sets it just below the limit and renders across the wrap. */
int nxhip_debug_set_scan_epoch(nxhip_ctx *ctx, uint32_t epoch);
/* Test hook: the trace kernels hand the same rays out again and again (a ray that re-queues itself — what an in-kernel restart did in
 * round 5 until the host's watchdog ended the process).  The per-launch progress check must end every such launch with
 * NXHIP_ERR_TRAVERSAL from nxhip_sync within milliseconds. */
int nxhip_debug_set_requeue(nxhip_ctx *ctx, int on);

/* Kernel-level test hooks for the shading functions (same role as nxhip_trace_batch for the traversal): run the device
 * BSDF sample / eval (the headers of Cuda/BSDF/ as restated in nx_bsdf.h) and the software texture fetch on host arrays.
 * sample: uses wi and rng (the xorshift state before the call); eval: uses wi and wo.  Directions are in the local
 * shading frame (z = normal).  `material` is one nx_material; its type selects the BSDF (a CONDUCTOR runs the extended
 * conductor BSDF whatever the context's conductor mode).  nx_bsdf_query / nx_bsdf_result: nexus_pod.h. */
int nxhip_bsdf_sample_batch(nxhip_ctx *ctx, const nx_material *material, const nx_bsdf_query *queries, uint32_t count, nx_bsdf_result *results);
int nxhip_bsdf_eval_batch(nxhip_ctx *ctx, const nx_material *material, const nx_bsdf_query *queries, uint32_t count, nx_bsdf_result *results);
/* kind as in nxhip_upload_texture (0 diffuse, 1 emissive, 2 hdr; textureId ignored for hdr); uv = 2*count floats,
 * rgba = 4*count floats (sRGB-decoded, bilinear, wrap addressing: what the shade kernels see). */
int nxhip_tex2d_batch(nxhip_ctx *ctx, int kind, int textureId, const float *uv, uint32_t count, float *rgba);

/* The transcendental functions of the shading path (include/nexus_fmath.h: the ONE text the kernels and the CPU oracle both
 * compile — sin / cos / exp / log / pow / atan2 / asin replacing the libm calls of Random.cuh:119-121, Microfacet.cuh:18,75,
 * PathTracer.cu:65-83, Utils.h:51-54) on host arrays: out[i] = nxf_apply(op, a[i], b[i]), op = NXF_OP_*; b may be NULL for the
 * one-argument functions.  tests/test_fmath.py compares the device's results with the oracle's bit for bit. */
int nxhip_fmath_batch(nxhip_ctx *ctx, int op, const double *a, const double *b, uint32_t count, double *out);

/* Visit counters of the two trace kernels (algorithmic bytes for the roofline).  When enabled
 * the trace kernels run their counting variant; off by default. */
typedef struct nxhip_trace_stats {
    uint64_t rays, nodes, tris, instances;
    /* SIMD-efficiency diagnostics: traversal-loop iterations summed over waves, and the number of lanes that were
     * busy / took a node step / took a primitive step in them (ideal: 64 per iteration). */
    uint64_t waveIters, lanesActive, lanesNode, lanesPrim;
    /* shader-clock cycles per loop section summed over waves: 0 refill, 1 pop/retire, 2 node select+fetch, 3 node
     * decode, 4 instance entry, 5 triangle fetch+test, 6-7 unused */
    uint64_t cycles[8];
} nxhip_trace_stats;
int nxhip_enable_trace_stats(nxhip_ctx *ctx, int enable);
int nxhip_read_trace_stats(nxhip_ctx *ctx, nxhip_trace_stats *closest, nxhip_trace_stats *shadow, int reset);

/* Per-kernel-class device time.  enable = 1: frames are launched kernel by kernel on the context's stream with a
 * hipEvent pair around every launch (no graph, no overlap between the trace and shadow-trace kernels).  enable = 2: the
 * frame graph is rebuilt with an event-record node before and after every kernel node, so each kernel is timed under the
 * conditions of the production replay (closest-hit and shadow traces of a bounce run concurrently); the events are read
 * after every replay.  enable = 3: the same graph, but replays are NOT separated by a sync: nxhip_read_kernel_times then
 * returns the kernels of the LAST replay only, timed under the sustained clocks of a back-to-back series.  0: off.  Classes: 0 generate, 1 trace, 2 shadow, 3 logic, 4 shade, 5 accumulate,
 * 6 thin (the launch behind the two trace launches of a level that finishes the last long rays of their dry waves: nx_trace.hip thin_kernel). */
enum { NXHIP_K_GENERATE = 0, NXHIP_K_TRACE = 1, NXHIP_K_SHADOW = 2, NXHIP_K_LOGIC = 3, NXHIP_K_SHADE = 4, NXHIP_K_ACCUMULATE = 5, NXHIP_K_THIN = 6, NXHIP_K_COUNT = 7 };
typedef struct nxhip_kernel_times {
    double ms[NXHIP_K_COUNT];
    uint64_t launches[NXHIP_K_COUNT];
} nxhip_kernel_times;
int nxhip_enable_kernel_timing(nxhip_ctx *ctx, int enable);
int nxhip_read_kernel_times(nxhip_ctx *ctx, nxhip_kernel_times *out, int reset);
/* With enable = 2 or 3: the kernels of the LAST timed replay one by one, in graph order — class (NXHIP_K_*), start relative to
 * the replay's first kernel and duration, both in ms — so that a caller can set the launches of one pass beside that pass's wall
 * time (bench.py: the nine closest-hit launches of the repetition its roofline is computed from).  Up to `capacity` entries are
 * written, *count receives the number of kernels in the graph.  Call it before nxhip_read_kernel_times(reset). */
int nxhip_read_graph_timeline(nxhip_ctx *ctx, int32_t *klass, float *startMs, float *durationMs, uint32_t capacity, uint32_t *count);

/* Build-time facts for tests: 1 if the library was compiled with device code for gfx950. */
int nxhip_has_gfx950_code(void);
/* Build facts as bits: 1 = device code for gfx950, 2 = built with the Makefile's scheduler flags, 4 = nxhip_debug_* hooks compiled in
 * (the default; a `make release` library keeps the symbols and refuses the calls). */
int nxhip_build_info(void);

/* ---- ABI stamp ---------------------------------------------------------------------------------------
 * A library that does not match its caller — a stale build picked up through NEXUS_AMD_LIB / LD_LIBRARY_PATH, a header of
 * another version — used to show as a GPU memory fault in the first launch (round 3: gpurun_out/r3_07).
 * Now it is an error status before anything is launched.
 *   nxhip_header_abi_stamp()  what THIS header says, compiled into the caller: API version + size and key offsets of every
 *                             struct that crosses the boundary, FNV-1a hashed;
 *   nxhip_abi_stamp()         the same function as compiled into the library;
 *   nxhip_check_library(s)    NXHIP_OK if s equals the library's stamp AND every translation unit of the library was compiled
 *                             with the same device-side layouts (DeviceState, Counters, InstTrav, record strides ...: a library
 *                             linked from objects of different source states is refused), else NXHIP_ERR_ABI + message.
 * nxhip_create performs the library-internal half by itself.  C / C++ callers: nxhip_check_library(nxhip_header_abi_stamp())
 * once after loading; the Python binding does the equivalent with the sizes of its own ctypes / numpy mirrors. */
uint64_t nxhip_abi_stamp(void);
int nxhip_check_library(uint64_t callerStamp);

static inline uint64_t nxhip_abi_mix(uint64_t h, uint64_t v)
{
    int i;
    for (i = 0; i < 8; i++) { h = (h ^ (v & 0xffu)) * 0x100000001b3ull; v >>= 8; }
    return h;
}
static inline uint64_t nxhip_header_abi_stamp(void)
{
    /* (the list the Python binding mirrors: nexus_amd/capi.py abi_words) */
    const uint64_t w[] = {
        NXHIP_API_VERSION, NX_PATH_MAX_LENGTH,
        sizeof(nx_bvh8_node), offsetof(nx_bvh8_node, meta), sizeof(nx_triangle), offsetof(nx_triangle, texCoord0),
        sizeof(nx_bvh_instance), offsetof(nx_bvh_instance, transform), offsetof(nx_bvh_instance, materialId),
        sizeof(nx_material), offsetof(nx_material, emissive), offsetof(nx_material, type), sizeof(nx_light), offsetof(nx_light, type),
        sizeof(nx_camera), offsetof(nx_camera, resolution), sizeof(nx_render_settings), offsetof(nx_render_settings, backgroundColor),
        sizeof(nx_ray), sizeof(nx_hit), sizeof(nx_bsdf_query), sizeof(nx_bsdf_result), offsetof(nx_bsdf_result, rngOut),
        sizeof(nxhip_queue_sizes), sizeof(nxhip_trace_stats), offsetof(nxhip_trace_stats, cycles), sizeof(nxhip_kernel_times), NXHIP_K_COUNT,
    };
    uint64_t h = 0xcbf29ce484222325ull;
    size_t i;
    for (i = 0; i < sizeof w / sizeof w[0]; i++) h = nxhip_abi_mix(h, w[i]);
    return h;
}

#ifdef __cplusplus
}
#endif
#endif /* NEXUS_HIP_H */
