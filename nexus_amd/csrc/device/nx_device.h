// nx_device.h — device-resident state of one context: scene tables, ray queues, counters.
//
// Layout in HBM (all 16-byte aligned, float4-packed so every lane moves 16 B per access):
//   BLAS k   : nodes   uint4[5*nodeCount]   the 80-byte CWBVH nodes as uploaded (reference D_BVH8Node)
//              isect   float4[3*triCount]   leaf-ordered intersection stream {p0,origTriIdx},{e0,0},{e1,0}:
//                                           built at upload so the trace kernel needs no triangleIdx
//                                           indirection and reads 48 B instead of 4 + 96
//              tris    nx_triangle[triCount] original 96-byte triangles (shading only)
//   TLAS     : nodes, instIdx u32[], instTrav InstTrav[instanceCount] (160 B: inverse transform rows, the BLAS root node + the
//              BLAS pointers, replaces the reference's 160-B instance + 32-B D_BVH8 double fetch),
//              instances nx_bvh_instance[] (shading only)
//   queues   : trace   rayO float4 (origin, -), rayD float4 (direction, pixelIdx), hit float4 (t,u,v,triIdx),
//                      hitInst u32            — reference D_TraceRequestSOA, Cuda/PathTracer/PathTracer.cuh:32-37
//              shadow  rayO float4 (origin, tmax), rayD float4 (direction, pixelIdx), rad float4
//                                             — D_ShadowTraceRequestSOA, PathTracer.cuh:39-45
//              material[4] hit float4 (pathIdx, u, v, triIdx), dirInst float4 (direction, instanceIdx)
//                                             — D_MaterialRequestSOA, PathTracer.cuh:47-52
//              path    throughput float4 (rgb, lastPdf), radiance float4, rayOrigin float4
//                                             — D_PathStateSOA, PathTracer.cuh:19-30
//   counters : per bounce queue sizes (reference D_QueueSize) + 8 per-XCD fetch heads per trace launch.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "nexus_pod.h"

// Pointers stored in device-resident structs are global-memory pointers.  Telling the device compiler so
// (address_space(1)) turns every access through them into global_load/store instead of flat_load/store (a pointer
// loaded from memory is otherwise "generic" and pays the flat path: LDS-aperture check, both wait counters).
// The host (and nxhip_api.hip, which only fills the structs) sees plain pointers of the same size; kernel translation
// units define NX_KERNEL_TU before including this header.
#if defined(__HIP_DEVICE_COMPILE__) && defined(NX_KERNEL_TU)
#define NX_G __attribute__((address_space(1)))
#else
#define NX_G
#endif

namespace nxd {

// Record strides of the device-side BVH arrays, in 16-byte units.  The payload is always 5 / 3 chunks (80-B node, 48-B
// triangle record); a larger stride pads every record so that it never straddles a 64-B sector / 128-B line.
#ifndef NX_NODE_STRIDE
#define NX_NODE_STRIDE 5
#endif
#ifndef NX_TRI_STRIDE
#define NX_TRI_STRIDE 3
#endif
// Stride of the SHADING triangles (nx_triangle, 96 B of payload) in bytes.  96: the array as uploaded — a record then spans 1.5
// 128-byte lines on average.  128: every record on a line of its own (+33 % memory for that array).
#ifndef NX_SHADE_TRI_STRIDE
#define NX_SHADE_TRI_STRIDE 96
#endif
constexpr int kShadeTriStride = NX_SHADE_TRI_STRIDE;
static_assert(kShadeTriStride >= (int)sizeof(nx_triangle) && kShadeTriStride % 16 == 0, "shading triangle stride");
constexpr int kNodeStride = NX_NODE_STRIDE;
constexpr int kTriStride = NX_TRI_STRIDE;

constexpr int kXcds = 8;        // MI355X: 8 XCDs, each with its own L2
constexpr int kMaterialTypeOffset = 56;  // nx_material::type (int8) and, in the device copy only, a derived flag in the
constexpr int kMaterialFlagOffset = 57;  // padding byte behind it (nxhip_set_materials)
static_assert(sizeof(nx_material) == 60 && offsetof(nx_material, type) == kMaterialTypeOffset, "nx_material layout");
constexpr int kEnvGuide = 64;   // buckets of the environment sampler's cdf search guides
constexpr int kWave = 64;       // CDNA wavefront
constexpr int kMaxBounceSlots = NX_PATH_MAX_LENGTH;
constexpr int kScanKinds = 5;   // kernels that hand out queue slots: logic + the four material kernels
constexpr int kScanWords = 4;   // status words per tile of such a kernel: one per queue it appends to (logic 4, material 2)

struct BlasDev {
    const NX_G uint4* nodes;
    const NX_G float4* isect;
    const NX_G nx_triangle* tris;   // shading triangles, kShadeTriStride bytes apart (shade_tri below)
    const NX_G uint32_t* triIdx;
    uint32_t nodeCount, triCount;
    uint32_t pad_[2];
};
static_assert(sizeof(BlasDev) == 48, "BlasDev layout");
inline __device__ const NX_G nx_triangle* shade_tri(const NX_G nx_triangle* tris, uint32_t k) { return (const NX_G nx_triangle*)((const NX_G char*)tris + (size_t)k * (size_t)kShadeTriStride); }

// Traversal record of one TLAS leaf, stored in TLAS *leaf order* (entry k belongs to tlasInstIdx[k]) so that entering an
// instance is one fetch with no index indirection: inverse transform rows, the BLAS arrays, the instance id, the BLAS root.
struct __attribute__((aligned(16))) InstTrav {
    float4 r0, r1, r2;  // rows 0..2 of invTransform: first, because a transformed entry needs them first (the lane starts the
                        // transform while the rest of the record and the BLAS root are still arriving)
    const NX_G uint4* nodes;
    const NX_G float4* isect;
    uint32_t instIdx;
    uint32_t flags;     // kInstIdentity: the rows above are exactly those of the identity matrix (kept per record by the host
                        // upload and the device refit; the kernels use the scene-wide DeviceState::sceneFlags derived from it)
    uint32_t pad_[2];
    uint4 root[5];      // a copy of the BLAS's root node: entering the instance and testing its root are one loop iteration
};
static_assert(sizeof(InstTrav) == 160, "InstTrav layout");
constexpr uint32_t kInstIdentity = 1u;
constexpr uint32_t kSceneAllIdentity = 1u;  // DeviceState::sceneFlags
// rows 0..2 of a row-major 4x4 matrix equal the identity's bit for bit (+0.0 and 1.0 exactly; -0.0 does not count)
inline __host__ __device__ bool rows_are_identity(const float* m)
{
    bool same = true;
    for (int i = 0; i < 12; i++) {
        union { float f; uint32_t u; } v;
        v.f = m[i];
        same = same && v.u == ((i % 5 == 0) ? 0x3f800000u : 0u);
    }
    return same;
}

// Shading record of one instance, in instance order: everything the logic and material kernels need of an instance behind ONE
// dependent load — the reference follows instance -> BVH descriptor -> triangle array and instance -> material (PathTracer.cu:
// 329-347), four dependent round trips for a kernel whose time is the sum of its round trips.  Built by the host from the
// instance, BLAS and material tables (nxhip_api.hip refresh_shade_inst) whenever one of them changes; the device-side
// transform update (nx_refit.hip) keeps the two matrices current.  Same values as the tables: nothing is recomputed.
struct __attribute__((aligned(16))) ShadeInst {
    float transform[12];     // rows 0..2 of nx_bvh_instance::transform
    float invTransform[12];  // rows 0..2 of nx_bvh_instance::invTransform
    const NX_G nx_triangle* tris;  // BlasDev::tris of the instance's BLAS (kShadeTriStride apart)
    uint32_t triCount;
    int32_t materialId;
    nx_material material;    // the device copy (derived flag byte behind `type` included: kMaterialFlagOffset)
    uint32_t pad_;
};
static_assert(sizeof(ShadeInst) == 176 && offsetof(ShadeInst, material) == 112, "ShadeInst layout");

struct TextureDev {
    const NX_G uint32_t* texels;  // RGBA8, row 0 first
    uint32_t width, height;
};

// The rays of the trace queue: what a material kernel writes and the trace kernel reads.
struct TraceRays {
    NX_G float4* rayO;     // origin; w: kRayPassThrough | kRaySurvives (bit patterns, below)
    NX_G float4* rayD;     // direction; w: path index
    NX_G float4* tp;       // the path's throughput (rgb) and last pdf (w) travel with its ray: see MaterialQueue::tp
};
// Two sets of rays: the SCAN pipeline's material kernels of bounce b read the queue of bounce b - 1 (hit, ray direction, path
// state) while they write the rays of bounce b, so the rays alternate between the two sets by bounce parity (rays[b & 1]); the hit
// records need no second set (written by trace(b), read by the material kernels of b + 1, which have finished before trace(b + 1)
// starts).  The CLASSIC pipeline (logic kernel + material queues: the reference's copy, PathTracer.cu:183-206) uses rays[0] only.
struct TraceQueue {
    TraceRays rays[2];
    NX_G float4* hit;        // (t, u, v, triangle)
    NX_G uint32_t* hitInst;  // instance of the hit | what the path does next << kHitCodeShift (SCAN pipeline; classic: the instance alone)
};
// rayO.w of a trace-queue ray, as a bit pattern
constexpr uint32_t kRayPassThrough = 1u;  // a pass-through continuation: the path's previous vertex stays what it was (keep_previous_vertex)
constexpr uint32_t kRaySurvives = 2u;     // SCAN pipeline: the Russian-roulette draw of the NEXT logic step (PathTracer.cu:167-175), made by the ray's
                                          // producer — the draw depends on the path's throughput and on a random number keyed by pixel (or slot), bounce
                                          // and frame, all known when the ray is made, and only counts if the ray hits
// hitInst word: the instance in the low bits, a code above them.  InstTrav::instIdx carries the instance's material type + 1 there
// (inst_code_kernel), so the closest-hit kernel has "which material kernel shades this hit" in the register that holds the hit's
// instance anyway.  0: nothing to shade (Russian roulette ended the path, or no kernel for the type); 1 .. 4: NX_MAT_* + 1; 7: a miss.
constexpr int kHitCodeShift = 29;
constexpr uint32_t kHitInstMask = (1u << kHitCodeShift) - 1u;
constexpr uint32_t kHitCodeMiss = 7u;
constexpr int kScanMiss = 4;  // the misses' place among the types of shade_scan_kernel (typeMask bit, ticket row)
// the trace kernels' `bounce` argument: bounce | kTraceScanFlag = the SCAN pipeline's launch (ray set by bounce parity, codes in
// the hit records); without the flag: rays[0], plain hit records (classic pipeline, ray-batch hooks)
constexpr int kTraceScanFlag = 0x100;
// ... | kTraceEntryFlag: the primary launch of a pass whose rays carry the number of their ENTRY STATE in rayO.w (bits 2 and up,
// 0 = none): the traversal of a run of 64 consecutive paths starts from the state their first node steps provably share (below)
constexpr int kTraceEntryFlag = 0x200;
constexpr int kRayEntryShift = 2;
// ... | kTraceThinFlag: a dry wave of the launch may hand its last long rays to the thin kernel (nx_trace.hip thin_kernel)
constexpr int kTraceThinFlag = 0x400;
// thin_kernel's argument: ... | kThinClosestOnly / kThinAnyOnly = one of the two lists only (the pass graph gives each trace launch of a
// level its own thin launch, so that the closest-hit rays' searches run beside whatever the any-hit launch still has to do)
constexpr int kThinClosestOnly = 0x800, kThinAnyOnly = 0x1000;

// The traversal state of a ray a dry trace wave hands to thin_kernel (nx_trace.hip, round 6): everything the loop keeps per lane at its
// loop top, so that the wave-wide search CONTINUES the ray instead of starting it again at the TLAS root — the stack's groups and the
// two current ones become the search's first work items, the hit found so far its bound.  (The ray itself is read from its queue slot;
// inside a transformed instance the search derives the ray in that frame from the instance record, as it does for any item.)
struct alignas(16) ThinState {
    uint2 ng, tg;            // the current node group / primitive group (loop top)
    float hitT, hitU, hitV;  // closest hit so far (1e30: none); any hit: hitT = the ray's tmax, the rest unused
    uint32_t hitTri, hitInst;
    int32_t sp, instSp;      // stack entries; entries [0, instSp) belong to the TLAS frame (instSp < 0: the ray is in the TLAS frame)
    uint32_t leaf;           // instSp >= 0: the instance record (TLAS leaf) the ray is inside
    uint32_t pad_[4];
    uint2 stack[32];         // kLdsDepth + kSpillDepth entries, bottom first
};
static_assert(sizeof(ThinState) == 320 && offsetof(ThinState, stack) == 64, "64-byte header + 32 stack entries");

// Entry state of a run of 64 consecutive primary paths (nx_entry.hip).  The 64 rays of an 8 x 8 pixel tile visit the same nodes
// with the same hit masks for their first ~4 of ~10 node steps (tools/entry_point_probe.py) — the same arithmetic done 64 times
// for one answer.  entry_state_kernel walks those steps ONCE per run with a conservative test of the run's whole ray bundle
// against every child box (hit by all rays of the bundle / missed by all / undecided: stop), and records the loop-top state the
// traversal has reached — stack, node group, leaf group, instance — which the closest-hit kernel installs at refill instead of
// starting at the root.  Every step it takes is one whose outcome is the same for every ray the bundle can contain, so each
// ray's hit record is what the full traversal gives (bit for bit: tests/test_gpu_entry.py); only the visit counts drop.
constexpr int kEntryMaxStack = 6;
struct __attribute__((aligned(16))) EntryState {
    uint2 stack[kEntryMaxStack];
    uint2 ng, tg;
    int32_t sp;        // stack entries in use
    int32_t instSp;    // -1: in the TLAS
    int32_t leafSlot;  // instance record (TLAS leaf order) the state is inside of, -1: none
    int32_t steps;     // node steps taken (0: the state is the root's; the kernel then starts as usual)
};
static_assert(sizeof(EntryState) == 80, "EntryState layout");
struct ShadowQueue {
    NX_G float4* rayO;
    NX_G float4* rayD;
    NX_G float4* radiance;
};
struct MaterialQueue {
    NX_G float4* hit;      // (path index, u, v, triangle)
    NX_G float4* dirInst;  // (ray direction, instance)
    // D_PathStateSOA::throughput / lastPdf (PathTracer.cuh:19-30) are per-pixel arrays in the reference: every logic and material
    // thread reads and writes 16 bytes at its path's pixel, i.e. scattered from the first bounce on.  Here the two values ride
    // along in the queue entries instead — written and read at the slot index, coalesced — and no per-pixel copy exists.
    NX_G float4* tp;
};

// Every queue is kept in kQueueShards REGIONS of its buffer (region k = slots [k * cap, (k + 1) * cap), cap =
// DeviceState::queueShardCap) with one size word per region.  A producer kernel cuts the tiles of ITS input queue into eight
// contiguous runs and appends the outputs of run r to region r (nx_queue.h ProducerRegions), so a region holds what one stretch
// of the previous queue produced, at most M / 8 + T slots per producer kernel of M items in tiles of T; the trace kernels' eight
// fetch heads (one per XCD group) walk one region each, and their words sit on separate cache lines (RegionCounters below).
// DeviceState::queueShards == 1 (ordered compaction: one workgroup, the reference's serial slot order) keeps everything in
// region 0, whose capacity is then the whole buffer.
constexpr int kQueueShards = 8;
static_assert(kQueueShards == kXcds, "one queue region per fetch head");
constexpr int kQueueShardSlack = 4096;  // slots a region may exceed its even share by (see above; tiles are at most 2 048 items: the logic kernel's)

// Mutable per-frame words, zeroed / advanced by begin_frame_kernel.  Mirrors D_QueueSize (PathTracer.cuh:61-73) with every
// size word and traceCount / traceShadowCount widened to one per region / XCD group — and each region's words on 4 KiB of
// their own: the returning atomics on them are device-scope and execute at the memory side, where what serialises them is the
// channel an address maps to; eight words in one cache line are one channel (measured: no gain at all over a single word),
// eight words 4 KiB apart are not.
struct RegionCounters {
    int32_t traceSize[kMaxBounceSlots];
    int32_t traceShadowSize[kMaxBounceSlots];
    int32_t materialSize[4][kMaxBounceSlots];  // enum order DIFFUSE, DIELECTRIC, PLASTIC, CONDUCTOR
    int32_t traceHead[kMaxBounceSlots];        // rays of this region handed out by the closest-hit / any-hit trace launch
    int32_t shadowHead[kMaxBounceSlots];
    // SCAN pipeline (nx_wavefront.hip shade_scan_kernel): tiles of this region's trace queue handed out so far to the material
    // kernel of a type, beyond the one every workgroup takes by its rank
    int32_t scanTile[5][kMaxBounceSlots];  // [NX_MAT_*], [kScanMiss]: the misses (a fifth "type" of the one material launch)
    int32_t pad_[2048 - 13 * kMaxBounceSlots];
};
static_assert(sizeof(RegionCounters) == 8192, "one region's counters per 8 KiB");
constexpr int kRegionStride = (int)(sizeof(RegionCounters) / sizeof(int32_t));  // distance between the same word of two regions

struct Counters {
    RegionCounters region[kQueueShards];
    int32_t orderedBase[8];  // (unused since the ordered compaction became grid-wide; kept so that the words behind it stay where they were)
    int32_t tailHead;      // paths of the tail kernel's queue handed out so far
    int32_t pad_[3];
    // ordered compaction (nx_wavefront.hip OrderedScan): tiles of a logic / material launch handed out so far, one word per
    // launch of a pass — kernel kind (0 logic, 1 + NX_MAT_* material) x bounce
    int32_t scanTicket[kScanKinds][kMaxBounceSlots];
    int32_t thinCount[2][kMaxBounceSlots];  // rays handed to the thin kernel by the closest-hit [0] / any-hit [1] launch of a bounce (may exceed thinCapacity: the lists hold min(count, capacity))
};

struct FrameState {
    uint32_t frameNumber;
    int32_t pixelQueryPixel;     // -1: none (D_PixelQuery, PathTracer.cuh:54-58)
    int32_t pixelQueryInstance;
    uint32_t errorWord;          // kErrTraversalStalled: set by a trace kernel that abandoned rays (see kStallLimit), read and cleared by nxhip_sync
    uint32_t scanEpoch;          // number of this pass among the passes of its slot (begin_frame_kernel's argument): tags the tile status words of the ordered compaction
    uint32_t pad_[3];
};
constexpr uint32_t kErrTraversalStalled = 1u;
// a trace wave was handed more rays than its launch's queue holds: rays come back into the queue they were taken from (round 5's in-kernel
// restart did that and ran to the host's watchdog: gpurun_out/r5_10 — the per-refill iteration count cannot see a wave that keeps retiring and re-taking)
constexpr uint32_t kErrRaysRetaken = 4u;
constexpr uint32_t kErrScanStalled = 2u;  // a workgroup of the ordered compaction gave up waiting for a predecessor tile (never seen; the guard turns a hang into a status)
// Ordered compaction, status word of a tile and queue: {tag = launch serial << 2 | state, value}.  The launch serial (pass epoch,
// bounce, kernel kind) makes words of earlier launches read as "not there yet", so the array is never cleared between launches.
constexpr uint32_t kScanEpochLimit = 1u << 20;  // pass epochs 1 .. limit - 1; the host clears the status arrays when it wraps
constexpr uint32_t kScanAggregate = 1u, kScanPrefix = 2u;
// Material kernels under the ordered compaction run workgroups (= tiles) of 1 024 items instead of 256: a tile costs a ticket (a
// returning atomic on ONE word, 88 per microsecond chip-wide) and a look-back, so a quarter of the tiles is what pays there
// (driver command, ordered mode: 1 600 -> 1 680 Msamples/s), while the racing allocator prefers small workgroups (the same 1 024
// threads cost it 3.5 %: 2 014 -> 1 944).
constexpr int kShadeBlockOrderedThreads = 1024;
// A wave of a trace kernel that has stayed in its traversal loop for this many iterations (about a second) without coming
// through its refill point — i.e. with rays that do not finish — gives up on them:
// the rays end as misses / unoccluded, the error word is set and nxhip_sync reports NXHIP_ERR_TRAVERSAL.  A ray of a well-formed
// tree visits every record at most once, so the bound is far above anything legitimate (the longest rays of the 10 M-triangle
// scene take a few thousand iterations); it exists so that a BVH that is not a tree — which the upload checks reject, but a
// device-side builder's output never passes through them — ends in an error status instead of a kernel that never returns.
constexpr uint32_t kStallLimit = 1u << 20;

struct TraceStatsDev {
    unsigned long long rays, nodes, tris, instances;
    // SIMD-efficiency diagnostics of the counting variant: loop iterations per wave, and how many lanes were busy /
    // took the node step / took the primitive step in them
    unsigned long long waveIters, lanesActive, lanesNode, lanesPrim;
    // shader-clock cycles per section of the traversal loop, summed over waves (counting variant only):
    // 0 refill, 1 pop/retire, 2 node select+fetch, 3 node decode, 4 instance entry, 5 triangle fetch+test, 6 rest
    unsigned long long cycles[8];
};

struct DeviceState {
    // scene
    const NX_G uint4* tlasNodes;
    const NX_G uint32_t* tlasInstIdx;
    const NX_G InstTrav* instTrav;
    const NX_G nx_bvh_instance* instances;
    const NX_G ShadeInst* shadeInst;   // [instanceCount], see ShadeInst
    const NX_G BlasDev* blas;
    const NX_G nx_material* materials;
    const NX_G nx_light* lights;
    const NX_G TextureDev* diffuseMaps;
    const NX_G TextureDev* emissiveMaps;
    const NX_G float* srgbLut;  // 256 floats
    TextureDev hdrMap;     // texels == nullptr: flat background
    uint32_t lightCount;
    uint32_t instanceCount;
    uint32_t sceneFlags;   // kSceneAllIdentity: every instance record carries kInstIdentity (the trace kernels then skip the transform rows)
    nx_camera camera;
    nx_render_settings settings;
    int32_t rngMode, compactMode, conductorMode;
    // environment importance sampling (extension, nxhip_set_env_sampling): piecewise-constant distribution over the texels of
    // hdrMap — marginal cdf over rows [height], conditional cdf per row [height][width], pdf per solid angle x cos(latitude)
    int32_t envSampling;
    const NX_G float* envMarginalCdf;
    const NX_G float* envRowCdf;
    const NX_G float* envDensity;
    // search guides: entry b of a guide = the first index whose cdf exceeds b / kEnvGuide (kEnvGuide + 1 entries per cdf), so
    // that inverting a cdf at r starts from the bracket of bucket floor(r * kEnvGuide) instead of the whole array
    const NX_G uint32_t* envMarginalGuide;
    const NX_G uint32_t* envRowGuide;
    // paths: a pass renders framesPerPass consecutive frames at once; path p belongs to frame slice p / localCount
    // and to local pixel p % localCount.  Batching frames keeps every kernel large (the tail of a trace launch is
    // set by its slowest ray) and uses HBM capacity instead of launches.
    uint32_t localCount;       // pixels rendered by this context
    uint32_t framesPerPass;    // S >= 1
    uint32_t pathCount;        // localCount * framesPerPass
    uint32_t queueShards;      // regions in use: kQueueShards, or 1 with ordered compaction
    uint32_t queueShardCap;    // slots per region
    const NX_G uint32_t* pixelMap;  // local -> global pixel, nullptr = identity
    NX_G float4* radiance;
    NX_G float4* rayOrigin;
    NX_G float4* accumulation;
    NX_G uint32_t* rgba8;
    TraceQueue trace;
    ShadowQueue shadow;
    MaterialQueue material[4];
    NX_G Counters* counters;
    NX_G FrameState* frame;
    NX_G TraceStatsDev* traceStats;  // [0] closest, [1] shadow
    NX_G unsigned long long* scanStatus;  // [tiles of the largest queue][kScanWords]: ordered compaction (nx_wavefront.hip OrderedScan)
    const NX_G uint32_t* leafOfInstance;  // [instanceCount]: the TLAS leaf (= index of the InstTrav record) of every instance
    NX_G ThinState* thinStates;      // [2][thinCapacity] (closest hit, any hit): the traversal state that goes with entry k of the lists below
    NX_G EntryState* entry;          // [ceil(localCount / 64)]: entry states of the primary rays' runs, nullptr: off (nxhip_set_entry_points)
    // the last long rays of dry trace waves, handed to thin_kernel (nx_trace.hip): queue slots (closest hit: | roulette bit 31)
    NX_G uint32_t* thinClosest;
    NX_G uint32_t* thinAny;
    uint32_t thinCapacity;           // entries per list
    uint32_t thinLanes, thinIters;   // hand-over rule: at most thinLanes busy lanes for at least thinIters iterations (16 / 16; a test hook sets 64 / 0: every ray of a dry wave)
    uint32_t thinPoolLimit;          // items a thin wave's pool may hold before a round puts items back (0: all of it; a test hook lowers it: nxhip_debug_set_thin_pool)
    uint32_t entryRuns;              // states in `entry`
    uint32_t debugRequeue;           // test hook (nxhip_debug_set_requeue): the trace kernels hand the same rays out again and again — a ray that re-queues itself (see kErrRaysRetaken)
};

// What a translation unit of the library believes about the device-resident structures and the compile-time knobs that shape
// them.  Every .hip file exports the value it was compiled with (layout_stamp_<unit>()); nxhip_create compares them, so a
// library linked from objects of different source states — host code filling a DeviceState the kernels read with other
// offsets: a wild device access in the first launch — is refused with NXHIP_ERR_ABI instead.
constexpr uint64_t layout_mix(uint64_t h, uint64_t v)
{
    for (int i = 0; i < 8; i++) { h = (h ^ (v & 0xffu)) * 0x100000001b3ull; v >>= 8; }
    return h;
}
constexpr uint64_t layout_stamp()
{
    uint64_t h = 0xcbf29ce484222325ull;
    const uint64_t w[] = {
        sizeof(DeviceState), offsetof(DeviceState, camera), offsetof(DeviceState, envSampling), offsetof(DeviceState, localCount), offsetof(DeviceState, pixelMap),
        offsetof(DeviceState, radiance), offsetof(DeviceState, trace), offsetof(DeviceState, shadow), offsetof(DeviceState, material), offsetof(DeviceState, counters),
        offsetof(DeviceState, frame), offsetof(DeviceState, traceStats), offsetof(DeviceState, scanStatus), offsetof(DeviceState, entry), offsetof(DeviceState, entryRuns), offsetof(DeviceState, debugRequeue), offsetof(DeviceState, thinStates), offsetof(DeviceState, leafOfInstance), sizeof(ThinState), offsetof(ThinState, stack), offsetof(DeviceState, thinClosest), offsetof(DeviceState, thinCapacity), offsetof(DeviceState, thinIters), offsetof(DeviceState, thinPoolLimit), offsetof(Counters, thinCount), sizeof(EntryState), offsetof(EntryState, sp), offsetof(Counters, scanTicket), offsetof(FrameState, scanEpoch),
        (uint64_t)kScanKinds, (uint64_t)kScanWords, (uint64_t)kScanEpochLimit, (uint64_t)kShadeBlockOrderedThreads,
        sizeof(Counters), sizeof(RegionCounters), offsetof(RegionCounters, traceShadowSize), offsetof(RegionCounters, materialSize), offsetof(RegionCounters, traceHead),
        offsetof(RegionCounters, shadowHead), offsetof(RegionCounters, scanTile), offsetof(Counters, orderedBase), offsetof(Counters, tailHead),
        sizeof(ShadeInst), offsetof(ShadeInst, tris), offsetof(ShadeInst, material), offsetof(DeviceState, shadeInst), sizeof(InstTrav), offsetof(InstTrav, nodes), offsetof(InstTrav, instIdx), offsetof(InstTrav, root), sizeof(BlasDev), offsetof(BlasDev, nodeCount),
        sizeof(TextureDev), sizeof(TraceQueue), sizeof(ShadowQueue), sizeof(MaterialQueue), sizeof(FrameState), sizeof(TraceStatsDev),
        (uint64_t)kNodeStride, (uint64_t)kTriStride, (uint64_t)kShadeTriStride, (uint64_t)kQueueShards, (uint64_t)kQueueShardSlack, (uint64_t)kRegionStride, (uint64_t)kMaxBounceSlots,
        (uint64_t)kEnvGuide, (uint64_t)kHitCodeShift, sizeof(TraceRays), offsetof(TraceQueue, hit), (uint64_t)kMaterialTypeOffset, sizeof(nx_material), sizeof(nx_bvh_instance), sizeof(nx_triangle), sizeof(nx_light), sizeof(nx_camera),
    };
    for (uint64_t v : w) h = layout_mix(h, v);
    return h;
}

}  // namespace nxd
