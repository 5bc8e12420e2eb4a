// nx_context.h — host-side bookkeeping of one device context (internal to the device layer).
#pragma once

#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "nexus_hip.h"
#include "nx_device.h"

namespace nxd {

void set_error(const std::string& msg);
bool hip_ok(hipError_t e, const char* what, const char* file, int line);

#define NX_HIP(call)                                                              \
    do {                                                                          \
        if (!::nxd::hip_ok((call), #call, __FILE__, __LINE__)) return NXHIP_ERR_HIP; \
    } while (0)

// Owning device allocation — or, with `pool` set, a VIEW of a range of one that several buffers share (the BLASes of one
// nxhip_build_blas_batch call live in four pooled allocations instead of 4 000: a thousand hipMalloc calls alone would cost
// more than the build).  The pool is freed when its last view goes.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    std::shared_ptr<DevBuf> pool;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), pool(std::move(o.pool)) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept
    {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; pool = std::move(o.pool); o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release()
    {
        if (pool) pool.reset();
        else if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    // (Re)allocate; contents undefined.  Returns false on failure (error string set).
    bool alloc(size_t n);
    // `n` bytes at `offset` of a shared allocation
    static DevBuf view(const std::shared_ptr<DevBuf>& of, size_t offset, size_t n)
    {
        DevBuf v;
        v.p = static_cast<char*>(of->p) + offset;
        v.bytes = n;
        v.pool = of;
        return v;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct BlasHost {
    DevBuf nodes, isect, tris, triIdx;
    DevBuf shadeTris;  // kShadeTriStride != 96: the triangles again, one per stride (what BlasDev::tris then points at; `tris` stays the packed array the builders read)
    uint32_t nodeCount = 0, triCount = 0;
    uint4 root[5] = {};      // host copy of node 0 (embedded in every InstTrav record of this BLAS), read back on first use
    bool rootKnown = false;
};

struct TextureHost {
    DevBuf texels;
    uint32_t width = 0, height = 0;
};

struct KernelTimer {
    hipEvent_t start = nullptr, stop = nullptr;
};

}  // namespace nxd

namespace nxd {

// Everything ONE in-flight pass owns: its path state and queues, its device-state block (scene pointers + these queues),
// counters, frame words, the stream it runs on and its instance of the pass graph.  A context is slot 0 (on the context's
// own stream); nxhip_set_passes_in_flight(R > 1) adds R - 1 more so that consecutive passes overlap on the GPU: the drain
// phase of one pass (a few long rays, most SIMDs idle) is filled by the bulk of the next.
struct PassSlot {
    hipStream_t stream = nullptr;
    bool ownsStream = false;
    DevBuf radiance, rayOrigin;
    DevBuf trRayO, trRayD, trHit, trHitInst, trTp;
    DevBuf trRayO2, trRayD2, trTp2;  // the second set of rays (TraceQueue::rays[1]): SCAN pipeline only
    bool queuesScan = false;         // which pipeline the buffers above were allocated for (scan: no material queues; classic: no second ray set)
    DevBuf shRayO, shRayD, shRadiance;
    DevBuf mqHit[4], mqDirInst[4], mqTp[4];
    DevBuf counters, frame, dState;
    DevBuf scanStatus;        // ordered compaction: tile status words (allocated with the queues)
    DevBuf thinStates;        // 2 x kThinListEntries ThinState: the traversal state of every listed ray
    DevBuf thinLists;         // 2 x kThinListEntries queue slots: the rays the trace launches of a level hand to the thin kernel (allocated with the queues)
    uint32_t scanEpoch = 0;   // passes begun in this slot since the status words were last cleared
    // Entry states of the primary rays' runs (nx_entry.hip), [ceil(localCount / 64)], allocated when entry points are on.  One table per
    // SLOT: the slot's pass graph writes it (entry_state_kernel, beside generate) and the slot's own primary closest-hit launch reads
    // it, so passes in flight never write a table another pass is reading.  Kernels find it through the slot's DeviceState (entry,
    // entryRuns), never through a pointer fixed into a graph node.
    DevBuf entryTable;
    uint32_t entryRuns = 0;
    // the slot's error word as the pass left it, copied into pinned host memory behind every pass on the slot's own stream: what
    // nxhip_sync reads instead of a blocking 4-byte device read per slot (errorFresh: no launch since that could have set it)
    uint32_t* hostError = nullptr;
    bool errorFresh = false;
    size_t pathCapacity = 0;  // paths this slot's queue buffers hold right now; 0: released (nxhip_ctx::queueCapacity is the nominal size)
    // Instances of the pass graph, one per SHAPE it has been asked for (see serial_shade, trace_blocks, tail_bounce in
    // nxhip_api.hip: a small pass, a large pass and a pass among several in flight are different graphs).  A pass of another
    // size class replays the instance built for that class instead of re-instantiating one inside the frame loop.
    struct GraphInstance {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        bool serialShade = false;
        int traceBlocks = 0, tailBounce = 0;
        int flavor = 0;  // pass_flavor(): pipeline, miss kernel, logic kernel variant
    };
    std::vector<GraphInstance> graphs;
    // pass bookkeeping (slots >= 1 and slot 0 alike)
    hipEvent_t done = nullptr, accumulated = nullptr;
    bool awaitingAccumulate = false;   // holds a rendered pass that nxhip_accumulate has not consumed yet
    bool accumulateRecorded = false;   // `accumulated` was recorded after the last use: the next use must wait for it
    uint32_t frames = 0, frameLast = 0;  // of the pass it holds
    DeviceState view{};  // host mirror of this slot's device-state block
};

}  // namespace nxd

struct nxhip_ctx : nxd::PassSlot {
    int device = 0;
    hipStream_t stream2 = nullptr;  // second branch of the per-bounce fork (shadow trace)
    int numCUs = 0;
    // passes in flight: slot 0 is the context itself, extra[k] is slot k + 1
    std::vector<std::unique_ptr<nxd::PassSlot>> extra;
    uint32_t passesInFlight = 1;
    uint32_t nextSlot = 0;                 // slot the next rendered pass goes to
    std::vector<nxd::PassSlot*> pending;   // rendered, not yet accumulated, oldest first
    nxd::PassSlot* lastRendered = nullptr;

    uint32_t width = 0, height = 0, localCount = 0;
    uint32_t framesPerPass = 1, pathCount = 0;
    // Nominal queue capacity in paths (grow-only across nxhip_set_frames_per_pass).  A slot allocates its ~0.28 KB per path when
    // it is about to be used (ensure_slot_queues) and gives it back when the context stops using it: the slots beyond the
    // passes in flight, and slot 0 while the passes render in the extra slots.
    size_t queueCapacity = 0;
    size_t radianceBoundCapacity = 0;  // float4 capacity of an externally bound radiance buffer, 0 = own buffer

    nxd::DeviceState h{};  // host mirror (scene + slot 0's queues), uploaded to every slot's dState when dirty
    bool stateDirty = true;

    // scene
    std::vector<nxd::BlasHost> blas;
    nxd::DevBuf blasTable;
    nxd::DevBuf tlasTightBoxes;  // [instanceCount] x 24 B: the instance boxes a device-built TLAS was built from (nx_instbox.h); empty after nxhip_set_tlas
    nxd::DevBuf tlasNodes, tlasInstIdx, instTrav, instances;
    std::vector<nx_bvh_instance> hostInstances;
    std::vector<nx_material> hostMaterials;  // kept for the cross-table index check before a render
    std::vector<nx_material> hostMaterialsDev;  // the device copy (derived flag byte included), for the instances' shading records
    nxd::DevBuf shadeInst;        // [instanceCount] ShadeInst (nx_device.h): rebuilt before a render when an instance, BLAS or material table changed
    bool shadeInstDirty = true;
    std::vector<nx_light> hostLights;
    std::vector<uint32_t> hostInstIdx;  // TLAS leaf order
    nxd::DevBuf materials, lights;
    // device-side dynamic transforms (nx_refit.hip): TLAS nodes grouped by depth (deepest first), leaf slot of every instance,
    // per-node box scratch, staging for the ids / matrices of one call
    nxd::DevBuf refitOrder, refitLevelStart, leafOfInstance, refitBoxes, refitIds, refitMatrices;
    uint32_t refitLevels = 0, tlasNodeCount = 0;
    std::vector<nxd::TextureHost> diffuseMaps, emissiveMaps;
    nxd::TextureHost hdrMap;
    nxd::DevBuf diffuseTable, emissiveTable, srgbLut;
    // environment importance sampling (extension): host copy of the hdr map and the tables built from it
    std::vector<uint8_t> hostHdr;
    bool envSampling = false;
    nxd::DevBuf envMarginalCdf, envRowCdf, envDensity, envMarginalGuide, envRowGuide;
    // paths / queues
    nxd::DevBuf pixelMap, accumulation, rgba8;
    nxd::DevBuf traceStats;
    void* hostStaging = nullptr;   // pinned host buffer for large uploads (nxhip_build_blas_batch), grown on demand, freed with the context
    size_t hostStagingBytes = 0;
    hipEvent_t stagingDone[2] = {nullptr, nullptr};  // one per half of the staging buffer: its last transfer has completed

    int deviceBuilderRadius = -1;  // nxhip_build_blas / nxhip_rebuild_tlas: -1 = top-down binned SAH (NXHIP_BUILDER_SAH), 0 = radix tree (LBVH), > 0 = neighbour search radius of the clustering builder
    uint32_t frameNumber = 0;  // host mirror of FrameState.frameNumber
    bool statsEnabled = false;
    bool timingEnabled = false;
    int timingMode = 0;  // 0 off, 1 eager launches with an event pair each, 2 / 3 event-record nodes inside the frame graph
    std::vector<nxd::KernelTimer> graphTimers;
    std::vector<int> graphTimerClass;
    bool graphTimersPending = false;  // mode 3: the graph's events hold an unread replay
    nxhip_kernel_times times{};
    std::vector<nxd::KernelTimer> timerPool;
    std::vector<int> timerClass;  // kernel class of timerPool[i]

    // multi-GPU tile split (nxhip_multigpu.hip): RCCL communicator (opaque), root-side gather / full-image buffers
    void* mgpuComm = nullptr;
    bool mgpuOwnsComm = false;
    int mgpuWorld = 1, mgpuRank = 0;
    uint32_t mgpuTileRows = 0;
    uint64_t pixelSetGeneration = 0;  // bumped whenever the context's pixel set changes (alloc_paths: resize, pixel map)
    uint64_t mgpuPixelSet = 0;        // the generation the tile split was set up for (nxhip_mgpu_gather refuses another)
    nxd::DevBuf mgpuGathered, mgpuMaps, mgpuFullAccum, mgpuFullRgba8;

    int traceBlocks = 0, shadowBlocks = 0, wideBlocks = 0;  // full-chip persistent grids (see trace_blocks)
    int tailBlocks = 0;
    int tailBounce = -1;  // first bounce of the tail kernel, 0 = off, -1 = automatic (see tail_bounce in nxhip_api.hip)
    bool traceGridForced = false;                            // NX_TRACE_BLOCKS_*: use them as they are
    int shadeBlocksPerCU = 10, logicBlocksPerCU = 2;  // grid-stride kernels: workgroups per CU
    bool serialShade = false;  // the four material kernels of a bounce as one graph branch instead of four
    int parallelShade = -1;    // NX_SHADE_PARALLEL (tuning experiments only): 1 = parallel branches whatever the pass size
    // material types the scene's materials use (bit NX_MAT_*): a type no material has can never receive a queue item, so its
    // kernel is left out of the pass graph (a launch on the critical path of every bounce, however empty)
    uint32_t materialTypeMask = 0xfu;
    // NX_TUNING_KNOBS=1 NX_PIPELINE_CLASSIC=1 (measurement only): the logic kernel + material queues also under fast compaction
    bool classicPipeline = false;
    bool thinInFlight = false;  // NX_THIN_IN_FLIGHT (NX_TUNING_KNOBS): the thin level also when several passes are in flight
    bool thinInHooks = false;  // nxhip_debug_set_thin: the ray-batch hooks hand over and launch the thin kernel too
    bool thinJoint = false;  // NX_THIN_JOINT=1 (measurement only): one thin launch per level instead of one per trace launch
    bool thinWaves = true;  // the trace launches of a pass finish the last long rays of a dry wave cooperatively (NX_NO_THIN=1 with NX_TUNING_KNOBS=1: off)
    bool dead = false;          // nxhip_sync_timeout gave up: no further device work is issued or waited for
    int pixelOrder = 0;         // nxhip_set_pixel_order: NXHIP_ORDER_* of the full frame, re-applied by nxhip_resize (a caller's own map is not)
    bool entryPoints = false;   // nxhip_set_entry_points (the tables: PassSlot::entryTable, one per slot)
    bool scanSeparate = false;  // NX_SCAN_SEPARATE=1 (measurement only): one material launch per type instead of one for all
};
