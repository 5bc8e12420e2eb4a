// nxhip_api.hip — implementation of the C-ABI device layer declared in include/nexus_hip.h.
// Each entry point cites, in the header, the reference interface it replaces; this file is the HIP runtime
// plumbing behind it: allocation, uploads, the per-frame hipGraph and the kernel-level test hooks.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <thread>
#include <string>
#include <vector>

#include "nx_context.h"

namespace nxd {

const void* trace_kernel_ptr(bool anyHit, bool stats);
const void* tail_kernel_ptr();
const void* logic_kernel_ptr(bool ordered, int items);
const void* shade_kernel_ptr(int type, bool ordered);
const void* shade_scan_kernel_ptr();
const void* miss_scan_kernel_ptr();
const void* count_scan_kernel_ptr();
const void* inst_code_kernel_ptr();
const void* entry_state_kernel_ptr();
const void* thin_kernel_ptr();
const void* begin_frame_kernel_ptr();
const void* hook_sizes_kernel_ptr();
const void* generate_kernel_ptr();
const void* accumulate_kernel_ptr();
const void* compose_kernel_ptr();
uint64_t layout_stamp_trace();
uint64_t layout_stamp_wavefront();
uint64_t layout_stamp_refit();
uint64_t layout_stamp_lbvh();
uint64_t layout_stamp_multigpu();
uint64_t layout_stamp_entry();
const void* bsdf_hook_kernel_ptr();
const void* fmath_hook_kernel_ptr();
const void* tex2d_hook_kernel_ptr();
const void* instance_transform_kernel_ptr();
const void* tlas_refit_kernel_ptr();
int lbvh_build(nxhip_ctx* c, const nx_triangle* dTris, uint32_t n, int plocRadius, DevBuf& nodes, DevBuf& primIdx, DevBuf& isect, uint32_t* nodeCount);
int lbvh_build_batch(nxhip_ctx* c, const nx_triangle* dTris, const std::vector<uint32_t>& counts, DevBuf& nodes, DevBuf& primIdx, DevBuf& isect, std::vector<uint32_t>& nodeFirst,
                     std::vector<uint32_t>& nodeCounts);
int lbvh_build_tlas(nxhip_ctx* c, const nx_bvh_instance* dInstances, uint32_t n, int plocRadius, DevBuf& nodes, DevBuf& primIdx, DevBuf& box, bool* boxesAreTight, uint32_t* nodeCount);

static thread_local std::string g_lastError;

void set_error(const std::string& msg) { g_lastError = msg; }

bool hip_ok(hipError_t e, const char* what, const char* file, int line)
{
    if (e == hipSuccess) return true;
    // the failure is reported through the status + message of this call; it must not stay behind as HIP's "last error"
    // (library code that polls hipGetLastError after its launches — rocPRIM does — would report it as its own)
    (void)hipGetLastError();
    char buf[512];
    std::snprintf(buf, sizeof buf, "HIP error %d (%s) in '%s' at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    set_error(buf);
    return false;
}

bool DevBuf::alloc(size_t n)
{
    release();
    if (n == 0) n = 16;
    void* q = nullptr;
    if (!hip_ok(hipMalloc(&q, n), "hipMalloc", __FILE__, __LINE__)) return false;
    p = q;
    bytes = n;
    return true;
}

#ifndef NX_TRACE_BLOCK
#define NX_TRACE_BLOCK 256
#endif
constexpr int kTraceBlockThreads = NX_TRACE_BLOCK;
constexpr int kWideBlockThreads = 256;
#ifndef NX_SHADE_BLOCK
#define NX_SHADE_BLOCK 256
#endif
#ifndef NX_LOGIC_BLOCK
#define NX_LOGIC_BLOCK 1024
#endif
constexpr int kLogicBlockThreads = NX_LOGIC_BLOCK;
constexpr int kShadeBlockThreads = NX_SHADE_BLOCK;
constexpr int kHookBounceSlot = NX_PATH_MAX_LENGTH - 1;  // queue-size slot used by the batch test hooks

}  // namespace nxd

using namespace nxd;

#define NX_CHECK_CTX(ctx)                      \
    do {                                       \
        if (!(ctx)) {                          \
            set_error("null context");         \
            return NXHIP_ERR_INVALID;          \
        }                                      \
        if ((ctx)->dead) {                     \
            set_error("the context is dead: nxhip_sync_timeout gave up waiting for the device (see include/nexus_hip.h)"); \
            return NXHIP_ERR_TIMEOUT;          \
        }                                      \
    } while (0)

#define NX_ALLOC(buf, n)                                  \
    do {                                                  \
        if (!(buf).alloc(n)) return NXHIP_ERR_HIP;        \
    } while (0)

static int read_float4_as_float3(nxhip_ctx* c, const void* dev, uint32_t count, float* dst);

static int fail_invalid(const char* msg)
{
    set_error(msg);
    return NXHIP_ERR_INVALID;
}
static int fail_invalid(const std::string& msg) { return fail_invalid(msg.c_str()); }

// Every translation unit of this library must have been compiled with the same device-side layouts (nx_device.h layout_stamp):
// the host code here fills DeviceState / Counters / InstTrav blocks that the kernels of the other units read.
static int check_layouts()
{
    const struct { const char* unit; uint64_t stamp; } units[] = {
        {"nx_trace.hip", layout_stamp_trace()}, {"nx_wavefront.hip", layout_stamp_wavefront()}, {"nx_refit.hip", layout_stamp_refit()},
        {"nx_lbvh.hip", layout_stamp_lbvh()}, {"nxhip_multigpu.hip", layout_stamp_multigpu()}, {"nx_entry.hip", layout_stamp_entry()},
    };
    for (const auto& u : units) {
        if (u.stamp != layout_stamp()) {
            char buf[320];
            std::snprintf(buf, sizeof buf, "libnexus_amd.so was linked from objects of different source states: %s was compiled with device layout stamp %016llx, "
                          "nxhip_api.hip with %016llx (DeviceState / Counters / record strides differ) - rebuild the library from one tree (make clean && make)",
                          u.unit, (unsigned long long)u.stamp, (unsigned long long)layout_stamp());
            set_error(buf);
            return NXHIP_ERR_ABI;
        }
    }
    return NXHIP_OK;
}

uint64_t nxhip_abi_stamp(void) { return nxhip_header_abi_stamp(); }

int nxhip_check_library(uint64_t callerStamp)
{
    if (callerStamp != nxhip_abi_stamp()) {
        char buf[320];
        std::snprintf(buf, sizeof buf, "ABI mismatch: the caller was built against a nexus_hip.h / nexus_pod.h with stamp %016llx, this library (API version %d) has %016llx - "
                      "a stale or foreign libnexus_amd.so (NEXUS_AMD_LIB, LD_LIBRARY_PATH), or bindings of another version",
                      (unsigned long long)callerStamp, NXHIP_API_VERSION, (unsigned long long)nxhip_abi_stamp());
        set_error(buf);
        return NXHIP_ERR_ABI;
    }
    return check_layouts();
}

// slot k of the context: 0 is the context itself, k >= 1 the extra in-flight passes
static PassSlot* slot_at(nxhip_ctx* c, uint32_t k) { return k == 0 ? static_cast<PassSlot*>(c) : c->extra[k - 1].get(); }
static uint32_t slot_count(const nxhip_ctx* c) { return 1u + (uint32_t)c->extra.size(); }

// Wait for everything the context has issued, on every slot's stream (scene edits, re-allocations, read-backs).
static int sync_all(nxhip_ctx* c)
{
    for (uint32_t k = 0; k < slot_count(c); k++) {
        PassSlot* s = slot_at(c, k);
        if (s->stream) NX_HIP(hipStreamSynchronize(s->stream));
    }
    return NXHIP_OK;
}
#define NX_SYNC_ALL(c)                                  \
    do {                                                \
        const int rcSync_ = sync_all(c);                \
        if (rcSync_ != NXHIP_OK) return rcSync_;        \
    } while (0)

// The nxhip_debug_* entry points exist for the tests (one of them plants a cycle in an uploaded BVH).  A `make release` library
// (NX_NO_DEBUG_HOOKS) keeps the symbols — the header and the ABI stamp are the same — and refuses the calls.
#ifdef NX_NO_DEBUG_HOOKS
#define NX_DEBUG_HOOK(name) return fail_invalid(name ": this library was built without the test hooks (make release)")
#else
#define NX_DEBUG_HOOK(name) do { } while (0)
#endif

static void invalidate_graph(nxhip_ctx* c)
{
    // a replay may still be executing (render calls are asynchronous): destroying its exec, graph and timing events
    // under it is not allowed.  Not a hot path: settings / mode / timing changes only.
    for (uint32_t k = 0; k < slot_count(c); k++) {
        PassSlot* s = slot_at(c, k);
        if (!s->graphs.empty() && s->stream) (void)hipStreamSynchronize(s->stream);
        for (auto& g : s->graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            if (g.graph) (void)hipGraphDestroy(g.graph);
        }
        s->graphs.clear();
    }
    for (auto& t : c->graphTimers) {
        if (t.start) (void)hipEventDestroy(t.start);
        if (t.stop) (void)hipEventDestroy(t.stop);
    }
    c->graphTimers.clear();
    c->graphTimerClass.clear();
}

// Slots per queue region for a capacity of n paths (nx_device.h, Counters): an even share, 64-aligned, plus the slack a region
// may run over it; a queue buffer holds kQueueShards regions.
static size_t queue_region_cap(size_t n) { return ((n + kQueueShards * 64 - 1) / (kQueueShards * 64)) * 64 + kQueueShardSlack; }
static size_t queue_buffer_slots(size_t n) { return queue_region_cap(n) * kQueueShards; }
// entries per list of rays handed to the thin kernel (nx_trace.hip): a launch hands over at most DeviceState::thinLanes (4) rays per wave, 20 480 for a
// full grid; a list that runs over only makes the waves it has no room for finish their rays themselves
constexpr uint32_t kThinListEntries = 1u << 15;  // (x 2 lists x (4 B + a 320-byte ThinState) = 21 MB per slot)
static size_t scan_status_tiles(size_t n) { return n / (size_t)std::min(kLogicBlockThreads, kShadeBlockThreads) + 2; }

// Which pipeline a pass runs (nx_wavefront.hip): SCAN — the logic step's decision rides in the hit records and the material kernels
// pick their items out of the trace queue — whenever slots are handed out by racing atomics; the CLASSIC logic kernel + material
// queues for the ordered compaction, whose serial slot order IS the reference's copy order (PathTracer.cu:183-206).
static bool scan_pipeline(const nxhip_ctx* c) { return c->h.compactMode == NX_COMPACT_FAST && !c->classicPipeline; }

static TraceQueue trace_queue_of(const PassSlot* s)
{
    TraceQueue t{};
    t.rays[0] = TraceRays{s->trRayO.as<float4>(), s->trRayD.as<float4>(), s->trTp.as<float4>()};
    t.rays[1] = TraceRays{s->trRayO2.as<float4>(), s->trRayD2.as<float4>(), s->trTp2.as<float4>()};
    t.hit = s->trHit.as<float4>();
    t.hitInst = s->trHitInst.as<uint32_t>();
    return t;
}

// The device-state block of a slot: the scene part of the host mirror plus the slot's own queues, counters and frame words.
static void compose_view(nxhip_ctx* c, PassSlot* s)
{
    DeviceState& v = s->view;
    v = c->h;
    if (s != static_cast<PassSlot*>(c)) {
        v.radiance = s->radiance.as<float4>();
        v.rayOrigin = s->rayOrigin.as<float4>();
        v.trace = trace_queue_of(s);
        v.shadow = ShadowQueue{s->shRayO.as<float4>(), s->shRayD.as<float4>(), s->shRadiance.as<float4>()};
        for (int m = 0; m < 4; m++) v.material[m] = MaterialQueue{s->mqHit[m].as<float4>(), s->mqDirInst[m].as<float4>(), s->mqTp[m].as<float4>()};
    }
    v.counters = s->counters.as<Counters>();
    v.frame = s->frame.as<FrameState>();
    v.scanStatus = s->scanStatus.as<unsigned long long>();
    v.thinClosest = s->thinLists.as<uint32_t>();
    v.thinAny = s->thinLists.p ? s->thinLists.as<uint32_t>() + kThinListEntries : nullptr;
    v.thinCapacity = (s->thinLists.p && s->thinStates.p) ? kThinListEntries : 0u;
    v.thinStates = s->thinStates.as<ThinState>();
    v.entry = (c->entryPoints && s->entryTable.p) ? s->entryTable.as<EntryState>() : nullptr;
    v.entryRuns = v.entry ? s->entryRuns : 0u;
    // queue regions: eight, or one spanning the buffer when slots are handed out in the reference's serial order
    const bool ordered = c->h.compactMode == NX_COMPACT_ORDERED;
    v.queueShards = ordered ? 1u : (uint32_t)kQueueShards;
    v.queueShardCap = (uint32_t)(ordered ? queue_buffer_slots(c->queueCapacity) : queue_region_cap(c->queueCapacity));
}

static int upload_state(nxhip_ctx* c)
{
    if (!c->stateDirty) return NXHIP_OK;
    NX_SYNC_ALL(c);  // no pass in flight may see half of an edit
    for (uint32_t k = 0; k < slot_count(c); k++) {
        PassSlot* s = slot_at(c, k);
        compose_view(c, s);
        NX_HIP(hipMemcpy(s->dState.p, &s->view, sizeof(DeviceState), hipMemcpyHostToDevice));
    }
    c->stateDirty = false;
    return NXHIP_OK;
}

// Queue / path-state buffers for n paths (contents undefined), and the device-state pointers to them.  All or nothing:
// the new set is allocated beside the old one and swapped in only when every allocation has succeeded, so a failed
// growth (out of device memory half-way through 25 buffers) leaves the context exactly as it was, still able to render
// at its previous capacity.
static int alloc_slot_queues(nxhip_ctx* c, PassSlot* q, size_t n)
{
    DevBuf* const slots[] = {&q->radiance, &q->rayOrigin, &q->trRayO, &q->trRayD, &q->trHit, &q->trHitInst, &q->trTp, &q->shRayO, &q->shRayD, &q->shRadiance,
                             &q->mqHit[0], &q->mqDirInst[0], &q->mqTp[0], &q->mqHit[1], &q->mqDirInst[1], &q->mqTp[1],
                             &q->mqHit[2], &q->mqDirInst[2], &q->mqTp[2], &q->mqHit[3], &q->mqDirInst[3], &q->mqTp[3],
                             &q->trRayO2, &q->trRayD2, &q->trTp2};
    const size_t elem[] = {16, 16, 16, 16, 16, 4, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
    constexpr int kCount = (int)(sizeof(slots) / sizeof(slots[0]));
    constexpr int kFirstMaterial = 10, kFirstSecondSet = 22;
    static_assert(sizeof(elem) / sizeof(elem[0]) == (size_t)kCount, "one element size per buffer");
    // the pipeline decides which buffers exist: SCAN has no material queues (192 B per path), CLASSIC no second set of rays (48 B)
    const bool scan = scan_pipeline(c);
    DevBuf fresh[kCount];
    for (int i = 0; i < kCount; i++) {  // the first two are per path, the rest are queues (regions + slack)
        const bool unused = scan ? (i >= kFirstMaterial && i < kFirstSecondSet) : i >= kFirstSecondSet;
        if (!fresh[i].alloc(unused ? 0 : (i < 2 ? n : queue_buffer_slots(n)) * elem[i])) return NXHIP_ERR_HIP;  // `fresh` frees what it got; the context is untouched
    }
    // ordered compaction: status words of the tiles of the largest possible launch (a queue never holds more than n items;
    // tiles of the smaller of the two workgroup sizes), zeroed once — tag 0 is never a launch's serial
    DevBuf freshStatus;
    const size_t statusBytes = scan_status_tiles(n) * kScanWords * sizeof(unsigned long long);
    if (!freshStatus.alloc(statusBytes)) return NXHIP_ERR_HIP;
    NX_HIP(hipMemset(freshStatus.p, 0, statusBytes));
    DevBuf freshThin;  // (contents: whatever the trace launches of a level write before the thin kernel of that level reads)
    if (!freshThin.alloc((size_t)2 * kThinListEntries * sizeof(uint32_t))) return NXHIP_ERR_HIP;
    DevBuf freshThinStates;  // (entry k of a list and state k belong together: the wave that writes one writes the other)
    if (!freshThinStates.alloc((size_t)2 * kThinListEntries * sizeof(ThinState))) return NXHIP_ERR_HIP;
    NX_HIP(hipMemset(fresh[0].p, 0, n * 16));  // radiance
    NX_HIP(hipMemset(fresh[1].p, 0, n * 16));  // the paths' previous vertices: a read of an entry nobody has written yet is at least deterministic
    NX_SYNC_ALL(c);                            // nothing in flight may still use the old buffers
    for (int i = 0; i < kCount; i++) *slots[i] = std::move(fresh[i]);
    q->scanStatus = std::move(freshStatus);
    q->thinLists = std::move(freshThin);
    q->thinStates = std::move(freshThinStates);
    q->scanEpoch = 0;
    q->pathCapacity = n;
    q->queuesScan = scan;
    if (q == static_cast<PassSlot*>(c)) {
        c->radianceBoundCapacity = 0;
        DeviceState& h = c->h;
        h.radiance = c->radiance.as<float4>();
        h.rayOrigin = c->rayOrigin.as<float4>();
        h.trace = trace_queue_of(c);
        h.shadow = ShadowQueue{c->shRayO.as<float4>(), c->shRayD.as<float4>(), c->shRadiance.as<float4>()};
        for (int m = 0; m < 4; m++) h.material[m] = MaterialQueue{c->mqHit[m].as<float4>(), c->mqDirInst[m].as<float4>(), c->mqTp[m].as<float4>()};
    }
    c->stateDirty = true;
    return NXHIP_OK;
}

// Give a slot's queue buffers back (nothing of it may be in flight: the caller has synchronised).
static void release_slot_queues(nxhip_ctx* c, PassSlot* q)
{
    DevBuf* const bufs[] = {&q->radiance, &q->rayOrigin, &q->trRayO, &q->trRayD, &q->trHit, &q->trHitInst, &q->trTp, &q->shRayO, &q->shRayD, &q->shRadiance,
                            &q->mqHit[0], &q->mqDirInst[0], &q->mqTp[0], &q->mqHit[1], &q->mqDirInst[1], &q->mqTp[1], &q->mqHit[2], &q->mqDirInst[2], &q->mqTp[2],
                            &q->mqHit[3], &q->mqDirInst[3], &q->mqTp[3], &q->trRayO2, &q->trRayD2, &q->trTp2};
    for (DevBuf* b : bufs) b->release();
    q->scanStatus.release();
    q->thinLists.release();
    q->thinStates.release();
    q->pathCapacity = 0;
    if (q == static_cast<PassSlot*>(c)) {
        DeviceState& h = c->h;
        h.radiance = h.rayOrigin = nullptr;
        h.trace = TraceQueue{};
        h.shadow = ShadowQueue{};
        for (int m = 0; m < 4; m++) h.material[m] = MaterialQueue{};
    }
    c->stateDirty = true;
}

// A new nominal capacity: every slot that holds buffers is re-allocated (slot 0 last, so that a failure in an extra slot leaves
// slot 0 untouched); released slots allocate when they are next used.
static int alloc_queues(nxhip_ctx* c, size_t n)
{
    for (uint32_t k = slot_count(c); k-- > 0;) {
        PassSlot* q = slot_at(c, k);
        if (k != 0 && q->pathCapacity == 0) continue;
        const int rc = alloc_slot_queues(c, q, n);
        if (rc != NXHIP_OK) return rc;
    }
    c->queueCapacity = n;
    return NXHIP_OK;
}

// Before a slot is used: its queues exist at the nominal capacity.
static bool slot_queues_ready(const nxhip_ctx* c, const PassSlot* q) { return q->pathCapacity >= c->queueCapacity && q->pathCapacity > 0 && q->queuesScan == scan_pipeline(c); }

static int ensure_slot_queues(nxhip_ctx* c, PassSlot* q)
{
    if (slot_queues_ready(c, q)) return NXHIP_OK;
    float4* const boundPtr = c->h.radiance;
    const size_t boundCap = c->radianceBoundCapacity;
    const int rc = alloc_slot_queues(c, q, std::max<size_t>(c->queueCapacity, 1));
    if (rc == NXHIP_OK && boundCap != 0 && q == static_cast<PassSlot*>(c)) {  // an external radiance binding survives
        c->h.radiance = boundPtr;
        c->radianceBoundCapacity = boundCap;
    }
    return rc;
}

// Everything sized by the pixel set of this context: queues for localCount * framesPerPass paths and a zeroed image.
static int alloc_paths(nxhip_ctx* c, uint32_t localCount)
{
    const size_t n = (size_t)std::max<uint32_t>(localCount, 1u) * c->framesPerPass;
    const size_t full = std::max<size_t>((size_t)c->width * c->height, localCount);
    if (n > 0x7fffffffull) return fail_invalid("more than 2^31 paths (pixels x frames per pass): lower nxhip_set_frames_per_pass first");
    DevBuf freshAccum, freshRgba;  // all or nothing, as alloc_queues
    if (!freshAccum.alloc(full * 16) || !freshRgba.alloc(full * 4)) return NXHIP_ERR_HIP;
    const int rc = alloc_queues(c, n);
    if (rc != NXHIP_OK) return rc;
    c->accumulation = std::move(freshAccum);
    c->rgba8 = std::move(freshRgba);
    NX_HIP(hipMemsetAsync(c->accumulation.p, 0, full * 16, c->stream));
    NX_HIP(hipMemsetAsync(c->rgba8.p, 0, full * 4, c->stream));
    c->localCount = localCount;
    c->pixelSetGeneration++;
    c->pathCount = localCount * c->framesPerPass;
    DeviceState& h = c->h;
    h.localCount = localCount;
    h.framesPerPass = c->framesPerPass;
    h.pathCount = c->pathCount;
    h.accumulation = c->accumulation.as<float4>();
    h.rgba8 = c->rgba8.as<uint32_t>();
    c->stateDirty = true;
    invalidate_graph(c);
    return NXHIP_OK;
}

// The frame counter lives on the host; every pass's begin_frame_kernel carries the number of its last frame.  Changing it
// must not overtake passes already issued with the old numbering.
static int set_frame_number_device(nxhip_ctx* c, uint32_t f)
{
    NX_SYNC_ALL(c);
    c->frameNumber = f;
    return NXHIP_OK;
}

extern "C" {

const char* nxhip_last_error(void) { return g_lastError.c_str(); }

int nxhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// bit 0: device code for gfx950; bit 1: built with the Makefile's scheduler flags (SCHEDFLAGS: about 10 % on the material kernels);
// bit 2: the nxhip_debug_* test hooks are compiled in (the default build; `make release` leaves them as stubs that refuse)
int nxhip_build_info(void)
{
    int f = 0;
#ifdef NX_BUILT_FOR_GFX950
    f |= 1;
#endif
#ifdef NX_SCHED_FLAGS
    f |= 2;
#endif
#ifndef NX_NO_DEBUG_HOOKS
    f |= 4;
#endif
    return f;
}

int nxhip_has_gfx950_code(void)
{
#ifdef NX_BUILT_FOR_GFX950
    return 1;
#else
    return 0;
#endif
}

int nxhip_create(int device, uint32_t width, uint32_t height, void* stream, nxhip_ctx** out)
{
    if (!out) return fail_invalid("nxhip_create: out is null");
    *out = nullptr;
    if (width == 0 || height == 0) return fail_invalid("nxhip_create: zero-sized viewport");
    if ((uint64_t)width * height > 0x7fffffffull) return fail_invalid("nxhip_create: more than 2^31 pixels");
    if (const int rcLayouts = check_layouts()) return rcLayouts;  // a library of mixed objects is refused before anything is launched
    if (nxhip_device_count() <= device || device < 0) {
        set_error("nxhip_create: no such HIP device");
        return NXHIP_ERR_NO_DEVICE;
    }
    NX_HIP(hipSetDevice(device));
    nxhip_ctx* c = new (std::nothrow) nxhip_ctx();
    if (!c) return fail_invalid("nxhip_create: out of host memory");
    c->device = device;
    c->width = width;
    c->height = height;
    int rc = NXHIP_OK;
    do {
        if (stream) c->stream = (hipStream_t)stream;
        else {
            if (!hip_ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate", __FILE__, __LINE__)) { rc = NXHIP_ERR_HIP; break; }
            c->ownsStream = true;
        }
        hipDeviceProp_t prop;
        if (!hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties", __FILE__, __LINE__)) { rc = NXHIP_ERR_HIP; break; }
        c->numCUs = prop.multiProcessorCount;

        if (!c->dState.alloc(sizeof(DeviceState)) || !c->counters.alloc(sizeof(Counters)) || !c->frame.alloc(sizeof(FrameState)) ||
            !c->traceStats.alloc(2 * sizeof(TraceStatsDev)) || !c->srgbLut.alloc(256 * sizeof(float))) { rc = NXHIP_ERR_HIP; break; }
        (void)hipMemsetAsync(c->counters.p, 0, sizeof(Counters), c->stream);
        (void)hipMemsetAsync(c->traceStats.p, 0, 2 * sizeof(TraceStatsDev), c->stream);
        FrameState fs{0u, -1, -1, 0u, 0u, {0u, 0u, 0u}};
        (void)hipMemcpyAsync(c->frame.p, &fs, sizeof fs, hipMemcpyHostToDevice, c->stream);
        float lut[256];
        for (int i = 0; i < 256; i++) {
            const float x = (float)i / 255.0f;
            lut[i] = x <= 0.04045f ? x / 12.92f : std::pow((x + 0.055f) / 1.055f, 2.4f);
        }
        (void)hipMemcpyAsync(c->srgbLut.p, lut, sizeof lut, hipMemcpyHostToDevice, c->stream);
        if (!hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize", __FILE__, __LINE__)) { rc = NXHIP_ERR_HIP; break; }

        DeviceState& h = c->h;
        std::memset(&h, 0, sizeof h);
        h.counters = c->counters.as<Counters>();
        h.frame = c->frame.as<FrameState>();
        h.traceStats = c->traceStats.as<TraceStatsDev>();
        h.srgbLut = c->srgbLut.as<float>();
        h.settings.useMIS = 1;  // Renderer/RenderSettings.h:4-10 defaults
        h.settings.pathLength = 10;
        h.settings.backgroundColor[0] = h.settings.backgroundColor[1] = h.settings.backgroundColor[2] = 1.0f;
        h.settings.backgroundIntensity = 0.0f;
        h.camera.resolution[0] = width;
        h.camera.resolution[1] = height;
        // (round 6: 16 / 16 — at most 16 busy lanes for 16 iterations, and still "four average rays" long: nx_trace.hip kThinFactor.  With the
        //  traversal state carried over: driver command +0.8 % over 4 / 64, a rank of 8's share 3.80 -> 3.70 ms; profiles/r06_handover.txt)
        h.thinLanes = 16u;
        h.thinIters = 16u;
        h.rngMode = NX_RNG_REFERENCE_SLOT;
        h.compactMode = NX_COMPACT_FAST;
        h.conductorMode = NX_CONDUCTOR_REFERENCE;
        rc = alloc_paths(c, width * height);
        if (rc != NXHIP_OK) break;

        // persistent launch geometry from the occupancy the kernels actually get
        int perCU = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, trace_kernel_ptr(false, false), kTraceBlockThreads, 0) != hipSuccess || perCU < 1) perCU = 4;
        c->traceBlocks = std::max(1, perCU) * c->numCUs;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, trace_kernel_ptr(true, false), kTraceBlockThreads, 0) != hipSuccess || perCU < 1) perCU = 4;
        c->shadowBlocks = std::max(1, perCU) * c->numCUs;
        c->wideBlocks = 8 * c->numCUs;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, tail_kernel_ptr(), kTraceBlockThreads, 0) != hipSuccess || perCU < 1) perCU = 2;
        c->tailBlocks = std::max(1, perCU) * c->numCUs;
        // Launch-geometry knobs of the measurement sweeps.  They are read only when NX_TUNING_KNOBS=1 says
        // that a sweep is running: a stray variable in a user's environment does not reconfigure the product.
        if (const char* on = std::getenv("NX_TUNING_KNOBS"); on && std::atoi(on) == 1) {
            if (const char* e = std::getenv("NX_TAIL_BOUNCE")) {  // tuning experiments only: 0 = off
                const int n = std::atoi(e);
                if (n == 0 || n == -1 || (n >= 2 && n <= NX_PATH_MAX_LENGTH)) c->tailBounce = n;
            }
            if (const char* e = std::getenv("NX_TRACE_BLOCKS_PER_CU")) {  // tuning experiments only
                const int n = std::atoi(e);
                if (n >= 1 && n <= 16) { c->traceBlocks = c->shadowBlocks = n * c->numCUs; c->traceGridForced = true; }
            }
            if (const char* e = std::getenv("NX_SHADOW_BLOCKS_PER_CU")) {  // tuning experiments only: the any-hit launches' grid alone
                const int n = std::atoi(e);
                if (n >= 1 && n <= 16) { c->shadowBlocks = n * c->numCUs; c->traceGridForced = true; }
            }
            if (const char* e = std::getenv("NX_CLOSEST_BLOCKS_PER_CU")) {  // tuning experiments only: the closest-hit launches' grid alone
                const int n = std::atoi(e);
                if (n >= 1 && n <= 16) { c->traceBlocks = n * c->numCUs; c->traceGridForced = true; }
            }
            if (const char* e = std::getenv("NX_SHADE_BLOCKS_PER_CU")) {  // tuning experiments only
                const int n = std::atoi(e);
                if (n >= 1 && n <= 64) c->shadeBlocksPerCU = n;
            }
            if (const char* e = std::getenv("NX_LOGIC_BLOCKS_PER_CU")) {  // tuning experiments only
                const int n = std::atoi(e);
                if (n >= 1 && n <= 64) c->logicBlocksPerCU = n;
            }
            if (const char* e = std::getenv("NX_THIN_JOINT")) c->thinJoint = std::atoi(e) != 0;  // measurement only
            if (const char* e = std::getenv("NX_THIN_LANES")) { const int n = std::atoi(e); if (n >= 1 && n <= 64) c->h.thinLanes = (uint32_t)n; }   // sweeps of the hand-over rule
            if (const char* e = std::getenv("NX_THIN_ITERS")) { const int n = std::atoi(e); if (n >= 1 && n <= 4096) c->h.thinIters = (uint32_t)n; }
            if (const char* e = std::getenv("NX_THIN_IN_FLIGHT")) c->thinInFlight = std::atoi(e) != 0;  // measurement only: the thin level with passes in flight too
            if (const char* e = std::getenv("NX_NO_THIN")) c->thinWaves = std::atoi(e) == 0;  // measurement only: no cooperative finish of a dry wave's last rays
            if (const char* e = std::getenv("NX_SCAN_SEPARATE")) c->scanSeparate = std::atoi(e) != 0;  // measurement only: one material launch per type in the SCAN pipeline
            if (const char* e = std::getenv("NX_PIPELINE_CLASSIC")) c->classicPipeline = std::atoi(e) != 0;  // measurement only: logic kernel + material queues under fast compaction too
            if (const char* e = std::getenv("NX_SHADE_SERIAL")) c->serialShade = std::atoi(e) != 0;  // tuning experiments only
            if (const char* e = std::getenv("NX_SHADE_PARALLEL")) c->parallelShade = std::atoi(e);    // tuning experiments only
            if (const char* e = std::getenv("NX_TRACE_BLOCKS_TOTAL")) {  // tuning experiments only
                const int n = std::atoi(e);
                if (n >= 1 && n <= 65536) { c->traceBlocks = c->shadowBlocks = n; c->traceGridForced = true; }
            }
        }
    } while (false);
    if (rc != NXHIP_OK) {
        nxhip_destroy(c);
        return rc;
    }
    *out = c;
    return NXHIP_OK;
}

void nxhip_destroy(nxhip_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->dead) {
        // nxhip_sync_timeout gave up on this context.  If the device has finished after all, everything below is safe; if it still has
        // not, waiting for it (stream synchronisation, hipFree) would hang the caller: the context's device memory stays where it is
        // until the process ends — which is what the caller of a timed-out context is about to do.
        bool busy = false;
        for (uint32_t k = 0; k < slot_count(c); k++)
            if (slot_at(c, k)->stream && hipStreamQuery(slot_at(c, k)->stream) != hipSuccess) busy = true;
        if (busy) return;
        c->dead = false;
    }
    (void)sync_all(c);
    (void)nxhip_mgpu_shutdown(c);
    invalidate_graph(c);
    for (uint32_t k = 0; k < slot_count(c); k++) {
        PassSlot* q = slot_at(c, k);
        if (q->done) (void)hipEventDestroy(q->done);
        if (q->accumulated) (void)hipEventDestroy(q->accumulated);
        if (q->hostError) (void)hipHostFree(q->hostError);
        if (k > 0 && q->ownsStream && q->stream) (void)hipStreamDestroy(q->stream);
    }
    c->extra.clear();
    for (auto& t : c->timerPool) {
        if (t.start) (void)hipEventDestroy(t.start);
        if (t.stop) (void)hipEventDestroy(t.stop);
    }
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->ownsStream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->hostStaging) (void)hipHostFree(c->hostStaging);
    for (hipEvent_t e : c->stagingDone)
        if (e) (void)hipEventDestroy(e);
    delete c;
}

// After a synchronisation: did a trace kernel abandon rays (FrameState::errorWord, nx_device.h kStallLimit)?  Read and cleared,
// slot by slot; 4 bytes each.
static int check_device_errors(nxhip_ctx* c)
{
    uint32_t any = 0u;
    for (uint32_t k = 0; k < slot_count(c); k++) {
        PassSlot* const q = slot_at(c, k);
        if (!q->frame.p) continue;  // (a slot that never rendered)
        uint32_t* word = &q->frame.as<FrameState>()->errorWord;
        uint32_t w = 0u;
        // (the caller has synchronised: a pass's own copy of the word is there; anything launched since — the ray-batch hooks — is read here)
        if (q->errorFresh && q->hostError) w = *q->hostError;
        else NX_HIP(hipMemcpy(&w, word, 4, hipMemcpyDeviceToHost));
        q->errorFresh = false;
        if (w) {
            const uint32_t zero = 0u;
            NX_HIP(hipMemcpy(word, &zero, 4, hipMemcpyHostToDevice));
            if (q->hostError) *q->hostError = 0u;
        }
        any |= w;
    }
    if (any & kErrScanStalled) {
        set_error("a workgroup of the ordered compaction gave up waiting for the tile before it (internal error)");
        return NXHIP_ERR_HIP;
    }
    if (any & kErrRaysRetaken) {
        set_error("a trace wave was handed more rays than its launch's queue holds: rays re-enter the queue they were taken from (internal error)");
        return NXHIP_ERR_TRAVERSAL;
    }
    if (any & kErrTraversalStalled) {
        set_error("a trace kernel abandoned rays that made no progress for millions of iterations: the uploaded or device-built BVH is not a tree");
        return NXHIP_ERR_TRAVERSAL;
    }
    return NXHIP_OK;
}

int nxhip_sync(nxhip_ctx* c)
{
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    return check_device_errors(c);
}

int nxhip_sync_timeout(nxhip_ctx* c, uint32_t timeoutMs)
{
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        bool busy = false;
        for (uint32_t k = 0; k < slot_count(c); k++) {
            PassSlot* s = slot_at(c, k);
            if (!s->stream) continue;
            const hipError_t e = hipStreamQuery(s->stream);
            if (e == hipErrorNotReady) busy = true;
            else if (e != hipSuccess) NX_HIP(e);
        }
        if (!busy) return check_device_errors(c);
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() >= (double)timeoutMs) {
            c->dead = true;
            set_error("nxhip_sync_timeout: the device did not finish within " + std::to_string(timeoutMs) + " ms; the context is dead (exit, or continue in a fresh process)");
            return NXHIP_ERR_TIMEOUT;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

int nxhip_resize(nxhip_ctx* c, uint32_t width, uint32_t height)
{
    NX_CHECK_CTX(c);
    if (width == 0 || height == 0) return fail_invalid("nxhip_resize: zero-sized viewport");
    if ((uint64_t)width * height * c->framesPerPass > 0x7fffffffull)
        return fail_invalid("nxhip_resize: more than 2^31 paths (pixels x frames per pass): lower nxhip_set_frames_per_pass first");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    // the new size is committed only when its buffers exist: a failed allocation leaves viewport, pixel map and image as they were
    const uint32_t oldW = c->width, oldH = c->height;
    c->width = width;  // (alloc_paths sizes the image from these)
    c->height = height;
    const int rc = alloc_paths(c, width * height);
    if (rc != NXHIP_OK) {
        c->width = oldW;
        c->height = oldH;
        return rc;
    }
    c->h.camera.resolution[0] = width;
    c->h.camera.resolution[1] = height;
    c->pixelMap.release();
    c->h.pixelMap = nullptr;
    c->stateDirty = true;
    const int rcFrame = set_frame_number_device(c, 0);
    if (rcFrame != NXHIP_OK) return rcFrame;
    // (the ORDER of the full frame survives a resize; a caller's own pixel map does not: it was made for the old size)
    if (c->pixelOrder != NXHIP_ORDER_ROWS) return nxhip_set_pixel_order(c, c->pixelOrder);
    return NXHIP_OK;
}

// ---- scene upload -----------------------------------------------------------------------------------

// 80-byte nodes at a stride of kNodeStride 16-byte chunks (5 = packed as uploaded)
static std::vector<uint4> pad_nodes(const nx_bvh8_node* nodes, uint32_t nodeCount)
{
    std::vector<uint4> out((size_t)nodeCount * kNodeStride, make_uint4(0u, 0u, 0u, 0u));
    for (uint32_t i = 0; i < nodeCount; i++) std::memcpy(&out[(size_t)i * kNodeStride], &nodes[i], sizeof(nx_bvh8_node));
    return out;
}

// The shading copy of a BLAS's triangles when they are kept one per kShadeTriStride bytes (nx_device.h): a strided device copy
// of the packed array.
static int make_shade_tris(BlasHost& b)
{
    if (kShadeTriStride == (int)sizeof(nx_triangle)) return NXHIP_OK;
    NX_ALLOC(b.shadeTris, (size_t)b.triCount * kShadeTriStride);
    NX_HIP(hipMemset(b.shadeTris.p, 0, (size_t)b.triCount * kShadeTriStride));
    NX_HIP(hipMemcpy2D(b.shadeTris.p, kShadeTriStride, b.tris.p, sizeof(nx_triangle), sizeof(nx_triangle), b.triCount, hipMemcpyDeviceToDevice));
    return NXHIP_OK;
}

static int refresh_blas_table(nxhip_ctx* c)
{
    std::vector<BlasDev> table(std::max<size_t>(1, c->blas.size()));
    std::memset(table.data(), 0, table.size() * sizeof(BlasDev));
    for (size_t i = 0; i < c->blas.size(); i++) {
        const BlasHost& b = c->blas[i];
        table[i].nodes = b.nodes.as<uint4>();
        table[i].isect = b.isect.as<float4>();
        table[i].tris = kShadeTriStride == (int)sizeof(nx_triangle) ? b.tris.as<nx_triangle>() : b.shadeTris.as<nx_triangle>();
        table[i].triIdx = b.triIdx.as<uint32_t>();
        table[i].nodeCount = b.nodeCount;
        table[i].triCount = b.triCount;
    }
    NX_SYNC_ALL(c);  // nothing may still read the old table
    NX_ALLOC(c->blasTable, table.size() * sizeof(BlasDev));
    NX_HIP(hipMemcpy(c->blasTable.p, table.data(), table.size() * sizeof(BlasDev), hipMemcpyHostToDevice));
    c->h.blas = c->blasTable.as<BlasDev>();
    c->stateDirty = true;
    c->shadeInstDirty = true;  // (the records hold the BLASes' triangle arrays)
    return NXHIP_OK;
}

// The shading records of the instances (nx_device.h ShadeInst) from the instance, BLAS and material tables.  Called before a
// render when one of them has changed (shadeInstDirty); the cross-table indices have been checked by then (check_scene_ready).
// The matrices come from the DEVICE's instance table: nxhip_set_instance_transforms computes the inverses there.
static int refresh_shade_inst(nxhip_ctx* c)
{
    const size_t n = c->hostInstances.size();
    std::vector<nx_bvh_instance> inst(n);
    NX_SYNC_ALL(c);
    if (n) NX_HIP(hipMemcpy(inst.data(), c->instances.p, n * sizeof(nx_bvh_instance), hipMemcpyDeviceToHost));
    std::vector<ShadeInst> rec(std::max<size_t>(1, n));
    std::memset(rec.data(), 0, rec.size() * sizeof(ShadeInst));
    for (size_t i = 0; i < n; i++) {
        const nx_bvh_instance& in = inst[i];
        if (in.bvhIdx >= c->blas.size()) return fail_invalid("instance refers to a BLAS id that has not been uploaded");
        if (in.materialId < 0 || (size_t)in.materialId >= c->hostMaterialsDev.size()) return fail_invalid("an instance refers to a material id that has not been set");
        const BlasHost& b = c->blas[in.bvhIdx];
        std::memcpy(rec[i].transform, in.transform.cell, sizeof rec[i].transform);
        std::memcpy(rec[i].invTransform, in.invTransform.cell, sizeof rec[i].invTransform);
        rec[i].tris = kShadeTriStride == (int)sizeof(nx_triangle) ? b.tris.as<nx_triangle>() : b.shadeTris.as<nx_triangle>();
        rec[i].triCount = b.triCount;
        rec[i].materialId = in.materialId;
        rec[i].material = c->hostMaterialsDev[(size_t)in.materialId];
    }
    NX_ALLOC(c->shadeInst, rec.size() * sizeof(ShadeInst));
    NX_HIP(hipMemcpy(c->shadeInst.p, rec.data(), rec.size() * sizeof(ShadeInst), hipMemcpyHostToDevice));
    c->h.shadeInst = c->shadeInst.as<ShadeInst>();
    c->stateDirty = true;
    c->shadeInstDirty = false;
    // the traversal records' material codes follow the shading records (nx_refit.hip inst_code_kernel)
    if (c->instTrav.p && !c->hostInstIdx.empty()) {
        const DeviceState* S = c->dState.as<DeviceState>();
        InstTrav* trav = c->instTrav.as<InstTrav>();
        const ShadeInst* si = c->shadeInst.as<ShadeInst>();
        const uint32_t count = (uint32_t)c->hostInstIdx.size();
        void* args[4] = {(void*)&S, (void*)&trav, (void*)&si, (void*)&count};
        NX_HIP(hipLaunchKernel(inst_code_kernel_ptr(), dim3((count + 255u) / 256u), dim3(256), args, 0, c->stream));
        NX_HIP(hipStreamSynchronize(c->stream));
    }
    return NXHIP_OK;
}

static int refresh_inst_trav(nxhip_ctx* c)
{
    // one record per TLAS leaf, in leaf order
    const size_t n = c->hostInstIdx.size();
    std::vector<InstTrav> trav(std::max<size_t>(1, n));
    std::memset(trav.data(), 0, trav.size() * sizeof(InstTrav));
    bool allIdentity = n > 0;
    for (size_t k = 0; k < n; k++) {
        const uint32_t i = c->hostInstIdx[k];
        const nx_bvh_instance& inst = c->hostInstances[i];
        if (inst.bvhIdx >= c->blas.size()) return fail_invalid("instance refers to a BLAS id that has not been uploaded");
        const float* m = inst.invTransform.cell;
        trav[k].r0 = make_float4(m[0], m[1], m[2], m[3]);
        trav[k].r1 = make_float4(m[4], m[5], m[6], m[7]);
        trav[k].r2 = make_float4(m[8], m[9], m[10], m[11]);
        BlasHost& b = c->blas[inst.bvhIdx];
        if (!b.rootKnown) {
            if (b.nodeCount) NX_HIP(hipMemcpy(b.root, b.nodes.p, sizeof b.root, hipMemcpyDeviceToHost));
            b.rootKnown = true;
        }
        trav[k].nodes = b.nodes.as<uint4>();
        trav[k].isect = b.isect.as<float4>();
        trav[k].instIdx = i;
        trav[k].flags = rows_are_identity(m) ? kInstIdentity : 0u;
        allIdentity = allIdentity && trav[k].flags != 0u;
        for (int q = 0; q < 5; q++) trav[k].root[q] = b.root[q];
    }
    NX_SYNC_ALL(c);
    NX_ALLOC(c->instTrav, trav.size() * sizeof(InstTrav));
    NX_HIP(hipMemcpy(c->instTrav.p, trav.data(), trav.size() * sizeof(InstTrav), hipMemcpyHostToDevice));
    c->h.instTrav = c->instTrav.as<InstTrav>();
    c->h.sceneFlags = allIdentity ? kSceneAllIdentity : 0u;
    c->stateDirty = true;
    c->shadeInstDirty = true;  // (the records' material codes are written when the shading records are rebuilt: refresh_shade_inst)
    return NXHIP_OK;
}

// What the traversal kernels assume of 8-wide nodes, checked before anything reaches the GPU (BLAS: primitives = triangles of
// the leaf-ordered list; TLAS: instances).  The kernels decode a slot from its META byte alone — bits 3 and 4 both set: an
// inner child whose hit bit goes to position 24 .. 31 and whose node is childBaseIdx + (imask bits below its slot); otherwise
// a leaf whose (meta >> 5) bits go to position (meta & 31) ... of the 24-bit primitive mask — so the check follows the meta
// bytes: an inner slot must be announced in imask and carry exactly one bit (more would land on other slots' positions and
// index one node past the children), a leaf's bits must stay below position 24 (beyond it they read as inner hits) and
// inside the primitive list, children must exist and follow their parent (no cycles: the traversal would never end).
static const char* wide_node_defect(const nx_bvh8_node& n, uint32_t i, uint32_t nodeCount, uint32_t primCount, bool childrenMustFollow)
{
    int inner = 0, prims = 0;
    for (int s = 0; s < 8; s++) {
        const uint32_t m = n.meta[s];
        if (n.imask & (1u << s)) inner++;
        if ((m & 0x18u) == 0x18u && (m >> 5) != 0u) {  // decoded as an inner child
            if (!(n.imask & (1u << s))) return "a slot is encoded as an inner child but not announced in imask";
            if ((m >> 5) != 1u) return "an inner slot carries more than one hit bit";
            if ((m & 0x07u) != (uint32_t)s) return "an inner slot is encoded with another slot's number";
        } else if (m >> 5) {  // decoded as a leaf of 1 .. 3 primitives at offset (m & 31)
            const int top = 32 - __builtin_clz(m >> 5);
            if ((int)(m & 0x1fu) + top > 24) return "a leaf slot's primitive bits leave the 24-bit primitive mask";
            prims = std::max(prims, (int)(m & 0x1fu) + top);
        }
    }
    if (inner && (uint64_t)n.childBaseIdx + (uint64_t)inner > nodeCount) return "child index out of range";
    if (childrenMustFollow && inner && n.childBaseIdx <= i) return "child nodes must follow their parent";
    if (prims && (uint64_t)n.triangleBaseIdx + (uint64_t)prims > primCount) return "leaf range out of range";
    return nullptr;
}
static const char* wide_nodes_defect(const nx_bvh8_node* nodes, uint32_t nodeCount, uint32_t primCount)
{
    for (uint32_t i = 0; i < nodeCount; i++)
        if (const char* defect = wide_node_defect(nodes[i], i, nodeCount, primCount, true)) return defect;
    return nullptr;
}

int nxhip_upload_blas(nxhip_ctx* c, const nx_bvh8_node* nodes, uint32_t nodeCount, const nx_triangle* tris, uint32_t triCount,
                      const uint32_t* triIdx, int32_t* blasId)
try {
    NX_CHECK_CTX(c);
    if (!nodes || !tris || !triIdx || nodeCount == 0 || triCount == 0) return fail_invalid("nxhip_upload_blas: empty input");
    NX_HIP(hipSetDevice(c->device));
    // validate what the kernels assume before anything reaches the GPU: indices in range
    for (uint32_t i = 0; i < triCount; i++)
        if (triIdx[i] >= triCount) return fail_invalid("nxhip_upload_blas: triangle index out of range");
    if (const char* defect = wide_nodes_defect(nodes, nodeCount, triCount)) return fail_invalid(std::string("nxhip_upload_blas: ") + defect);
    BlasHost b;
    b.nodeCount = nodeCount;
    b.triCount = triCount;
    // leaf-ordered intersection stream: {p0 | original index}, {edge0}, {edge1}; the edges are the same float
    // subtractions the reference performs per test (Triangle.cuh:55-56), done once here
    std::vector<float4> isect((size_t)triCount * kTriStride, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (uint32_t k = 0; k < triCount; k++) {
        const uint32_t t = triIdx[k];
        const nx_triangle& tr = tris[t];
        float idBits;
        std::memcpy(&idBits, &t, 4);
        isect[kTriStride * (size_t)k + 0] = make_float4(tr.pos0[0], tr.pos0[1], tr.pos0[2], idBits);
        isect[kTriStride * (size_t)k + 1] = make_float4(tr.pos1[0] - tr.pos0[0], tr.pos1[1] - tr.pos0[1], tr.pos1[2] - tr.pos0[2], 0.0f);
        isect[kTriStride * (size_t)k + 2] = make_float4(tr.pos2[0] - tr.pos0[0], tr.pos2[1] - tr.pos0[1], tr.pos2[2] - tr.pos0[2], 0.0f);
    }
    std::vector<uint4> padded = pad_nodes(nodes, nodeCount);
    NX_ALLOC(b.nodes, padded.size() * sizeof(uint4));
    NX_ALLOC(b.isect, isect.size() * sizeof(float4));
    NX_ALLOC(b.tris, (size_t)triCount * sizeof(nx_triangle));
    NX_ALLOC(b.triIdx, (size_t)triCount * 4);
    NX_HIP(hipMemcpy(b.nodes.p, padded.data(), padded.size() * sizeof(uint4), hipMemcpyHostToDevice));
    NX_HIP(hipMemcpy(b.isect.p, isect.data(), isect.size() * sizeof(float4), hipMemcpyHostToDevice));
    NX_HIP(hipMemcpy(b.tris.p, tris, (size_t)triCount * sizeof(nx_triangle), hipMemcpyHostToDevice));
    NX_HIP(hipMemcpy(b.triIdx.p, triIdx, (size_t)triCount * 4, hipMemcpyHostToDevice));
    if (const int rcs = make_shade_tris(b)) return rcs;
    c->blas.push_back(std::move(b));
    if (blasId) *blasId = (int32_t)c->blas.size() - 1;
    return refresh_blas_table(c);
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_upload_blas: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_build_blas(nxhip_ctx* c, const nx_triangle* tris, uint32_t triCount, int32_t* blasId)
try {
    NX_CHECK_CTX(c);
    if (!tris || triCount == 0) return fail_invalid("nxhip_build_blas: empty input");
    if (kNodeStride != 5) return fail_invalid("nxhip_build_blas: built with padded node records");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    BlasHost b;
    b.triCount = triCount;
    NX_ALLOC(b.tris, (size_t)triCount * sizeof(nx_triangle));
    NX_HIP(hipMemcpy(b.tris.p, tris, (size_t)triCount * sizeof(nx_triangle), hipMemcpyHostToDevice));
    DevBuf wide;
    uint32_t nodeCount = 0;
    const int rc = lbvh_build(c, b.tris.as<nx_triangle>(), triCount, c->deviceBuilderRadius, wide, b.triIdx, b.isect, &nodeCount);
    if (rc != NXHIP_OK) return rc;
    NX_ALLOC(b.nodes, (size_t)nodeCount * sizeof(nx_bvh8_node));  // the builder's array is sized for the worst case
    NX_HIP(hipMemcpy(b.nodes.p, wide.p, (size_t)nodeCount * sizeof(nx_bvh8_node), hipMemcpyDeviceToDevice));
    b.nodeCount = nodeCount;
    if (const int rcs = make_shade_tris(b)) return rcs;
    c->blas.push_back(std::move(b));
    if (blasId) *blasId = (int32_t)c->blas.size() - 1;
    return refresh_blas_table(c);
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_build_blas: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_build_blas_batch(nxhip_ctx* c, const nx_triangle* const* tris, const uint32_t* triCounts, uint32_t meshCount, int32_t* blasIds)
try {
    NX_CHECK_CTX(c);
    if (!tris || !triCounts || meshCount == 0) return fail_invalid("nxhip_build_blas_batch: empty input");
    if (kNodeStride != 5) return fail_invalid("nxhip_build_blas_batch: built with padded node records");
    uint64_t total = 0;
    for (uint32_t m = 0; m < meshCount; m++) {
        if (!tris[m] || triCounts[m] == 0) return fail_invalid("nxhip_build_blas_batch: a mesh without triangles");
        total += triCounts[m];
    }
    if (total > 0x7fffffffull) return fail_invalid("nxhip_build_blas_batch: more than 2^31 triangles in one batch");
    if (c->deviceBuilderRadius != NXHIP_BUILDER_SAH || meshCount == 1) {
        // the other builders (radix tree, clustering) have no forest form: one build per mesh
        for (uint32_t m = 0; m < meshCount; m++) {
            int32_t id = -1;
            const int rc = nxhip_build_blas(c, tris[m], triCounts[m], &id);
            if (rc != NXHIP_OK) return rc;
            if (blasIds) blasIds[m] = id;
        }
        return NXHIP_OK;
    }
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    // NX_TUNING_KNOBS=1 NX_BATCH_TIMING=1: where the call's time goes, to stderr (tools / tests only)
    const bool timing = std::getenv("NX_TUNING_KNOBS") && std::atoi(std::getenv("NX_TUNING_KNOBS")) == 1 && std::getenv("NX_BATCH_TIMING");
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto lap = [&](const char* what, std::chrono::steady_clock::time_point& t0) {
        if (!timing) return;
        (void)hipStreamSynchronize(c->stream);
        const auto t1 = now();
        std::fprintf(stderr, "[nxhip_build_blas_batch] %-34s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    auto tLap = now();
    // the triangles of all meshes, concatenated, through a pinned staging buffer the context keeps (one transfer instead of one
    // per mesh)
    const size_t bytes = (size_t)total * sizeof(nx_triangle);
    constexpr size_t kHalf = (size_t)16 << 20;  // two halves of a 32 MiB pinned buffer, allocated once per context: a staging buffer
                                                // as large as the batch would cost more to pin than the transfer takes
    if (!c->hostStaging) {
        NX_HIP(hipHostMalloc(&c->hostStaging, 2 * kHalf, hipHostMallocDefault));
        c->hostStagingBytes = 2 * kHalf;
        NX_HIP(hipEventCreateWithFlags(&c->stagingDone[0], hipEventDisableTiming));
        NX_HIP(hipEventCreateWithFlags(&c->stagingDone[1], hipEventDisableTiming));
    }
    std::vector<uint32_t> counts(triCounts, triCounts + meshCount);
    auto trisPool = std::make_shared<DevBuf>();
    auto nodesPool = std::make_shared<DevBuf>(), idxPool = std::make_shared<DevBuf>(), isectPool = std::make_shared<DevBuf>();
    if (!trisPool->alloc(bytes)) return NXHIP_ERR_HIP;
    {
        // the meshes as one byte stream through the two halves: while one half is on its way to the device the other is filled
        size_t sent = 0, inHalf = 0;
        int half = 0;
        bool used[2] = {false, false};
        char* const base = static_cast<char*>(c->hostStaging);
        auto flush = [&]() -> int {
            if (inHalf == 0) return NXHIP_OK;
            NX_HIP(hipMemcpyAsync(static_cast<char*>(trisPool->p) + sent, base + (size_t)half * kHalf, inHalf, hipMemcpyHostToDevice, c->stream));
            NX_HIP(hipEventRecord(c->stagingDone[half], c->stream));
            used[half] = true;
            sent += inHalf;
            inHalf = 0;
            half ^= 1;
            if (used[half]) NX_HIP(hipEventSynchronize(c->stagingDone[half]));  // the half about to be refilled has left
            return NXHIP_OK;
        };
        for (uint32_t m = 0; m < meshCount; m++) {
            const char* src = reinterpret_cast<const char*>(tris[m]);
            size_t left = (size_t)counts[m] * sizeof(nx_triangle);
            while (left) {
                const size_t take = std::min(left, kHalf - inHalf);
                std::memcpy(base + (size_t)half * kHalf + inHalf, src, take);
                inHalf += take;
                src += take;
                left -= take;
                if (inHalf == kHalf)
                    if (const int rcf = flush()) return rcf;
            }
        }
        if (const int rcf = flush()) return rcf;
    }
    std::vector<uint32_t> nodeFirst, nodeCounts;
    lap("triangles through pinned staging", tLap);
    const int rc = lbvh_build_batch(c, trisPool->as<nx_triangle>(), counts, *nodesPool, *idxPool, *isectPool, nodeFirst, nodeCounts);
    if (rc != NXHIP_OK) return rc;
    lap("device build", tLap);
    // every mesh's root node, for the instance records (refresh_inst_trav would otherwise fetch them one by one)
    std::vector<nx_bvh8_node> allNodes(nodesPool->bytes / sizeof(nx_bvh8_node));
    NX_HIP(hipMemcpy(allNodes.data(), nodesPool->p, allNodes.size() * sizeof(nx_bvh8_node), hipMemcpyDeviceToHost));
    size_t first = 0;
    for (uint32_t m = 0; m < meshCount; m++) {
        BlasHost b;
        b.triCount = counts[m];
        b.nodeCount = nodeCounts[m];
        b.nodes = DevBuf::view(nodesPool, (size_t)nodeFirst[m] * sizeof(nx_bvh8_node), (size_t)nodeCounts[m] * sizeof(nx_bvh8_node));
        b.isect = DevBuf::view(isectPool, first * kTriStride * sizeof(float4), (size_t)counts[m] * kTriStride * sizeof(float4));
        b.tris = DevBuf::view(trisPool, first * sizeof(nx_triangle), (size_t)counts[m] * sizeof(nx_triangle));
        b.triIdx = DevBuf::view(idxPool, first * 4, (size_t)counts[m] * 4);
        std::memcpy(b.root, &allNodes[nodeFirst[m]], sizeof b.root);
        b.rootKnown = true;
        if (const int rcs = make_shade_tris(b)) return rcs;
        c->blas.push_back(std::move(b));
        if (blasIds) blasIds[m] = (int32_t)c->blas.size() - 1;
        first += counts[m];
    }
    lap("roots read back, BLAS records", tLap);
    const int rcTable = refresh_blas_table(c);
    lap("BLAS table", tLap);
    return rcTable;
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_build_blas_batch: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_read_blas_batch(nxhip_ctx* c, int32_t firstBlasId, uint32_t count, nx_bvh8_node* nodes, uint32_t nodeCapacity, uint32_t* nodeCounts, uint32_t* primIdx, uint32_t primCapacity)
{
    NX_CHECK_CTX(c);
    if (firstBlasId < 0 || (size_t)firstBlasId + count > c->blas.size()) return fail_invalid("nxhip_read_blas_batch: no such BLAS range");
    if (count == 0) return NXHIP_OK;
    if (kNodeStride != 5) return fail_invalid("nxhip_read_blas_batch: built with padded node records");
    uint64_t nodeTotal = 0, primTotal = 0;
    bool oneRun = true;  // the BLASes of one nxhip_build_blas_batch call lie back to back in their pools: one copy each for nodes and indices
    for (uint32_t k = 0; k < count; k++) {
        const BlasHost& b = c->blas[(size_t)firstBlasId + k];
        if (nodeCounts) nodeCounts[k] = b.nodeCount;
        if (k) {
            const BlasHost& a = c->blas[(size_t)firstBlasId + k - 1];
            oneRun = oneRun && a.nodes.pool && a.nodes.pool == b.nodes.pool && static_cast<char*>(a.nodes.p) + a.nodes.bytes == b.nodes.p &&
                     a.triIdx.pool == b.triIdx.pool && static_cast<char*>(a.triIdx.p) + a.triIdx.bytes == b.triIdx.p;
        }
        nodeTotal += b.nodeCount;
        primTotal += b.triCount;
    }
    if ((nodes && nodeCapacity < nodeTotal) || (primIdx && primCapacity < primTotal)) return fail_invalid("nxhip_read_blas_batch: destination too small");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    const BlasHost& b0 = c->blas[(size_t)firstBlasId];
    if (oneRun) {
        if (nodes) NX_HIP(hipMemcpy(nodes, b0.nodes.p, (size_t)nodeTotal * sizeof(nx_bvh8_node), hipMemcpyDeviceToHost));
        if (primIdx) NX_HIP(hipMemcpy(primIdx, b0.triIdx.p, (size_t)primTotal * 4, hipMemcpyDeviceToHost));
        return NXHIP_OK;
    }
    size_t nodeAt = 0, primAt = 0;
    for (uint32_t k = 0; k < count; k++) {
        const BlasHost& b = c->blas[(size_t)firstBlasId + k];
        if (nodes) NX_HIP(hipMemcpy(nodes + nodeAt, b.nodes.p, (size_t)b.nodeCount * sizeof(nx_bvh8_node), hipMemcpyDeviceToHost));
        if (primIdx) NX_HIP(hipMemcpy(primIdx + primAt, b.triIdx.p, (size_t)b.triCount * 4, hipMemcpyDeviceToHost));
        nodeAt += b.nodeCount;
        primAt += b.triCount;
    }
    return NXHIP_OK;
}

int nxhip_set_device_builder(nxhip_ctx* c, int clusteringRadius)
{
    NX_CHECK_CTX(c);
    if (clusteringRadius < NXHIP_BUILDER_SAH || clusteringRadius > 256) return fail_invalid("nxhip_set_device_builder: NXHIP_BUILDER_SAH (-1), 0 (radix tree) or a clustering radius up to 256");
    c->deviceBuilderRadius = clusteringRadius;
    return NXHIP_OK;
}

int nxhip_read_blas(nxhip_ctx* c, int32_t blasId, nx_bvh8_node* nodes, uint32_t nodeCapacity, uint32_t* primIdx, uint32_t primCapacity, uint32_t* nodeCount)
{
    NX_CHECK_CTX(c);
    if (blasId < 0 || (size_t)blasId >= c->blas.size()) return fail_invalid("nxhip_read_blas: no such BLAS");
    const BlasHost& b = c->blas[(size_t)blasId];
    if (nodeCount) *nodeCount = b.nodeCount;
    if ((nodes && nodeCapacity < b.nodeCount) || (primIdx && primCapacity < b.triCount)) return fail_invalid("nxhip_read_blas: destination too small");
    if (kNodeStride != 5) return fail_invalid("nxhip_read_blas: built with padded node records");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    if (nodes) NX_HIP(hipMemcpy(nodes, b.nodes.p, (size_t)b.nodeCount * sizeof(nx_bvh8_node), hipMemcpyDeviceToHost));
    if (primIdx) NX_HIP(hipMemcpy(primIdx, b.triIdx.p, (size_t)b.triCount * 4, hipMemcpyDeviceToHost));
    return NXHIP_OK;
}

int nxhip_debug_write_blas_node(nxhip_ctx* c, int32_t blasId, uint32_t nodeIdx, const nx_bvh8_node* node)
{
    NX_DEBUG_HOOK("nxhip_debug_write_blas_node");  // (first: a release library refuses whatever it is handed)
    NX_CHECK_CTX(c);
    if (!node || blasId < 0 || (size_t)blasId >= c->blas.size()) return fail_invalid("nxhip_debug_write_blas_node: no such BLAS");
    BlasHost& b = c->blas[(size_t)blasId];
    if (nodeIdx >= b.nodeCount) return fail_invalid("nxhip_debug_write_blas_node: no such node");
    // the upload checks, as the traversal decodes a node (meta bytes), minus "children follow their parent": a node that points back
    // at itself is what the hook exists for (the stall guard's test); everything that could index past an array is refused
    if (const char* defect = wide_node_defect(*node, nodeIdx, b.nodeCount, b.triCount, false)) return fail_invalid(std::string("nxhip_debug_write_blas_node: ") + defect);
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    NX_HIP(hipMemcpy(b.nodes.as<uint4>() + (size_t)nodeIdx * kNodeStride, node, sizeof(nx_bvh8_node), hipMemcpyHostToDevice));
    if (nodeIdx == 0) {  // the root is also embedded in the instance records
        std::memcpy(b.root, node, sizeof b.root);
        b.rootKnown = true;
        if (!c->hostInstIdx.empty()) return refresh_inst_trav(c);
    }
    return NXHIP_OK;
}

int nxhip_debug_set_scan_epoch(nxhip_ctx* c, uint32_t epoch)
{
    NX_DEBUG_HOOK("nxhip_debug_set_scan_epoch");  // (first: a release library refuses whatever it is handed)
    NX_CHECK_CTX(c);
    NX_SYNC_ALL(c);
    for (uint32_t k = 0; k < slot_count(c); k++) slot_at(c, k)->scanEpoch = std::min(epoch, kScanEpochLimit - 1u);
    return NXHIP_OK;
}

int nxhip_clear_blas(nxhip_ctx* c)
try {
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    c->blas.clear();
    c->tlasTightBoxes.release();
    c->hostInstances.clear();
    c->hostInstIdx.clear();
    c->h.tlasNodes = nullptr;
    c->h.instanceCount = 0;
    return refresh_blas_table(c);
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_clear_blas: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_set_tlas(nxhip_ctx* c, const nx_bvh8_node* nodes, uint32_t nodeCount, const uint32_t* instanceIdx, const nx_bvh_instance* instances,
                   uint32_t instanceCount)
try {
    NX_CHECK_CTX(c);
    if (!nodes || !instanceIdx || !instances || nodeCount == 0 || instanceCount == 0) return fail_invalid("nxhip_set_tlas: empty input");
    if (instanceCount > kHitInstMask) return fail_invalid("nxhip_set_tlas: more than 2^29 - 1 instances (a hit record keeps the instance in 29 bits)");
    NX_HIP(hipSetDevice(c->device));
    for (uint32_t i = 0; i < instanceCount; i++) {
        if (instanceIdx[i] >= instanceCount) return fail_invalid("nxhip_set_tlas: instance index out of range");
        if (instances[i].bvhIdx >= c->blas.size()) return fail_invalid("nxhip_set_tlas: instance refers to a BLAS id that has not been uploaded");
    }
    // (checked here, before the context is touched: a failure must leave the previous TLAS and its traversal records in place)
    if (const char* defect = wide_nodes_defect(nodes, nodeCount, instanceCount)) return fail_invalid(std::string("nxhip_set_tlas: ") + defect);
    NX_SYNC_ALL(c);
    c->tlasTightBoxes.release();  // (a tree from outside: its boxes are the records')
    std::vector<uint4> padded = pad_nodes(nodes, nodeCount);
    NX_ALLOC(c->tlasNodes, padded.size() * sizeof(uint4));
    NX_ALLOC(c->tlasInstIdx, (size_t)instanceCount * 4);
    NX_ALLOC(c->instances, (size_t)instanceCount * sizeof(nx_bvh_instance));
    NX_HIP(hipMemcpy(c->tlasNodes.p, padded.data(), padded.size() * sizeof(uint4), hipMemcpyHostToDevice));
    NX_HIP(hipMemcpy(c->tlasInstIdx.p, instanceIdx, (size_t)instanceCount * 4, hipMemcpyHostToDevice));
    NX_HIP(hipMemcpy(c->instances.p, instances, (size_t)instanceCount * sizeof(nx_bvh_instance), hipMemcpyHostToDevice));
    c->hostInstances.assign(instances, instances + instanceCount);
    c->hostInstIdx.assign(instanceIdx, instanceIdx + instanceCount);
    c->h.tlasNodes = c->tlasNodes.as<uint4>();
    c->h.tlasInstIdx = c->tlasInstIdx.as<uint32_t>();
    c->h.instances = c->instances.as<nx_bvh_instance>();
    c->h.instanceCount = instanceCount;
    c->stateDirty = true;
    c->shadeInstDirty = true;
    {
        // schedule of the device-side refit (nxhip_set_instance_transforms): node indices grouped by depth, deepest first
        std::vector<uint32_t> depth(nodeCount, 0u);
        uint32_t maxDepth = 0;
        for (uint32_t i = 0; i < nodeCount; i++) {  // children follow their parent in the array: one ascending sweep
            const nx_bvh8_node& n = nodes[i];
            const uint32_t inner = (uint32_t)__builtin_popcount(n.imask);
            for (uint32_t k = 0; k < inner; k++) {
                depth[n.childBaseIdx + k] = depth[i] + 1;
                maxDepth = std::max(maxDepth, depth[i] + 1);
            }
        }
        std::vector<uint32_t> levelStart(maxDepth + 2, 0u), order(nodeCount);
        for (uint32_t i = 0; i < nodeCount; i++) levelStart[(maxDepth - depth[i]) + 1]++;
        for (uint32_t l = 0; l <= maxDepth; l++) levelStart[l + 1] += levelStart[l];
        std::vector<uint32_t> cursor(levelStart.begin(), levelStart.end() - 1);
        for (uint32_t i = 0; i < nodeCount; i++) order[cursor[maxDepth - depth[i]]++] = i;
        std::vector<uint32_t> leafOf(instanceCount, 0u);
        for (uint32_t k = 0; k < instanceCount; k++) leafOf[instanceIdx[k]] = k;
        NX_ALLOC(c->refitOrder, (size_t)nodeCount * 4);
        NX_ALLOC(c->refitLevelStart, levelStart.size() * 4);
        NX_ALLOC(c->leafOfInstance, (size_t)instanceCount * 4);
        NX_ALLOC(c->refitBoxes, (size_t)nodeCount * 24);
        NX_HIP(hipMemcpy(c->refitOrder.p, order.data(), order.size() * 4, hipMemcpyHostToDevice));
        NX_HIP(hipMemcpy(c->refitLevelStart.p, levelStart.data(), levelStart.size() * 4, hipMemcpyHostToDevice));
        NX_HIP(hipMemcpy(c->leafOfInstance.p, leafOf.data(), leafOf.size() * 4, hipMemcpyHostToDevice));
        c->h.leafOfInstance = c->leafOfInstance.as<uint32_t>();  // (also what a handed-over ray names its instance record by: ThinState::leaf)
        c->refitLevels = maxDepth + 1;
        c->tlasNodeCount = nodeCount;
    }
    return refresh_inst_trav(c);
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_set_tlas: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_rebuild_tlas(nxhip_ctx* c, const nx_bvh_instance* instances, uint32_t instanceCount)
try {
    NX_CHECK_CTX(c);
    if (!instances || instanceCount == 0) return fail_invalid("nxhip_rebuild_tlas: empty input");
    if (kNodeStride != 5) return fail_invalid("nxhip_rebuild_tlas: built with padded node records");
    for (uint32_t i = 0; i < instanceCount; i++)
        if (instances[i].bvhIdx >= c->blas.size()) return fail_invalid("nxhip_rebuild_tlas: instance refers to a BLAS id that has not been uploaded");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    DevBuf dInst, wide, primIdx, boxes;
    bool boxesAreTight = false;
    NX_ALLOC(dInst, (size_t)instanceCount * sizeof(nx_bvh_instance));
    NX_HIP(hipMemcpy(dInst.p, instances, (size_t)instanceCount * sizeof(nx_bvh_instance), hipMemcpyHostToDevice));
    uint32_t nodeCount = 0;
    int rc = lbvh_build_tlas(c, dInst.as<nx_bvh_instance>(), instanceCount, c->deviceBuilderRadius, wide, primIdx, boxes, &boxesAreTight, &nodeCount);
    if (rc != NXHIP_OK) return rc;
    // The tree is a few hundred nodes per thousand instances: it comes back once so that nxhip_set_tlas — range checks, the
    // traversal records in leaf order, the schedule of the device-side refit — installs it like any other TLAS.
    std::vector<nx_bvh8_node> nodes(nodeCount);
    std::vector<uint32_t> idx(instanceCount);
    NX_HIP(hipMemcpy(nodes.data(), wide.p, (size_t)nodeCount * sizeof(nx_bvh8_node), hipMemcpyDeviceToHost));
    NX_HIP(hipMemcpy(idx.data(), primIdx.p, (size_t)instanceCount * 4, hipMemcpyDeviceToHost));
    rc = nxhip_set_tlas(c, nodes.data(), nodeCount, idx.data(), instances, instanceCount);
    // the boxes the tree was built from stay with it: the device-side refit keeps using them instead of the records' looser ones
    if (rc == NXHIP_OK && boxesAreTight) c->tlasTightBoxes = std::move(boxes);
    return rc;
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_rebuild_tlas: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_read_tlas_index(nxhip_ctx* c, uint32_t* instanceIdx, uint32_t capacity, uint32_t* nodeCount)
{
    NX_CHECK_CTX(c);
    if (!c->h.tlasNodes) return fail_invalid("nxhip_read_tlas_index: no TLAS has been set");
    if (nodeCount) *nodeCount = c->tlasNodeCount;
    if (instanceIdx) {
        if (capacity < c->hostInstIdx.size()) return fail_invalid("nxhip_read_tlas_index: destination too small");
        std::memcpy(instanceIdx, c->hostInstIdx.data(), c->hostInstIdx.size() * 4);
    }
    return NXHIP_OK;
}

int nxhip_set_instance_transforms(nxhip_ctx* c, const uint32_t* instanceIds, const float* transforms16, uint32_t count)
try {
    NX_CHECK_CTX(c);
    if (count == 0) return NXHIP_OK;
    if (!instanceIds || !transforms16) return fail_invalid("nxhip_set_instance_transforms: null argument");
    if (!c->h.tlasNodes || c->refitLevels == 0) return fail_invalid("nxhip_set_instance_transforms: no TLAS has been set");
    for (uint32_t i = 0; i < count; i++)
        if (instanceIds[i] >= c->h.instanceCount) return fail_invalid("nxhip_set_instance_transforms: instance id out of range");
    NX_HIP(hipSetDevice(c->device));
    int rc = upload_state(c);
    if (rc != NXHIP_OK) return rc;
    if (slot_count(c) > 1) NX_SYNC_ALL(c);  // passes on the other slots' streams still traverse the old placement
    if (c->refitIds.bytes < (size_t)count * 4) NX_ALLOC(c->refitIds, (size_t)count * 4);
    if (c->refitMatrices.bytes < (size_t)count * 64) NX_ALLOC(c->refitMatrices, (size_t)count * 64);
    // stream order does the rest: a frame already in flight finishes with the old placement, the next one sees the new
    NX_HIP(hipMemcpyAsync(c->refitIds.p, instanceIds, (size_t)count * 4, hipMemcpyHostToDevice, c->stream));
    NX_HIP(hipMemcpyAsync(c->refitMatrices.p, transforms16, (size_t)count * 64, hipMemcpyHostToDevice, c->stream));
    {
        const DeviceState* S = c->dState.as<DeviceState>();
        nx_bvh_instance* inst = c->instances.as<nx_bvh_instance>();
        InstTrav* trav = c->instTrav.as<InstTrav>();
        const uint32_t* leafOf = c->leafOfInstance.as<uint32_t>();
        const uint32_t* ids = c->refitIds.as<uint32_t>();
        const float* mats = c->refitMatrices.as<float>();
        void* tight = c->tlasTightBoxes.p;
        // (the shading records follow the matrices when they are current; a stale set is rebuilt from the device's instance table)
        ShadeInst* shadeInst = c->shadeInstDirty ? nullptr : c->shadeInst.as<ShadeInst>();
        void* args[9] = {(void*)&S, (void*)&inst, (void*)&trav, (void*)&leafOf, (void*)&ids, (void*)&mats, (void*)&count, (void*)&tight, (void*)&shadeInst};
        const unsigned grid = std::min<unsigned>((count + 255u) / 256u, (unsigned)c->wideBlocks);
        NX_HIP(hipLaunchKernel(instance_transform_kernel_ptr(), dim3(grid), dim3(256), args, 0, c->stream));
    }
    {
        nx_bvh8_node* nodes = c->tlasNodes.as<nx_bvh8_node>();
        const uint32_t* primIdx = c->tlasInstIdx.as<uint32_t>();
        const nx_bvh_instance* inst = c->instances.as<nx_bvh_instance>();
        const uint32_t* order = c->refitOrder.as<uint32_t>();
        const uint32_t* levelStart = c->refitLevelStart.as<uint32_t>();
        const uint32_t levels = c->refitLevels;
        void* boxes = c->refitBoxes.p;
        void* tight = c->tlasTightBoxes.p;
        void* args[8] = {(void*)&nodes, (void*)&primIdx, (void*)&inst, (void*)&order, (void*)&levelStart, (void*)&levels, (void*)&boxes, (void*)&tight};
        NX_HIP(hipLaunchKernel(tlas_refit_kernel_ptr(), dim3(1), dim3(1024), args, 0, c->stream));
    }
    // pageable host arrays: the copies above are staged before hipMemcpyAsync returns on this runtime, but that is not a
    // documented guarantee — wait, the call is not on the per-frame path
    NX_SYNC_ALL(c);
    for (uint32_t i = 0; i < count; i++) std::memcpy(c->hostInstances[instanceIds[i]].transform.cell, transforms16 + 16 * (size_t)i, 64);
    // The kernel has set each moved record's identity flag from the inverse it computed.  The scene-wide "no instance transforms
    // a ray" flag is the host's to keep: it survives only if every new matrix is the identity itself (whose inverse, by the
    // cofactor formula, is the identity bit for bit).
    if (c->h.sceneFlags & kSceneAllIdentity) {
        bool still = true;
        for (uint32_t i = 0; i < count && still; i++) {
            const float* m = transforms16 + 16 * (size_t)i;
            still = rows_are_identity(m) && m[12] == 0.0f && m[13] == 0.0f && m[14] == 0.0f && m[15] == 1.0f;
        }
        if (!still) {
            c->h.sceneFlags &= ~kSceneAllIdentity;
            c->stateDirty = true;
        }
    }
    return NXHIP_OK;
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_set_instance_transforms: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_read_tlas(nxhip_ctx* c, nx_bvh8_node* nodes, uint32_t nodeCapacity, nx_bvh_instance* instances, uint32_t instanceCapacity)
{
    NX_CHECK_CTX(c);
    if (!c->h.tlasNodes) return fail_invalid("nxhip_read_tlas: no TLAS has been set");
    if ((nodes && nodeCapacity < c->tlasNodeCount) || (instances && instanceCapacity < c->h.instanceCount)) return fail_invalid("nxhip_read_tlas: destination too small");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    if (nodes) {
        if (kNodeStride == 5) NX_HIP(hipMemcpy(nodes, c->tlasNodes.p, (size_t)c->tlasNodeCount * sizeof(nx_bvh8_node), hipMemcpyDeviceToHost));
        else return fail_invalid("nxhip_read_tlas: built with padded node records");
    }
    if (instances) NX_HIP(hipMemcpy(instances, c->instances.p, (size_t)c->h.instanceCount * sizeof(nx_bvh_instance), hipMemcpyDeviceToHost));
    return NXHIP_OK;
}

int nxhip_set_materials(nxhip_ctx* c, const nx_material* materials, uint32_t count)
try {
    NX_CHECK_CTX(c);
    if (!materials || count == 0) return fail_invalid("nxhip_set_materials: empty input");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    NX_ALLOC(c->materials, (size_t)count * sizeof(nx_material));
    c->hostMaterials.assign(materials, materials + count);
    // The device copy carries one derived flag in the padding byte behind `type` (offset 57 of the 60-byte record): the
    // material can emit or let a path pass through, i.e. its shading may look at / must keep the path's previous vertex
    // (nx_wavefront.hip keep_previous_vertex).  The logic kernel reads type and flag with the one load it already does.
    std::vector<nx_material> dev(materials, materials + count);
    for (nx_material& m : dev) {
        // "can emit" exactly as shade_path tests it: maxcomp3(emissive * intensity) > 0 (a negative intensity with a negative
        // component emits too)
        const bool flag = m.emissiveMapId != -1 || m.diffuseMapId != -1 || m.opacity < 1.0f ||
                          std::max(std::max(m.emissive[0] * m.intensity, m.emissive[1] * m.intensity), m.emissive[2] * m.intensity) > 0.0f;
        reinterpret_cast<unsigned char*>(&m)[kMaterialFlagOffset] = flag ? 1u : 0u;
    }
    NX_HIP(hipMemcpy(c->materials.p, dev.data(), (size_t)count * sizeof(nx_material), hipMemcpyHostToDevice));
    c->h.materials = c->materials.as<nx_material>();
    c->hostMaterialsDev = dev;
    c->stateDirty = true;
    c->shadeInstDirty = true;  // (the records hold a copy of their instance's material)
    uint32_t mask = 0u;
    for (const nx_material& m : dev)
        if (m.type >= 0 && m.type <= 3) mask |= 1u << m.type;
    if (mask != c->materialTypeMask) {  // the pass graphs hold one material kernel per type in use
        c->materialTypeMask = mask;
        invalidate_graph(c);
    }
    return NXHIP_OK;
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_set_materials: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_set_lights(nxhip_ctx* c, const nx_light* lights, uint32_t count)
try {
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    NX_ALLOC(c->lights, std::max<size_t>(1, count) * sizeof(nx_light));
    if (count) NX_HIP(hipMemcpy(c->lights.p, lights, (size_t)count * sizeof(nx_light), hipMemcpyHostToDevice));
    c->hostLights.assign(lights, lights + (lights ? count : 0));
    c->h.lights = c->lights.as<nx_light>();
    c->h.lightCount = count;
    c->stateDirty = true;
    return NXHIP_OK;
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_set_lights: ") + e.what());
    return NXHIP_ERR_INVALID;
}

static int refresh_texture_tables(nxhip_ctx* c)
{
    auto build = [&](std::vector<TextureHost>& v, DevBuf& table, const TextureDev*& dst) -> int {
        std::vector<TextureDev> t(std::max<size_t>(1, v.size()));
        std::memset(t.data(), 0, t.size() * sizeof(TextureDev));
        for (size_t i = 0; i < v.size(); i++) t[i] = TextureDev{v[i].texels.as<uint32_t>(), v[i].width, v[i].height};
        NX_ALLOC(table, t.size() * sizeof(TextureDev));
        NX_HIP(hipMemcpy(table.p, t.data(), t.size() * sizeof(TextureDev), hipMemcpyHostToDevice));
        dst = table.as<TextureDev>();
        return NXHIP_OK;
    };
    NX_SYNC_ALL(c);
    int rc = build(c->diffuseMaps, c->diffuseTable, c->h.diffuseMaps);
    if (rc != NXHIP_OK) return rc;
    rc = build(c->emissiveMaps, c->emissiveTable, c->h.emissiveMaps);
    if (rc != NXHIP_OK) return rc;
    c->h.hdrMap = TextureDev{c->hdrMap.texels.as<uint32_t>(), c->hdrMap.width, c->hdrMap.height};
    c->stateDirty = true;
    return NXHIP_OK;
}

// Sampling distribution of the environment map (see nx_wavefront.hip, "Environment importance sampling"): texel weight =
// luminance of the sRGB-decoded texel x sin(polar angle of its row) + 1e-6, accumulated in double; cdfs as float ending in
// exactly 1; density = weight / total x width x height / (2 pi^2) = pdf per solid angle x cos(latitude).
static int build_env_tables(nxhip_ctx* c)
{
    const uint32_t W = c->hdrMap.width, H = c->hdrMap.height;
    if (!c->envSampling || W == 0 || H == 0 || c->hostHdr.size() != (size_t)W * H * 4) {
        c->h.envSampling = 0;
        c->h.envMarginalCdf = c->h.envRowCdf = c->h.envDensity = nullptr;
        c->h.envMarginalGuide = c->h.envRowGuide = nullptr;
        c->stateDirty = true;
        return NXHIP_OK;
    }
    float lut[256];
    for (int i = 0; i < 256; i++) {
        const float x = (float)i / 255.0f;
        lut[i] = x <= 0.04045f ? x / 12.92f : std::pow((x + 0.055f) / 1.055f, 2.4f);
    }
    const double pi = 3.14159265358979323846;
    std::vector<float> marginal(H), row((size_t)W * H), density((size_t)W * H);
    std::vector<double> rowSum(H);
    double total = 0.0;
    for (uint32_t y = 0; y < H; y++) {
        const double sinTheta = std::sin(pi * ((double)y + 0.5) / (double)H);
        double run = 0.0;
        for (uint32_t x = 0; x < W; x++) {
            const uint8_t* t = &c->hostHdr[4 * ((size_t)y * W + x)];
            const double lum = 0.2126 * (double)lut[t[0]] + 0.7152 * (double)lut[t[1]] + 0.0722 * (double)lut[t[2]];
            const double wgt = lum * sinTheta + 1e-6;
            density[(size_t)y * W + x] = (float)wgt;
            run += wgt;
            row[(size_t)y * W + x] = (float)run;
        }
        rowSum[y] = run;
        total += run;
    }
    double run = 0.0;
    for (uint32_t y = 0; y < H; y++) {
        for (uint32_t x = 0; x < W; x++) {
            const size_t i = (size_t)y * W + x;
            row[i] = x == W - 1 ? 1.0f : (float)((double)row[i] / rowSum[y]);
            density[i] = (float)((double)density[i] / total * (double)W * (double)H / (2.0 * pi * pi));
        }
        run += rowSum[y];
        marginal[y] = y == H - 1 ? 1.0f : (float)(run / total);
    }
    // guides for the device's cdf inversion (nx_wavefront.hip cdf_find): bracket per bucket of the random number
    auto make_guide = [](const float* cdf, uint32_t n, uint32_t* guide) {
        uint32_t idx = 0;
        for (int b = 0; b <= kEnvGuide; b++) {
            const float bound = (float)b / (float)kEnvGuide;
            while (idx < n - 1 && !(cdf[idx] > bound)) idx++;
            guide[b] = idx;
        }
    };
    std::vector<uint32_t> marginalGuide(kEnvGuide + 1), rowGuide((size_t)H * (kEnvGuide + 1));
    make_guide(marginal.data(), H, marginalGuide.data());
    for (uint32_t y = 0; y < H; y++) make_guide(&row[(size_t)y * W], W, &rowGuide[(size_t)y * (kEnvGuide + 1)]);
    NX_SYNC_ALL(c);
    NX_ALLOC(c->envMarginalGuide, marginalGuide.size() * 4);
    NX_ALLOC(c->envRowGuide, rowGuide.size() * 4);
    NX_HIP(hipMemcpy(c->envMarginalGuide.p, marginalGuide.data(), marginalGuide.size() * 4, hipMemcpyHostToDevice));
    NX_HIP(hipMemcpy(c->envRowGuide.p, rowGuide.data(), rowGuide.size() * 4, hipMemcpyHostToDevice));
    c->h.envMarginalGuide = c->envMarginalGuide.as<uint32_t>();
    c->h.envRowGuide = c->envRowGuide.as<uint32_t>();
    NX_ALLOC(c->envMarginalCdf, marginal.size() * 4);
    NX_ALLOC(c->envRowCdf, row.size() * 4);
    NX_ALLOC(c->envDensity, density.size() * 4);
    NX_HIP(hipMemcpy(c->envMarginalCdf.p, marginal.data(), marginal.size() * 4, hipMemcpyHostToDevice));
    NX_HIP(hipMemcpy(c->envRowCdf.p, row.data(), row.size() * 4, hipMemcpyHostToDevice));
    NX_HIP(hipMemcpy(c->envDensity.p, density.data(), density.size() * 4, hipMemcpyHostToDevice));
    c->h.envSampling = 1;
    c->h.envMarginalCdf = c->envMarginalCdf.as<float>();
    c->h.envRowCdf = c->envRowCdf.as<float>();
    c->h.envDensity = c->envDensity.as<float>();
    c->stateDirty = true;
    return NXHIP_OK;
}

int nxhip_set_env_sampling(nxhip_ctx* c, int enable)
try {
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    if (enable && !c->hdrMap.texels.p) return fail_invalid("nxhip_set_env_sampling: upload the environment map first (nxhip_upload_texture kind 2)");
    c->envSampling = enable != 0;
    return build_env_tables(c);
} catch (const std::exception& e) {
    set_error(std::string("nxhip_set_env_sampling: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_upload_texture(nxhip_ctx* c, int kind, const uint8_t* rgba8, uint32_t width, uint32_t height, int32_t* texId)
try {
    NX_CHECK_CTX(c);
    if (!rgba8 || width == 0 || height == 0 || kind < 0 || kind > 2) return fail_invalid("nxhip_upload_texture: bad arguments");
    NX_HIP(hipSetDevice(c->device));
    TextureHost t;
    t.width = width;
    t.height = height;
    NX_ALLOC(t.texels, (size_t)width * height * 4);
    NX_HIP(hipMemcpy(t.texels.p, rgba8, (size_t)width * height * 4, hipMemcpyHostToDevice));
    int32_t id = 0;
    if (kind == 0) { c->diffuseMaps.push_back(std::move(t)); id = (int32_t)c->diffuseMaps.size() - 1; }
    else if (kind == 1) { c->emissiveMaps.push_back(std::move(t)); id = (int32_t)c->emissiveMaps.size() - 1; }
    else {
        NX_SYNC_ALL(c);
        c->hdrMap = std::move(t);
        c->hostHdr.assign(rgba8, rgba8 + (size_t)width * height * 4);
    }
    if (texId) *texId = id;
    const int rc = refresh_texture_tables(c);
    if (rc != NXHIP_OK || kind != 2) return rc;
    return build_env_tables(c);  // a new map under an enabled sampler gets new tables
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_upload_texture: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_clear_textures(nxhip_ctx* c)
try {
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    c->diffuseMaps.clear();
    c->emissiveMaps.clear();
    c->hdrMap = TextureHost();
    c->hostHdr.clear();
    c->envSampling = false;
    const int rc = build_env_tables(c);
    if (rc != NXHIP_OK) return rc;
    return refresh_texture_tables(c);
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_clear_textures: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_set_camera(nxhip_ctx* c, const nx_camera* camera)
{
    NX_CHECK_CTX(c);
    if (!camera) return fail_invalid("nxhip_set_camera: null camera");
    if (camera->resolution[0] != c->width || camera->resolution[1] != c->height)
        return fail_invalid("nxhip_set_camera: camera resolution differs from the context viewport (call nxhip_resize first)");
    // (the reference's Render loop hands the camera and the settings over every frame: an unchanged one must not cost the
    //  state upload, which waits for every pass in flight)
    if (std::memcmp(&c->h.camera, camera, sizeof *camera) == 0) return NXHIP_OK;
    c->h.camera = *camera;
    c->stateDirty = true;
    return NXHIP_OK;
}

int nxhip_set_render_settings(nxhip_ctx* c, const nx_render_settings* s)
{
    NX_CHECK_CTX(c);
    if (!s) return fail_invalid("nxhip_set_render_settings: null settings");
    if (s->pathLength < 1 || s->pathLength > NX_PATH_MAX_LENGTH - 2) return fail_invalid("nxhip_set_render_settings: pathLength must be in [1, 98]");
    if (std::memcmp(&c->h.settings, s, sizeof *s) == 0) return NXHIP_OK;
    if (s->pathLength != c->h.settings.pathLength) invalidate_graph(c);
    c->h.settings = *s;
    c->stateDirty = true;
    return NXHIP_OK;
}

int nxhip_set_modes(nxhip_ctx* c, int rngMode, int compactMode, int conductorMode)
{
    NX_CHECK_CTX(c);
    if (rngMode < 0 || rngMode > 1 || compactMode < 0 || compactMode > 1 || conductorMode < 0 || conductorMode > 1)
        return fail_invalid("nxhip_set_modes: unknown mode");
    if (rngMode == c->h.rngMode && compactMode == c->h.compactMode && conductorMode == c->h.conductorMode) return NXHIP_OK;
    if (compactMode != c->h.compactMode || conductorMode != c->h.conductorMode) invalidate_graph(c);
    c->h.rngMode = rngMode;
    c->h.compactMode = compactMode;
    c->h.conductorMode = conductorMode;
    c->stateDirty = true;
    return NXHIP_OK;
}

int nxhip_set_pixel_map(nxhip_ctx* c, const uint32_t* pixelMap, uint32_t localCount)
{
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    const uint32_t full = c->width * c->height;
    c->pixelOrder = NXHIP_ORDER_ROWS;  // (a caller's own map, or none: nxhip_set_pixel_order sets it again behind its own call)
    // the queues are re-allocated first (all or nothing): a failure leaves the previous pixel set, map and queues in place
    if (!pixelMap) {
        const int rc = alloc_paths(c, full);
        if (rc != NXHIP_OK) return rc;
        c->pixelMap.release();
        c->h.pixelMap = nullptr;
        c->stateDirty = true;
        return set_frame_number_device(c, 0);
    }
    if (localCount == 0 || localCount > full) return fail_invalid("nxhip_set_pixel_map: localCount out of range");
    for (uint32_t i = 0; i < localCount; i++)
        if (pixelMap[i] >= full) return fail_invalid("nxhip_set_pixel_map: pixel index out of range");
    DevBuf freshMap;
    NX_ALLOC(freshMap, (size_t)localCount * 4);
    NX_HIP(hipMemcpy(freshMap.p, pixelMap, (size_t)localCount * 4, hipMemcpyHostToDevice));
    const int rc = alloc_paths(c, localCount);
    if (rc != NXHIP_OK) return rc;
    c->pixelMap = std::move(freshMap);
    c->h.pixelMap = c->pixelMap.as<uint32_t>();
    c->stateDirty = true;
    return set_frame_number_device(c, 0);
}

int nxhip_set_pixel_order(nxhip_ctx* c, int order)
try {
    NX_CHECK_CTX(c);
    if (order != NXHIP_ORDER_ROWS && order != NXHIP_ORDER_TILES) return fail_invalid("nxhip_set_pixel_order: order must be NXHIP_ORDER_ROWS or NXHIP_ORDER_TILES");
    if (order == NXHIP_ORDER_ROWS) {
        const int rc = nxhip_set_pixel_map(c, nullptr, 0);
        if (rc == NXHIP_OK) c->pixelOrder = order;
        return rc;
    }
    std::vector<uint32_t> map((size_t)c->width * c->height);
    uint32_t n = 0;
    int rc = nxhip_tile_pixel_map(c->width, c->height, 1, 0, 1, 1, map.data(), &n);
    if (rc != NXHIP_OK) return rc;
    rc = nxhip_set_pixel_map(c, map.data(), n);
    if (rc == NXHIP_OK) c->pixelOrder = order;
    return rc;
} catch (const std::exception& e) {
    set_error(std::string("nxhip_set_pixel_order: ") + e.what());
    return NXHIP_ERR_INVALID;
}

// ---- rendering --------------------------------------------------------------------------------------

int nxhip_reset_frame_number(nxhip_ctx* c)
{
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    return set_frame_number_device(c, 0);
}

int nxhip_set_frame_number(nxhip_ctx* c, uint32_t f)
{
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    return set_frame_number_device(c, f);
}

uint32_t nxhip_frame_number(nxhip_ctx* c) { return c ? c->frameNumber : 0u; }

static int check_scene_ready(nxhip_ctx* c)
{
    if (!c->h.tlasNodes || c->h.instanceCount == 0) return fail_invalid("no TLAS has been set");
    if (!c->h.materials) return fail_invalid("no materials have been set");
    // Indices that cross tables are followed by the shading kernels without a bounds test (as in the reference): a bad one
    // is a wild device read, so they are checked here, on the host copies, before anything is launched.
    for (const nx_bvh_instance& inst : c->hostInstances)
        if (inst.materialId < 0 || (size_t)inst.materialId >= c->hostMaterials.size()) return fail_invalid("an instance refers to a material id that has not been set");
    for (const nx_material& m : c->hostMaterials) {
        if (m.diffuseMapId < -1 || (m.diffuseMapId >= 0 && (size_t)m.diffuseMapId >= c->diffuseMaps.size())) return fail_invalid("a material refers to a diffuse map that has not been uploaded");
        if (m.emissiveMapId < -1 || (m.emissiveMapId >= 0 && (size_t)m.emissiveMapId >= c->emissiveMaps.size())) return fail_invalid("a material refers to an emissive map that has not been uploaded");
    }
    for (const nx_light& l : c->hostLights)
        if (l.type == NX_LIGHT_MESH && l.mesh.meshId >= c->hostInstances.size()) return fail_invalid("a mesh light refers to an instance that does not exist");
    return NXHIP_OK;
}

}  // extern "C"

namespace {

struct Launch {
    const void* fn;
    dim3 grid, block;
    int klass;
    // argument storage (pointers into this struct are handed to HIP)
    const DeviceState* s;
    int bounce;
    int type;  // nargs 3: (S, bounce, type)
    void* ptr;  // nargs 30: (S, ptr, count)
    int after;  // -1: depends on the previous level; k: on launch k of ITS OWN level only (a chain inside the level)
    const float4* src;
    uint32_t count, slices, sliceStride, firstFrame;
    const uint32_t* dstMap;
    int nargs;  // 1: (S), 2: (S, bounce), 3: (S, bounce, type), 7: accumulate
};

Launch make_launch(const void* fn, int grid, int block, int klass, const DeviceState* s, int bounce = -1)
{
    Launch l{};
    l.fn = fn;
    l.grid = dim3((unsigned)grid);
    l.block = dim3((unsigned)block);
    l.klass = klass;
    l.s = s;
    l.bounce = bounce;
    l.nargs = bounce >= 0 ? 2 : 1;
    l.after = -1;
    return l;
}

// Every pass starts with begin_frame_kernel, launched outside the graph because its argument (the pass size) may change from
// pass to pass.
int launch_begin_frame(nxhip_ctx* c, PassSlot* q, uint32_t frames, uint32_t frameLast)
{
    DeviceState* S = q->dState.as<DeviceState>();
    // the pass's number among the passes of this slot tags the tile status words of the ordered compaction (nx_device.h
    // kScanEpochLimit); when it wraps, the words of the previous round of numbers are cleared, in stream order
    if (++q->scanEpoch >= kScanEpochLimit) {
        if (q->scanStatus.p) NX_HIP(hipMemsetAsync(q->scanStatus.p, 0, q->scanStatus.bytes, q->stream));
        q->scanEpoch = 1;
    }
    const uint32_t epoch = q->scanEpoch;
    void* args[4] = {(void*)&S, (void*)&frames, (void*)&frameLast, (void*)&epoch};
    NX_HIP(hipLaunchKernel(begin_frame_kernel_ptr(), dim3(1), dim3(kWideBlockThreads), args, 0, q->stream));
    (void)c;
    return NXHIP_OK;
}

// Small passes: the four material kernels of a bounce as ONE graph branch.  Parallel branches are spread over hardware
// queues by the runtime, and with many queues configured (passes in flight need them) the cross-queue hand-offs cost more
// than the overlap of four short kernels gives: one frame per pass 288 -> 375 Msamples/s at 24 queues, and 6 passes in flight
// then fit the queues (584 against 537 with 4).  Large passes keep the four branches (20 frames per pass: 1 560 vs 1 505).
// "Small" = up to 4 frames' worth of paths at 1080p.
// The size of a pass in 1080p frames: what the small-pass rules below were measured in.  A rank of a tile split renders a
// fraction of the image, so its 20-frame pass is a small one (8 ranks: 2.5 frames' worth of paths).
double pass_size_in_frames(const nxhip_ctx* c) { return (double)c->localCount * (double)c->framesPerPass / (1920.0 * 1080.0); }

// (round 3: with only the material kernels of types in use in the graph and the slot counters spread over the queue regions, one
//  branch is also the faster form for large passes — 64 frames per pass, 4 in flight: 1 988 against 1 974 Msamples/s — so it is
//  what every pass uses; NX_SHADE_PARALLEL=1 brings the parallel branches back for experiments)
bool serial_shade(const nxhip_ctx* c) { return c->parallelShade != 1; }

// Passes in flight right now: kernel timing and the counting variant measure one pass at a time, and a caller-bound
// radiance buffer exists once.
uint32_t effective_slots(const nxhip_ctx* c)
{
    if (c->timingEnabled || c->statsEnabled || c->radianceBoundCapacity != 0) return 1u;
    return std::min<uint32_t>(c->passesInFlight, (uint32_t)c->extra.size());
}

// Workgroups of a persistent trace launch.  The grid that fills the chip (6 workgroups per CU) is right for ONE large pass.
// With several passes in flight every slot's closest-hit and any-hit launches would each claim the whole chip and the short
// logic / shade kernels of the other slots queue behind their drains (one frame per pass, 6 in flight: 580 Msamples/s with
// full grids, 912 with one workgroup per CU and launch; the logic kernel alone took 9x its solo time).  Throughput saturates
// at 2-3 waves per SIMD, so the slots share the CU: about 6 / R workgroups per CU each, half as many again for large passes.
// A single small pass also drains faster on a half-size grid (1 frame per pass 368 -> 419, 8 frames 1 019 -> 1 065).
int trace_blocks(const nxhip_ctx* c, int fullGrid)
{
    if (c->traceGridForced) return fullGrid;
    const int maxPerCU = std::max(1, fullGrid / std::max(1, c->numCUs));
    const int R = (int)std::max(1u, effective_slots(c));
    const bool small = pass_size_in_frames(c) <= 8.0;
    int perCU;
    // (round 5, with the thin level taking the drains: 4 per CU for a small single pass — 1 / 2 / 4 / 8 frames per pass 648 / 914 / 1 347 /
    //  1 740 Msamples/s against 609 / 896 / 1 304 / 1 719 with 3 and 643 / 906 / 1 314 / 1 724 with 5; a rank of 8's share 4.49 against 4.60 ms)
    if (R == 1) perCU = small ? std::min(4, maxPerCU) : maxPerCU;
    else perCU = ((small ? 2 : 3) * maxPerCU + 2 * R - 1) / (2 * R);
    return std::min(std::max(perCU, 1), maxPerCU) * c->numCUs;
}

// The bounce from which the tail kernel (nx_wavefront.hip) runs the rest of every path, 0 = never.  It needs random numbers
// that do not depend on queue slots and has no ordered-compaction or instrumented form.
// Automatic choice (sweeps of rounds 2 and 3, configs[1], Msamples/s without / with the tail kernel):
//   passes in flight: bounce 5 for passes of up to 4 frames' worth of paths (by then a pass carries ~2 % of its primary rays; one
//     frame per pass, 6 in flight: 969 -> 1 123; an earlier start loses: bounce 4, 4 frames: -9 %), bounce 3 for up to 2.5 frames'
//     worth in at most 3 slots (one frame, 3 in flight: 832 -> 889);
//   one pass at a time: one frame's worth from bounce 3 (521 -> 599), up to 4 frames' worth from bounce 4 (2: 793 -> 875,
//     3: 952 -> 999; 2.5 — a rank of 8 — 4.70 ms from 4 or 5, 4.93 without), up to 10 from bounce 5 (5 — a rank of 4 — 7.45 ms from 5,
//     7.57 from 4, 7.69 without; 8 frames: 1 299 -> 1 355) — the share of a rank of 4 or 8 in the
//     driver's 20-frame job is such a pass (rank of 4: 8.65 -> 8.2 ms).
// Larger passes lose (12 frames +-1 %, 20 frames -2 % from bounce 6 and -3 % from 5, 64 frames -7 %: a wave of the tail kernel keeps
// 64 lanes for as long as its longest path, and there the level-by-level launches are already amortised).
int tail_bounce(const nxhip_ctx* c)
{
    int bounce = c->tailBounce;
    if (bounce < 0) {
        const double frames = pass_size_in_frames(c);
        const unsigned slots = effective_slots(c);
        if (slots > 1u) bounce = frames > 4.0 ? 0 : (frames <= 2.5 && slots <= 3u) ? 3 : 5;
        // (one pass at a time, round 5: the thin level takes the drains out of the trace launches, and the tail kernel — whose waves keep
        //  64 lanes for as long as their longest path, outlier rays included — only still pays for the smallest passes: one frame 595 ->
        //  620 Msamples/s with it, four frames 1 335 -> 1 315, a rank of 8 the same median with a 6.3 ms pass in seven instead of none)
        else if (c->thinWaves) bounce = frames <= 1.5 ? 3 : 0;
        else bounce = frames <= 1.5 ? 3 : frames <= 4.0 ? 4 : frames <= 10.0 ? 5 : 0;
        if (bounce > (int)c->h.settings.pathLength) bounce = 0;
    }
    if (bounce < 2 || bounce > (int)c->h.settings.pathLength) return 0;
    if (c->h.rngMode != NX_RNG_PIXEL_KEYED || c->h.compactMode != NX_COMPACT_FAST || c->statsEnabled || c->timingEnabled) return 0;
    return bounce;
}

// What else shapes a pass graph besides the launch geometry: the pipeline, whether a miss can contribute (SCAN pipeline: the miss
// kernel is in the graph only for a scene with an environment map or a background that is not exactly black — PathTracer.cu:
// 152-164 adds throughput x background, and +0 changes nothing), and the logic kernel's variant (one item per thread under an
// environment map).  Part of a graph instance's key, so a change of any of them picks or builds the matching instance.
constexpr int kFlavorScan = 1, kFlavorMissKernel = 2, kFlavorEnvMap = 4, kFlavorEntry = 8, kFlavorThin = 16;
int pass_flavor(const nxhip_ctx* c)
{
    int f = 0;
    if (scan_pipeline(c)) f |= kFlavorScan;
    if (c->hdrMap.texels.p) f |= kFlavorEnvMap;
    const nx_render_settings& s = c->h.settings;
    bool black = true;
    for (int k = 0; k < 3; k++) {
        const float v = s.backgroundColor[k] * s.backgroundIntensity;  // (the device's own product: sample_background)
        uint32_t bits;
        std::memcpy(&bits, &v, 4);
        black = black && bits == 0u;
    }
    if (c->hdrMap.texels.p || !black) f |= kFlavorMissKernel;
    if (c->entryPoints) f |= kFlavorEntry;  // (the slot's table exists before its graph is asked for: ensure_entry_table)
    // The thin kernel (nx_trace.hip) pays when ONE pass runs at a time: the lanes a dry wave leaves idle are then idle SIMD time, and
    // a level ends with its slowest ray (driver command: mean of five repetitions 19.9 -> 19.0 ms, 512 frames in 64-frame passes one at a
    // time +2.9 %).  With several passes in flight the other passes' waves fill those lanes anyway and the hand-over is extra work
    // (four in flight: -2.4 %, configs[4] -1.1 %): off.
    // Small passes one at a time keep it as well: the median of single passes does not show it (a rank of 8's 2.5 frames' worth: 4.57 ->
    // 4.6-4.7 ms, an extra launch per level), but one pass in seven holds an outlier ray (6.2 ms instead of 4.6) and sequences of small
    // passes are what a viewer or a rank of a tile split renders: four frames per pass 1 223 -> 1 335 Msamples/s, one frame 602 -> 620.
    if (c->thinWaves && !c->statsEnabled && (c->passesInFlight <= 1u || c->thinInFlight)) f |= kFlavorThin;  // (the caller's setting, not effective_slots(): a timing replay of a run with passes in flight keeps that run's kernels)
    return f;
}

// The per-frame kernel sequence, in dependency "levels": launches of one level may run concurrently, a level
// starts after the previous one has finished.  Reference DAG: Renderer/PathTracer.cpp:114-124, :259-278.
std::vector<std::vector<Launch>> frame_levels(nxhip_ctx* c, PassSlot* q)
{
    const DeviceState* S = q->dState.as<DeviceState>();
    const bool ordered = c->h.compactMode == NX_COMPACT_ORDERED;
    const bool stats = c->statsEnabled;
    const int wide = c->wideBlocks;
    const int wideThreads = kWideBlockThreads;
    std::vector<std::vector<Launch>> levels;
    levels.push_back({make_launch(generate_kernel_ptr(), wide, wideThreads, NXHIP_K_GENERATE, S)});
    const bool entry = (pass_flavor(c) & kFlavorEntry) != 0;
    if (entry) {  // beside the generate kernel: the entry states of the primary rays' runs (nx_entry.hip), read by the launch below
        // (table and count come from the slot's DeviceState: a graph node holds no pointer that a re-allocation could leave dangling)
        levels.back().push_back(make_launch(entry_state_kernel_ptr(), (int)((q->entryRuns + 63u) / 64u), 64, NXHIP_K_GENERATE, S));
    }
    const int traceBlocks = trace_blocks(c, c->traceBlocks), shadowBlocks = trace_blocks(c, c->shadowBlocks);
    // (the dry waves of a pass's trace launches may hand their last long rays to the thin kernel: nx_trace.hip)
    const int thinFlag = (pass_flavor(c) & kFlavorThin) ? kTraceThinFlag : 0;
    levels.push_back({make_launch(trace_kernel_ptr(false, stats), traceBlocks, kTraceBlockThreads, NXHIP_K_TRACE, S, (entry ? kTraceEntryFlag : 0) | thinFlag)});
    // behind the trace launch(es) of a level: the rays their dry waves handed over, a wave each (thin_kernel)
    // — each trace launch of the level gets its own, chained to it alone, so that the closest-hit rays' searches run beside whatever
    // the any-hit launch still has to do (it is the longer one of the early levels) and the other way round in the late ones
    int thinBlocks = 3 * c->numCUs;  // (48 KiB of LDS: three workgroups per CU)
    if (const char* on = std::getenv("NX_TUNING_KNOBS"); on && std::atoi(on) == 1)
        if (const char* e = std::getenv("NX_THIN_BLOCKS_PER_CU")) { const int n = std::atoi(e); if (n >= 1 && n <= 16) thinBlocks = n * c->numCUs; }  // sweeps only
    auto thin_level = [&](int bounceArg) {
        if (!thinFlag) return;
        if (c->thinJoint) {  // (NX_THIN_JOINT, measurement only: one launch for both lists behind the whole level)
            levels.push_back({make_launch(thin_kernel_ptr(), thinBlocks, kTraceBlockThreads, NXHIP_K_THIN, S, bounceArg)});
            return;
        }
        std::vector<Launch>& level = levels.back();
        const int n = (int)level.size();  // 1: the primary level (closest-hit only); 2: closest-hit, any-hit
        for (int k = 0; k < n; k++) {
            Launch t = make_launch(thin_kernel_ptr(), thinBlocks, kTraceBlockThreads, NXHIP_K_THIN, S, bounceArg | (k == 0 ? kThinClosestOnly : kThinAnyOnly));
            t.after = k;
            level.push_back(t);
        }
    };
    const int pathLength = c->h.settings.pathLength;
    // (grids of the producer kernels stay multiples of the queue regions: harmless, and what a round-robin tile-to-region mapping
    //  would need)
    auto whole_regions = [](int g) { return (g + kQueueShards - 1) / kQueueShards * kQueueShards; };
    // (ordered compaction: the same grids — tiles are handed out by ticket and their slots found by look-back, nx_wavefront.hip)
    const int og = whole_regions(c->shadeBlocksPerCU * c->numCUs), ob = ordered ? kShadeBlockOrderedThreads : kShadeBlockThreads;
    const int lg = whole_regions(c->logicBlocksPerCU * c->numCUs), lb = kLogicBlockThreads;
    const int tailFrom = tail_bounce(c);
    auto in_use = [&](int type) { return (c->materialTypeMask >> type) & 1u; };
    if (scan_pipeline(c)) {
        // SCAN pipeline (nx_wavefront.hip): the closest-hit launch leaves the logic step's decision in its hit records, the material
        // kernels find their items there.  Per bounce: [miss kernel, only when a miss can contribute] -> the material kernels of the
        // types in use, one after the other -> trace || shadow trace.  One launch less per bounce than the reference's DAG.
        levels[1][0].bounce |= kTraceScanFlag;
        thin_level(0 | kTraceScanFlag);
        const bool misses = pass_flavor(c) & kFlavorMissKernel;
        for (int bounce = 1; bounce <= pathLength; bounce++) {
            if (bounce == tailFrom) {  // the rest of the pass in one launch
                levels.push_back({make_launch(tail_kernel_ptr(), c->tailBlocks, kTraceBlockThreads, NXHIP_K_SHADE, S, bounce | kTraceScanFlag)});
                break;
            }
            if (misses && c->scanSeparate) levels.push_back({make_launch(miss_scan_kernel_ptr(), lg, kWideBlockThreads, NXHIP_K_LOGIC, S, bounce)});
            std::vector<Launch> shade;
            // the material kernels of the types in use: ONE launch for all of them (shade_scan_kernel), or — NX_SCAN_SEPARATE, measurement
            // only — one per type in the reference's graph order
            int mask = (int)(c->materialTypeMask & 0xfu);
            if (c->h.conductorMode != NX_CONDUCTOR_EXTENDED) mask &= ~(1 << NX_MAT_CONDUCTOR);
            if (mask == 0) mask = 1 << NX_MAT_DIFFUSE;  // (a level cannot be empty)
            if (misses && !c->scanSeparate) mask |= 1 << kScanMiss;  // the misses: a fifth type of the one launch
            auto scan_launch = [&](int m) {
                Launch l = make_launch(shade_scan_kernel_ptr(), og, kShadeBlockThreads, NXHIP_K_SHADE, S, bounce);
                l.type = m;
                l.nargs = 3;
                return l;
            };
            if (c->scanSeparate) {
                for (int type : {NX_MAT_DIFFUSE, NX_MAT_PLASTIC, NX_MAT_DIELECTRIC, NX_MAT_CONDUCTOR})
                    if ((mask >> type) & 1) shade.push_back(scan_launch(1 << type));
            } else shade.push_back(scan_launch(mask));
            if (in_use(NX_MAT_CONDUCTOR) && c->h.conductorMode != NX_CONDUCTOR_EXTENDED) {  // (counted, not shaded: count_scan_kernel)
                Launch l = make_launch(count_scan_kernel_ptr(), lg, kWideBlockThreads, NXHIP_K_LOGIC, S, bounce);
                l.type = NX_MAT_CONDUCTOR;
                l.nargs = 3;
                shade.push_back(l);
            }
            for (auto& l : shade) levels.push_back({l});
            levels.push_back({make_launch(trace_kernel_ptr(false, stats), traceBlocks, kTraceBlockThreads, NXHIP_K_TRACE, S, bounce | kTraceScanFlag | thinFlag),
                              make_launch(trace_kernel_ptr(true, stats), shadowBlocks, kTraceBlockThreads, NXHIP_K_SHADOW, S, bounce | thinFlag)});
            thin_level(bounce | kTraceScanFlag);
        }
        return levels;
    }
    thin_level(0);
    for (int bounce = 1; bounce <= pathLength; bounce++) {
        if (bounce == tailFrom) {  // the rest of the pass in one launch
            levels.push_back({make_launch(tail_kernel_ptr(), c->tailBlocks, kTraceBlockThreads, NXHIP_K_SHADE, S, bounce)});
            break;
        }
        levels.push_back({make_launch(logic_kernel_ptr(ordered, c->hdrMap.texels.p ? 1 : 2), lg, lb, NXHIP_K_LOGIC, S, bounce)});
        // graph insertion order of the reference: Diffuse, Plastic, Dielectric, Conductor (PathTracer.cpp:116-120)
        // (only the types some material of the scene has: a queue no material feeds stays empty)
        std::vector<Launch> shade;
        if (in_use(NX_MAT_DIFFUSE)) shade.push_back(make_launch(shade_kernel_ptr(NX_MAT_DIFFUSE, ordered), og, ob, NXHIP_K_SHADE, S, bounce));
        if (in_use(NX_MAT_PLASTIC)) shade.push_back(make_launch(shade_kernel_ptr(NX_MAT_PLASTIC, ordered), og, ob, NXHIP_K_SHADE, S, bounce));
        if (in_use(NX_MAT_DIELECTRIC)) shade.push_back(make_launch(shade_kernel_ptr(NX_MAT_DIELECTRIC, ordered), og, ob, NXHIP_K_SHADE, S, bounce));
        if (in_use(NX_MAT_CONDUCTOR) && c->h.conductorMode == NX_CONDUCTOR_EXTENDED) shade.push_back(make_launch(shade_kernel_ptr(NX_MAT_CONDUCTOR, ordered), og, ob, NXHIP_K_SHADE, S, bounce));
        if (shade.empty()) shade.push_back(make_launch(shade_kernel_ptr(NX_MAT_DIFFUSE, ordered), og, ob, NXHIP_K_SHADE, S, bounce));  // (a level cannot be empty)
        // serial slot order needs the kernels one after the other; so do several passes in flight, which would otherwise ask
        // for four hardware queues per slot (the material kernels of one bounce serialise on the CUs anyway: each grid fills them)
        if (ordered || serial_shade(c)) for (auto& l : shade) levels.push_back({l});
        else levels.push_back(shade);
        levels.push_back({make_launch(trace_kernel_ptr(false, stats), traceBlocks, kTraceBlockThreads, NXHIP_K_TRACE, S, bounce | thinFlag),
                          make_launch(trace_kernel_ptr(true, stats), shadowBlocks, kTraceBlockThreads, NXHIP_K_SHADOW, S, bounce | thinFlag)});
        thin_level(bounce);
    }
    return levels;
}

void fill_args(Launch& l, void** args)
{
    args[0] = (void*)&l.s;
    if (l.nargs == 2 || l.nargs == 3) args[1] = (void*)&l.bounce;
    if (l.nargs == 3) args[2] = (void*)&l.type;
    if (l.nargs == 30) { args[1] = (void*)&l.ptr; args[2] = (void*)&l.count; }
    if (l.nargs == 7) {
        args[1] = (void*)&l.src; args[2] = (void*)&l.count; args[3] = (void*)&l.slices; args[4] = (void*)&l.sliceStride;
        args[5] = (void*)&l.firstFrame; args[6] = (void*)&l.dstMap;
    }
}

int launch_now(nxhip_ctx* c, Launch& l)
{
    void* args[8];
    fill_args(l, args);
    KernelTimer* t = nullptr;
    if (c->timingEnabled) {
        c->timerPool.emplace_back();
        t = &c->timerPool.back();
        NX_HIP(hipEventCreate(&t->start));
        NX_HIP(hipEventCreate(&t->stop));
        NX_HIP(hipEventRecord(t->start, c->stream));
    }
    NX_HIP(hipLaunchKernel(l.fn, l.grid, l.block, args, 0, c->stream));
    // Any launch outside a pass graph that can set the slot's error word (traversal / ordered-scan stall guards) makes the pinned copy
    // the last pass left behind stale: nxhip_sync then reads the word itself.  Here, once, for every such launch — the eager timing
    // path, the ray-batch hooks, the thin kernel behind them and whatever comes later (accumulate cannot set it and is issued after
    // every pass: it keeps the copy fresh).
    if (l.klass != NXHIP_K_ACCUMULATE) c->errorFresh = false;
    if (t) {
        NX_HIP(hipEventRecord(t->stop, c->stream));
        c->times.launches[l.klass]++;  // elapsed times are resolved in nxhip_read_kernel_times
    }
    return NXHIP_OK;
}

}  // namespace

// The pass graph of one slot for the shape the context asks for right now, built on first use and kept (a handful of shapes
// exist: small / large pass, passes in flight, tail kernel on or off).  Timing nodes (event records around every kernel) exist
// only in slot 0, and only one such instance at a time: kernel timing runs one pass at a time and the events are the context's.
constexpr size_t kMaxGraphInstances = 8;

static int pass_graph(nxhip_ctx* c, PassSlot* q, hipGraphExec_t* execOut)
{
    const bool serial = serial_shade(c);
    const int blocks = trace_blocks(c, c->traceBlocks), tail = tail_bounce(c), flavor = pass_flavor(c);
    for (auto& g : q->graphs)
        if (g.serialShade == serial && g.traceBlocks == blocks && g.tailBounce == tail && g.flavor == flavor) {
            *execOut = g.exec;
            return NXHIP_OK;
        }
    const bool isMain = q == static_cast<PassSlot*>(c);
    const bool timed = isMain && c->timingMode >= 2;  // event-record nodes around every kernel node; the DAG (and its overlap) is unchanged
    if (timed) invalidate_graph(c);  // the timing events belong to ONE instance
    if (q->graphs.size() >= kMaxGraphInstances) {  // (not reached by the shapes above; a bound all the same)
        (void)hipStreamSynchronize(q->stream);
        PassSlot::GraphInstance& old = q->graphs.front();
        (void)hipGraphExecDestroy(old.exec);
        (void)hipGraphDestroy(old.graph);
        q->graphs.erase(q->graphs.begin());
    }
    PassSlot::GraphInstance inst;
    inst.serialShade = serial;
    inst.traceBlocks = blocks;
    inst.tailBounce = tail;
    inst.flavor = flavor;
    NX_HIP(hipGraphCreate(&inst.graph, 0));
    auto fail = [&](int rc) {
        if (inst.exec) (void)hipGraphExecDestroy(inst.exec);
        if (inst.graph) (void)hipGraphDestroy(inst.graph);
        return rc;
    };
    auto levels = frame_levels(c, q);
    std::vector<hipGraphNode_t> prev;
    for (auto& level : levels) {
        std::vector<hipGraphNode_t> cur;
        std::vector<hipGraphNode_t> done;  // per launch of this level: the node its dependents wait for
        std::vector<bool> hasFollower(level.size(), false);
        for (auto& l : level)
            if (l.after >= 0) hasFollower[(size_t)l.after] = true;
        for (auto& l : level) {
            // what this launch waits for: the previous level, or one launch of its own level (Launch::after)
            std::vector<hipGraphNode_t> deps = l.after >= 0 ? std::vector<hipGraphNode_t>{done[(size_t)l.after]} : prev;
            void* args[8];
            fill_args(l, args);
            hipKernelNodeParams p;
            std::memset(&p, 0, sizeof p);
            p.func = const_cast<void*>(l.fn);
            p.gridDim = l.grid;
            p.blockDim = l.block;
            p.sharedMemBytes = 0;
            p.kernelParams = args;
            p.extra = nullptr;
            hipGraphNode_t node;
            if (timed) {
                c->graphTimers.emplace_back();
                KernelTimer& t = c->graphTimers.back();
                c->graphTimerClass.push_back(l.klass);
                if (!hip_ok(hipEventCreate(&t.start), "hipEventCreate", __FILE__, __LINE__) || !hip_ok(hipEventCreate(&t.stop), "hipEventCreate", __FILE__, __LINE__)) return fail(NXHIP_ERR_HIP);
                hipGraphNode_t before, after;
                if (!hip_ok(hipGraphAddEventRecordNode(&before, inst.graph, deps.empty() ? nullptr : deps.data(), deps.size(), t.start), "hipGraphAddEventRecordNode", __FILE__, __LINE__) ||
                    !hip_ok(hipGraphAddKernelNode(&node, inst.graph, &before, 1, &p), "hipGraphAddKernelNode", __FILE__, __LINE__) ||
                    !hip_ok(hipGraphAddEventRecordNode(&after, inst.graph, &node, 1, t.stop), "hipGraphAddEventRecordNode", __FILE__, __LINE__)) return fail(NXHIP_ERR_HIP);
                done.push_back(after);
            } else {
                if (!hip_ok(hipGraphAddKernelNode(&node, inst.graph, deps.empty() ? nullptr : deps.data(), deps.size(), &p), "hipGraphAddKernelNode", __FILE__, __LINE__)) return fail(NXHIP_ERR_HIP);
                done.push_back(node);
            }
        }
        for (size_t i = 0; i < level.size(); i++)  // the next level waits for the ends of this level's chains
            if (!hasFollower[i]) cur.push_back(done[i]);
        prev.swap(cur);
    }
    if (!hip_ok(hipGraphInstantiate(&inst.exec, inst.graph, nullptr, nullptr, 0), "hipGraphInstantiate", __FILE__, __LINE__)) return fail(NXHIP_ERR_HIP);
    q->graphs.push_back(inst);
    *execOut = inst.exec;
    return NXHIP_OK;
}

// Where pass number i of the round robin renders.  One pass at a time: the context itself, on its own stream.  R > 1: the R
// extra slots, each on a stream of its own — the context's stream then only carries the accumulates (and whatever the caller
// puts behind them, e.g. the multi-GPU gather), so that no pass ever queues behind the accumulate of its predecessor.
static PassSlot* render_slot(nxhip_ctx* c, uint32_t R, uint32_t i) { return R <= 1 ? static_cast<PassSlot*>(c) : c->extra[i].get(); }

// Entry points on: slot q has a table of one state per run of 64 local pixels.  The graph instances of a slot are keyed by shape,
// not by table size, and entry_state_kernel's grid is the run count: a slot whose run count changes drops its graphs.
static int ensure_entry_table(nxhip_ctx* c, PassSlot* q)
{
    const uint32_t runs = (c->localCount + 63u) / 64u;
    if (q->entryRuns == runs && q->entryTable.p) return NXHIP_OK;
    NX_SYNC_ALL(c);
    if (q->entryRuns != runs) invalidate_graph(c);
    NX_ALLOC(q->entryTable, (size_t)std::max(1u, runs) * sizeof(EntryState));
    NX_HIP(hipMemset(q->entryTable.p, 0, (size_t)std::max(1u, runs) * sizeof(EntryState)));  // (steps 0: "start at the root")
    q->entryRuns = runs;
    c->stateDirty = true;
    return NXHIP_OK;
}

static int ensure_slot_events(PassSlot* q)
{
    if (!q->done) NX_HIP(hipEventCreateWithFlags(&q->done, hipEventDisableTiming));
    if (!q->accumulated) NX_HIP(hipEventCreateWithFlags(&q->accumulated, hipEventDisableTiming));
    return NXHIP_OK;
}

extern "C" {

int nxhip_render_frame(nxhip_ctx* c)
try {
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    int rc = check_scene_ready(c);
    if (rc != NXHIP_OK) return rc;
    if (c->shadeInstDirty) {
        rc = refresh_shade_inst(c);
        if (rc != NXHIP_OK) return rc;
    }
    // the slot this pass renders in: round robin over the passes in flight (one slot: everything on the context's stream,
    // exactly the single-pass behaviour)
    const uint32_t R = std::max(1u, effective_slots(c));
    if (c->nextSlot >= R) c->nextSlot = 0;
    PassSlot* q = render_slot(c, R, c->nextSlot);
    c->nextSlot = (c->nextSlot + 1) % R;
    if (c->entryPoints) {  // the slot's entry-state table follows the pixel set (one state per run of 64 local pixels)
        rc = ensure_entry_table(c, q);
        if (rc != NXHIP_OK) return rc;
    }
    rc = upload_state(c);
    if (rc != NXHIP_OK) return rc;
    if (!slot_queues_ready(c, q)) {
        rc = ensure_slot_queues(c, q);
        if (rc != NXHIP_OK) return rc;
        rc = upload_state(c);
        if (rc != NXHIP_OK) return rc;
    }
    if (R > 1 && c->pathCapacity != 0 && !c->awaitingAccumulate) {
        // the passes render in the extra slots: slot 0's queue set (as large as any of theirs) would sit idle
        NX_SYNC_ALL(c);
        release_slot_queues(c, c);
        rc = upload_state(c);
        if (rc != NXHIP_OK) return rc;
    }
    const uint32_t frames = c->framesPerPass, frameLast = c->frameNumber + frames;
    if (R > 1) {
        rc = ensure_slot_events(q);
        if (rc != NXHIP_OK) return rc;
        // the slot's previous pass must have been consumed by its accumulate before its radiance is overwritten
        if (q->accumulateRecorded) {
            NX_HIP(hipStreamWaitEvent(q->stream, q->accumulated, 0));
            q->accumulateRecorded = false;
        }
    }
    if (q->awaitingAccumulate) {  // rendered again without an accumulate in between: the older pass is dropped
        c->pending.erase(std::remove(c->pending.begin(), c->pending.end(), q), c->pending.end());
        q->awaitingAccumulate = false;
    }
    rc = launch_begin_frame(c, q, frames, frameLast);
    if (rc != NXHIP_OK) return rc;
    if (c->timingEnabled && c->timingMode == 1) {
        // eager path: one event pair per launch, launches strictly in level order on one stream
        auto levels = frame_levels(c, q);
        for (auto& level : levels)
            for (auto& l : level) {
                const size_t before = c->timerPool.size();
                rc = launch_now(c, l);
                if (rc != NXHIP_OK) return rc;
                if (c->timerPool.size() > before) c->timerClass.push_back(l.klass);
            }
    } else {
        // (one instance per shape: a pass that crosses the small-pass threshold, or a change of the passes in flight, replays
        //  the instance of its shape — built once)
        hipGraphExec_t exec = nullptr;
        rc = pass_graph(c, q, &exec);
        if (rc != NXHIP_OK) return rc;
        NX_HIP(hipGraphLaunch(exec, q->stream));
        if (c->timingMode == 3) c->graphTimersPending = true;  // read at nxhip_read_kernel_times: the last replay only
        if (c->timingMode == 2) {
            // the graph's events are re-recorded by the next replay: read them now (timing mode is not the fast path)
            NX_SYNC_ALL(c);
            for (size_t i = 0; i < c->graphTimers.size(); i++) {
                float ms = 0.0f;
                NX_HIP(hipEventElapsedTime(&ms, c->graphTimers[i].start, c->graphTimers[i].stop));
                c->times.ms[c->graphTimerClass[i]] += ms;
                c->times.launches[c->graphTimerClass[i]]++;
            }
        }
    }
    // the error word of this pass travels behind it (nxhip_sync then needs no read of its own: a blocking device read costs about 0.1 ms,
    // once per frame for a viewer that synchronises every frame)
    if (!q->hostError) NX_HIP(hipHostMalloc((void**)&q->hostError, sizeof(uint32_t), hipHostMallocDefault));
    NX_HIP(hipMemcpyAsync(q->hostError, &q->frame.as<FrameState>()->errorWord, sizeof(uint32_t), hipMemcpyDeviceToHost, q->stream));
    q->errorFresh = true;
    if (R > 1) NX_HIP(hipEventRecord(q->done, q->stream));
    q->frames = frames;
    q->frameLast = frameLast;
    q->awaitingAccumulate = true;
    c->pending.push_back(q);
    c->lastRendered = q;
    c->frameNumber = frameLast;
    return NXHIP_OK;
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_render_frame: ") + e.what());
    return NXHIP_ERR_INVALID;
}

// AccumulateKernel on the context's (main) stream, reading `stateSlot`'s device state (its radiance, pass size, frame number).
static int launch_accumulate(nxhip_ctx* c, PassSlot* stateSlot, const float4* src, uint32_t count, uint32_t slices, uint32_t sliceStride, uint32_t firstFrame,
                             const uint32_t* dstMap)
{
    Launch l = make_launch(accumulate_kernel_ptr(), c->wideBlocks, kWideBlockThreads, NXHIP_K_ACCUMULATE, stateSlot->dState.as<DeviceState>());
    l.nargs = 7;
    l.src = src;
    l.count = count;
    l.slices = slices;
    l.sliceStride = sliceStride;
    l.firstFrame = firstFrame;
    l.dstMap = dstMap;
    const size_t before = c->timerPool.size();
    const int rc = launch_now(c, l);
    if (rc == NXHIP_OK && c->timerPool.size() > before) c->timerClass.push_back(l.klass);
    return rc;
}

int nxhip_accumulate(nxhip_ctx* c)
{
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    int rc = upload_state(c);
    if (rc != NXHIP_OK) return rc;
    if (c->pending.empty()) {
        // nothing rendered since the last accumulate: the reference's AccumulateKernel would fold the same radiance in again
        PassSlot* q = c->lastRendered ? c->lastRendered : static_cast<PassSlot*>(c);
        return launch_accumulate(c, q, nullptr, c->localCount, 0u, 0u, 0u, nullptr);
    }
    // every rendered pass, oldest first (the running mean is order dependent); all on the context's stream, each after
    // its pass has finished on the slot's stream
    for (PassSlot* q : c->pending) {
        const bool other = q->stream != c->stream;
        if (other) NX_HIP(hipStreamWaitEvent(c->stream, q->done, 0));
        rc = launch_accumulate(c, q, nullptr, c->localCount, 0u, 0u, 0u, nullptr);
        if (rc != NXHIP_OK) return rc;
        if (other) {
            NX_HIP(hipEventRecord(q->accumulated, c->stream));
            q->accumulateRecorded = true;
        }
        q->awaitingAccumulate = false;
    }
    c->pending.clear();
    return NXHIP_OK;
}

// Tail kernel: from `bounce` on (2 .. pathLength; 0 = off; NXHIP_TAIL_AUTO = the default) every path is finished by one launch
// instead of a graph level per kernel and bounce.  Takes effect with pixel-keyed random numbers and the workgroup-aggregated
// compaction only.
int nxhip_set_tail_bounce(nxhip_ctx* c, uint32_t bounce)
{
    NX_CHECK_CTX(c);
    if (bounce != NXHIP_TAIL_AUTO && (bounce == 1u || bounce > (uint32_t)NX_PATH_MAX_LENGTH))
        return fail_invalid("nxhip_set_tail_bounce: bounce must be 0 (off), NXHIP_TAIL_AUTO or in [2, NX_PATH_MAX_LENGTH]");
    c->tailBounce = bounce == NXHIP_TAIL_AUTO ? -1 : (int)bounce;  // the graphs notice at their next use (render_frame compares tail_bounce())
    return NXHIP_OK;
}

int nxhip_set_entry_points(nxhip_ctx* c, int on)
{
    NX_CHECK_CTX(c);
    if ((on != 0) == c->entryPoints) return NXHIP_OK;
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    c->entryPoints = on != 0;
    if (!c->entryPoints)
        for (uint32_t k = 0; k < slot_count(c); k++) {
            slot_at(c, k)->entryTable.release();
            slot_at(c, k)->entryRuns = 0;
        }
    // (nxhip_render_frame allocates the rendering slot's table for the current pixel set; compose_view publishes it in the slot's
    //  DeviceState — the pass graphs read it from there, so no graph instance can hold a freed table: ADVICE r5)
    c->stateDirty = true;
    return NXHIP_OK;
}

int nxhip_debug_set_thin(nxhip_ctx* c, uint32_t lanes, uint32_t iters, int inHooks)
{
    NX_DEBUG_HOOK("nxhip_debug_set_thin");  // (first: a release library refuses whatever it is handed)
    NX_CHECK_CTX(c);
    if (lanes == 0 || lanes > 64u) return fail_invalid("nxhip_debug_set_thin: lanes must be in [1, 64]");
    NX_SYNC_ALL(c);
    c->h.thinLanes = lanes | ((inHooks & 2) ? 0x80000000u : 0u);  // (bit 31: hand over at any time, see nx_trace.hip)
    c->h.thinIters = iters;
    c->thinInHooks = (inHooks & 1) != 0;
    c->stateDirty = true;
    return NXHIP_OK;
}

int nxhip_debug_set_requeue(nxhip_ctx* c, int on)
{
    NX_DEBUG_HOOK("nxhip_debug_set_requeue");  // (first: a release library refuses whatever it is handed)
    NX_CHECK_CTX(c);
    NX_SYNC_ALL(c);
    c->h.debugRequeue = on ? 1u : 0u;
    c->stateDirty = true;
    return NXHIP_OK;
}

int nxhip_debug_set_thin_pool(nxhip_ctx* c, uint32_t slots)
{
    NX_DEBUG_HOOK("nxhip_debug_set_thin_pool");  // (first: a release library refuses whatever it is handed)
    NX_CHECK_CTX(c);
    NX_SYNC_ALL(c);
    c->h.thinPoolLimit = slots;
    c->stateDirty = true;
    return NXHIP_OK;
}

int nxhip_debug_thin_counts(nxhip_ctx* c, int32_t counts[2])
{
    NX_DEBUG_HOOK("nxhip_debug_thin_counts");  // (first: a release library refuses whatever it is handed)
    NX_CHECK_CTX(c);
    if (!counts) return fail_invalid("nxhip_debug_thin_counts: null destination");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    for (int k = 0; k < 2; k++)
        NX_HIP(hipMemcpy(&counts[k], &c->counters.as<Counters>()->thinCount[k][kHookBounceSlot], 4, hipMemcpyDeviceToHost));
    return NXHIP_OK;
}

int nxhip_debug_thin_counts_of_pass(nxhip_ctx* c, uint32_t bounce, int32_t counts[2])
{
    NX_DEBUG_HOOK("nxhip_debug_thin_counts_of_pass");  // (first: a release library refuses whatever it is handed)
    NX_CHECK_CTX(c);
    if (!counts || bounce >= (uint32_t)kMaxBounceSlots) return fail_invalid("nxhip_debug_thin_counts_of_pass: null destination or bounce out of range");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    const PassSlot* q = c->lastRendered ? c->lastRendered : static_cast<const PassSlot*>(c);
    for (int k = 0; k < 2; k++)
        NX_HIP(hipMemcpy(&counts[k], &q->counters.as<Counters>()->thinCount[k][bounce], 4, hipMemcpyDeviceToHost));
    return NXHIP_OK;
}

int nxhip_read_entry_states(nxhip_ctx* c, void* out, uint32_t capacityRuns, uint32_t* count)
{
    NX_CHECK_CTX(c);
    if (!count) return fail_invalid("nxhip_read_entry_states: null count");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    // the table of the slot that rendered last (every slot's table holds the same states: camera, pixel set and scene are the context's)
    const PassSlot* q = c->lastRendered ? c->lastRendered : static_cast<const PassSlot*>(c);
    const bool have = c->entryPoints && q->entryTable.p;
    *count = have ? q->entryRuns : 0u;
    if (out && have) NX_HIP(hipMemcpy(out, q->entryTable.p, (size_t)std::min(capacityRuns, q->entryRuns) * sizeof(EntryState), hipMemcpyDeviceToHost));
    return NXHIP_OK;
}

// Passes in flight (default 1).  R > 1: consecutive nxhip_render_frame calls go to R slots round robin, each with its own
// queues, stream and graph instance; nxhip_accumulate folds the finished passes into the one accumulation in order.
int nxhip_set_passes_in_flight(nxhip_ctx* c, uint32_t passes)
try {
    NX_CHECK_CTX(c);
    if (passes == 0 || passes > 8) return fail_invalid("nxhip_set_passes_in_flight: passes must be in [1, 8]");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    while (passes > 1 && c->extra.size() < passes) {
        std::unique_ptr<PassSlot> q(new PassSlot());
        NX_HIP(hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking));
        q->ownsStream = true;
        if (!q->dState.alloc(sizeof(DeviceState)) || !q->counters.alloc(sizeof(Counters)) || !q->frame.alloc(sizeof(FrameState))) {
            (void)hipStreamDestroy(q->stream);
            return NXHIP_ERR_HIP;
        }
        NX_HIP(hipMemset(q->counters.p, 0, sizeof(Counters)));
        FrameState fs{0u, -1, -1, 0u, 0u, {0u, 0u, 0u}};
        NX_HIP(hipMemcpy(q->frame.p, &fs, sizeof fs, hipMemcpyHostToDevice));
        c->extra.push_back(std::move(q));  // (its queues are allocated when it first renders: ensure_slot_queues)
    }
    // slots beyond the passes in flight give their queues back (a context that once ran 8 passes in flight does not keep 8 queue sets)
    for (size_t k = passes > 1 ? passes : 0; k < c->extra.size(); k++)
        if (c->extra[k]->pathCapacity) release_slot_queues(c, c->extra[k].get());
    for (uint32_t k = 0; k < slot_count(c); k++) {
        const int rc = ensure_slot_events(slot_at(c, k));
        if (rc != NXHIP_OK) return rc;
    }
    c->passesInFlight = passes;
    c->nextSlot = 0;
    c->stateDirty = true;
    return NXHIP_OK;
} catch (const std::exception& e) {
    set_error(std::string("nxhip_set_passes_in_flight: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_set_frames_per_pass(nxhip_ctx* c, uint32_t frames)
{
    NX_CHECK_CTX(c);
    if (frames == 0 || frames > 1024) return fail_invalid("nxhip_set_frames_per_pass: frames must be in [1, 1024]");
    if ((uint64_t)c->localCount * frames > 0x7fffffffull) return fail_invalid("nxhip_set_frames_per_pass: more than 2^31 paths");
    const size_t n = (size_t)std::max<uint32_t>(c->localCount, 1u) * frames;
    if (c->radianceBoundCapacity != 0 && n > c->radianceBoundCapacity)
        return fail_invalid("nxhip_set_frames_per_pass: the bound radiance buffer is too small for this many frames; rebind first");
    if (n > c->queueCapacity) {
        // grow only: a later, smaller pass (e.g. the remainder of a frame budget) reuses the buffers.  The frame counter
        // and the accumulation are left alone: a pass size is a scheduling choice, not a new image.
        NX_HIP(hipSetDevice(c->device));
        NX_SYNC_ALL(c);
        float4* const boundPtr = c->h.radiance;
        const size_t boundCap = c->radianceBoundCapacity;
        const int rc = alloc_queues(c, n);
        if (rc != NXHIP_OK) return rc;
        if (boundCap != 0) {  // an external radiance binding survives the growth
            c->h.radiance = boundPtr;
            c->radianceBoundCapacity = boundCap;
        }
    }
    // Within the capacity nothing is uploaded and nothing waits: the next pass's begin_frame_kernel carries the size as a
    // kernel argument and publishes it on the device, in stream order.
    c->framesPerPass = frames;
    c->pathCount = c->localCount * frames;
    c->h.framesPerPass = frames;
    c->h.pathCount = c->pathCount;
    return NXHIP_OK;
}

int nxhip_release_queues(nxhip_ctx* c)
{
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    if (c->radianceBoundCapacity != 0) return fail_invalid("nxhip_release_queues: a caller-owned radiance buffer is bound (nxhip_bind_radiance(ctx, NULL, 0) first)");
    c->pending.clear();
    for (uint32_t k = 0; k < slot_count(c); k++) {
        PassSlot* q = slot_at(c, k);
        q->awaitingAccumulate = false;
        if (q->pathCapacity) release_slot_queues(c, q);
    }
    return NXHIP_OK;
}

int nxhip_bind_radiance(nxhip_ctx* c, void* radianceDevice, uint32_t capacity)
{
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    if (radianceDevice) {
        if (capacity < c->pathCount) return fail_invalid("nxhip_bind_radiance: buffer smaller than localCount * framesPerPass");
        c->h.radiance = static_cast<float4*>(radianceDevice);
        c->radianceBoundCapacity = capacity;
    } else {
        c->h.radiance = c->radiance.as<float4>();  // (null while slot 0's queues are released: they come back with its next use)
        c->radianceBoundCapacity = 0;
    }
    c->stateDirty = true;
    return NXHIP_OK;
}

int nxhip_read_full_accumulation(nxhip_ctx* c, float* dst)
{
    NX_CHECK_CTX(c);
    return read_float4_as_float3(c, c->accumulation.p, c->width * c->height, dst);
}

int nxhip_read_full_rgba8(nxhip_ctx* c, uint32_t* dst)
{
    NX_CHECK_CTX(c);
    if (!dst) return fail_invalid("null destination");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    NX_HIP(hipMemcpy(dst, c->rgba8.p, (size_t)c->width * c->height * 4, hipMemcpyDeviceToHost));
    return NXHIP_OK;
}

int nxhip_accumulate_external(nxhip_ctx* c, const void* src, uint32_t count, uint32_t slices, uint32_t sliceStride, uint32_t firstFrame,
                              const void* srcPixelMapDevice)
{
    NX_CHECK_CTX(c);
    if (!src || count == 0 || count > c->width * c->height || firstFrame == 0 || slices == 0 || sliceStride < count)
        return fail_invalid("nxhip_accumulate_external: bad arguments");
    NX_HIP(hipSetDevice(c->device));
    const int rc = upload_state(c);
    if (rc != NXHIP_OK) return rc;
    return launch_accumulate(c, c, static_cast<const float4*>(src), count, slices, sliceStride, firstFrame, static_cast<const uint32_t*>(srcPixelMapDevice));
}

int nxhip_compose_tiles(nxhip_ctx* c, const void* srcAccumulation, uint32_t count, const void* srcPixelMapDevice, void* dstAccumulationDevice,
                        void* dstRgba8Device)
{
    NX_CHECK_CTX(c);
    if (!srcAccumulation || !dstAccumulationDevice || count == 0 || count > c->width * c->height) return fail_invalid("nxhip_compose_tiles: bad arguments");
    NX_HIP(hipSetDevice(c->device));
    const float4* src = static_cast<const float4*>(srcAccumulation);
    const uint32_t* map = static_cast<const uint32_t*>(srcPixelMapDevice);
    float4* dstA = static_cast<float4*>(dstAccumulationDevice);
    uint32_t* dstP = static_cast<uint32_t*>(dstRgba8Device);
    void* args[5] = {(void*)&src, (void*)&count, (void*)&map, (void*)&dstA, (void*)&dstP};
    NX_HIP(hipLaunchKernel(compose_kernel_ptr(), dim3(c->wideBlocks), dim3(kWideBlockThreads), args, 0, c->stream));
    return NXHIP_OK;
}

int nxhip_render(nxhip_ctx* c, uint32_t frames)
{
    NX_CHECK_CTX(c);
    const uint32_t S = c->framesPerPass;
    int rc = NXHIP_OK;
    for (uint32_t f = 0; f < frames && rc == NXHIP_OK; f += S) {
        const uint32_t n = std::min(S, frames - f);  // the last pass may be shorter
        if (n != c->framesPerPass) rc = nxhip_set_frames_per_pass(c, n);
        if (rc == NXHIP_OK) rc = nxhip_render_frame(c);
        if (rc == NXHIP_OK) rc = nxhip_accumulate(c);
    }
    if (c->framesPerPass != S) {
        const int rc2 = nxhip_set_frames_per_pass(c, S);
        if (rc == NXHIP_OK) rc = rc2;
    }
    return rc;
}

static int read_float4_as_float3(nxhip_ctx* c, const void* dev, uint32_t count, float* dst)
try {
    if (!dst) return fail_invalid("null destination");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    std::vector<float4> tmp(count);
    NX_HIP(hipMemcpy(tmp.data(), dev, (size_t)count * 16, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < count; i++) {
        dst[3 * (size_t)i + 0] = tmp[i].x;
        dst[3 * (size_t)i + 1] = tmp[i].y;
        dst[3 * (size_t)i + 2] = tmp[i].z;
    }
    return NXHIP_OK;
} catch (const std::exception& e) {
    set_error(std::string("read-back: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_read_radiance(nxhip_ctx* c, float* dst)
{
    NX_CHECK_CTX(c);
    // the pass rendered last (with several passes in flight: the newest one's slot)
    const PassSlot* q = c->lastRendered ? c->lastRendered : static_cast<PassSlot*>(c);
    const void* src = q == static_cast<PassSlot*>(c) ? (const void*)c->h.radiance : (const void*)q->radiance.p;
    return read_float4_as_float3(c, src, c->pathCount, dst);
}

int nxhip_read_accumulation(nxhip_ctx* c, float* dst)
{
    NX_CHECK_CTX(c);
    return read_float4_as_float3(c, c->accumulation.p, c->localCount, dst);
}

int nxhip_write_accumulation(nxhip_ctx* c, const float* src, uint32_t frameNumber)
try {
    NX_CHECK_CTX(c);
    if (!src) return fail_invalid("nxhip_write_accumulation: null source");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    std::vector<float4> tmp(c->localCount);
    for (uint32_t i = 0; i < c->localCount; i++) tmp[i] = make_float4(src[3 * (size_t)i], src[3 * (size_t)i + 1], src[3 * (size_t)i + 2], 0.0f);
    NX_HIP(hipMemcpy(c->accumulation.p, tmp.data(), (size_t)c->localCount * 16, hipMemcpyHostToDevice));
    return set_frame_number_device(c, frameNumber);
} catch (const std::exception& e) {
    set_error(std::string("nxhip_write_accumulation: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_read_rgba8(nxhip_ctx* c, uint32_t* dst)
{
    NX_CHECK_CTX(c);
    if (!dst) return fail_invalid("null destination");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    NX_HIP(hipMemcpy(dst, c->rgba8.p, (size_t)c->localCount * 4, hipMemcpyDeviceToHost));
    return NXHIP_OK;
}

// Slot 0's radiance buffer (or the bound one).  Passes in flight > 1 render in other slots and slot 0's queues are released:
// NULL then — use nxhip_bind_radiance or nxhip_read_radiance.
void* nxhip_radiance_device_ptr(nxhip_ctx* c) { return c ? (void*)c->h.radiance : nullptr; }
void* nxhip_accumulation_device_ptr(nxhip_ctx* c) { return c ? c->accumulation.p : nullptr; }

int nxhip_read_queue_sizes(nxhip_ctx* c, nxhip_queue_sizes* out)
{
    NX_CHECK_CTX(c);
    if (!out) return fail_invalid("null destination");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    Counters h;
    NX_HIP(hipMemcpy(&h, (c->lastRendered ? c->lastRendered : static_cast<PassSlot*>(c))->counters.p, sizeof h, hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof *out);
    for (int k = 0; k < kQueueShards; k++)  // a queue's size = the sum over its regions
        for (int b = 0; b < kMaxBounceSlots; b++) {
            const RegionCounters& r = h.region[k];
            out->traceSize[b] += r.traceSize[b];
            out->traceShadowSize[b] += r.traceShadowSize[b];
            out->diffuseSize[b] += r.materialSize[NX_MAT_DIFFUSE][b];
            out->plasticSize[b] += r.materialSize[NX_MAT_PLASTIC][b];
            out->dielectricSize[b] += r.materialSize[NX_MAT_DIELECTRIC][b];
            out->conductorSize[b] += r.materialSize[NX_MAT_CONDUCTOR][b];
        }
    return NXHIP_OK;
}

int nxhip_set_pixel_query(nxhip_ctx* c, uint32_t x, uint32_t y)
{
    NX_CHECK_CTX(c);
    if (x >= c->width || y >= c->height) return fail_invalid("nxhip_set_pixel_query: pixel outside the viewport");
    NX_HIP(hipSetDevice(c->device));
    const int32_t q[2] = {(int32_t)(c->width * y + x), -1};
    NX_SYNC_ALL(c);
    for (uint32_t k = 0; k < slot_count(c); k++) NX_HIP(hipMemcpy(&slot_at(c, k)->frame.as<FrameState>()->pixelQueryPixel, q, 8, hipMemcpyHostToDevice));
    return NXHIP_OK;
}

int nxhip_get_selected_instance(nxhip_ctx* c, int32_t* instanceIdx)
{
    NX_CHECK_CTX(c);
    if (!instanceIdx) return fail_invalid("null destination");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    NX_HIP(hipMemcpy(instanceIdx, &(c->lastRendered ? c->lastRendered : static_cast<PassSlot*>(c))->frame.as<FrameState>()->pixelQueryInstance, 4, hipMemcpyDeviceToHost));
    return NXHIP_OK;
}

// ---- kernel-level hooks -----------------------------------------------------------------------------

// The ray-batch hooks number their n rays densely and cut them into one contiguous piece per queue region in use (as
// generate_kernel does for the primary rays): ray i lives in slot (i / piece) * cap + i % piece.
struct HookLayout {
    uint32_t shards, cap, piece;
    int32_t sizes[kQueueShards];
};
static HookLayout hook_layout(const nxhip_ctx* c, uint32_t n)
{
    HookLayout l{};
    l.shards = c->view.queueShards;
    l.cap = c->view.queueShardCap;
    l.piece = ((n + l.shards * 64u - 1u) / (l.shards * 64u)) * 64u;
    for (uint32_t k = 0; k < (uint32_t)kQueueShards; k++) l.sizes[k] = k < l.shards ? (int32_t)std::min(l.piece, n - std::min(n, k * l.piece)) : 0;
    return l;
}
// host array (n elements of `elem` bytes, dense numbering) <-> a queue buffer's regions
static int hook_copy(nxhip_ctx* c, const HookLayout& l, void* dev, void* host, size_t elem, bool toDevice)
{
    for (uint32_t k = 0; k < l.shards; k++) {
        if (l.sizes[k] <= 0) continue;
        char* d = static_cast<char*>(dev) + (size_t)k * l.cap * elem;
        char* h = static_cast<char*>(host) + (size_t)k * l.piece * elem;
        if (toDevice) NX_HIP(hipMemcpyAsync(d, h, (size_t)l.sizes[k] * elem, hipMemcpyHostToDevice, c->stream));
        else NX_HIP(hipMemcpyAsync(h, d, (size_t)l.sizes[k] * elem, hipMemcpyDeviceToHost, c->stream));
    }
    return NXHIP_OK;
}

static int run_trace_chunk(nxhip_ctx* c, bool anyHit, uint32_t n)
{
    // region sizes + zeroed fetch heads for the reserved bounce slot, computed on the device from n (hook_layout's rule): no copy
    // from a stack-local of this function is left in flight when it returns
    {
        DeviceState* S = c->dState.as<DeviceState>();
        const int any = anyHit ? 1 : 0, slot = kHookBounceSlot;
        void* args[4] = {(void*)&S, (void*)&n, (void*)&any, (void*)&slot};
        NX_HIP(hipLaunchKernel(hook_sizes_kernel_ptr(), dim3(1), dim3(64), args, 0, c->stream));
    }
    c->errorFresh = false;  // (this launch may set the error word after the last pass's copy of it)
    const bool thin = c->thinInHooks && !c->statsEnabled;
    Launch l = make_launch(trace_kernel_ptr(anyHit, c->statsEnabled), anyHit ? c->shadowBlocks : c->traceBlocks, kTraceBlockThreads,
                           anyHit ? NXHIP_K_SHADOW : NXHIP_K_TRACE, c->dState.as<DeviceState>(), kHookBounceSlot | (thin ? kTraceThinFlag : 0));
    const size_t before = c->timerPool.size();
    int rc = launch_now(c, l);
    if (rc == NXHIP_OK && c->timerPool.size() > before) c->timerClass.push_back(l.klass);
    if (rc == NXHIP_OK && thin) {  // (nxhip_debug_set_thin: what the dry waves handed over, a wave each)
        Launch t = make_launch(thin_kernel_ptr(), 3 * c->numCUs, kTraceBlockThreads, NXHIP_K_THIN, c->dState.as<DeviceState>(), kHookBounceSlot);
        const size_t before2 = c->timerPool.size();
        rc = launch_now(c, t);
        if (rc == NXHIP_OK && c->timerPool.size() > before2) c->timerClass.push_back(t.klass);
    }
    return rc;
}

int nxhip_trace_batch(nxhip_ctx* c, const nx_ray* rays, uint32_t count, nx_hit* hits)
try {
    NX_CHECK_CTX(c);
    if (count == 0) return NXHIP_OK;
    if (!rays || !hits) return fail_invalid("nxhip_trace_batch: null buffer");
    NX_HIP(hipSetDevice(c->device));
    if (!c->h.tlasNodes) return fail_invalid("no TLAS has been set");
    int rc = ensure_slot_queues(c, c);
    if (rc != NXHIP_OK) return rc;
    rc = upload_state(c);
    if (rc != NXHIP_OK) return rc;
    const uint32_t cap = c->pathCount;
    std::vector<float4> o(std::min(cap, count)), d(std::min(cap, count)), h(std::min(cap, count));
    std::vector<uint32_t> hi(std::min(cap, count));
    for (uint32_t first = 0; first < count; first += cap) {
        const uint32_t n = std::min(cap, count - first);
        for (uint32_t i = 0; i < n; i++) {
            const nx_ray& r = rays[first + i];
            float idx;
            std::memcpy(&idx, &i, 4);
            o[i] = make_float4(r.origin[0], r.origin[1], r.origin[2], 0.0f);
            d[i] = make_float4(r.direction[0], r.direction[1], r.direction[2], idx);
        }
        const HookLayout l = hook_layout(c, n);
        if ((rc = hook_copy(c, l, c->trRayO.p, o.data(), 16, true)) != NXHIP_OK) return rc;
        if ((rc = hook_copy(c, l, c->trRayD.p, d.data(), 16, true)) != NXHIP_OK) return rc;
        rc = run_trace_chunk(c, false, n);
        if (rc != NXHIP_OK) return rc;
        if ((rc = hook_copy(c, l, c->trHit.p, h.data(), 16, false)) != NXHIP_OK) return rc;
        if ((rc = hook_copy(c, l, c->trHitInst.p, hi.data(), 4, false)) != NXHIP_OK) return rc;
        NX_SYNC_ALL(c);
        for (uint32_t i = 0; i < n; i++) {
            nx_hit& out = hits[first + i];
            out.hitDistance = h[i].x;
            out.u = h[i].y;
            out.v = h[i].z;
            std::memcpy(&out.triIdx, &h[i].w, 4);
            out.instanceIdx = hi[i];
        }
    }
    return NXHIP_OK;
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_trace_batch: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_trace_shadow_batch(nxhip_ctx* c, const nx_ray* rays, const float* tmax, uint32_t count, uint8_t* occluded)
try {
    NX_CHECK_CTX(c);
    if (count == 0) return NXHIP_OK;
    if (!rays || !tmax || !occluded) return fail_invalid("nxhip_trace_shadow_batch: null buffer");
    NX_HIP(hipSetDevice(c->device));
    if (!c->h.tlasNodes) return fail_invalid("no TLAS has been set");
    int rc = ensure_slot_queues(c, c);
    if (rc != NXHIP_OK) return rc;
    rc = upload_state(c);
    if (rc != NXHIP_OK) return rc;
    const uint32_t cap = c->pathCount;
    const uint32_t m = std::min(cap, count);
    std::vector<float4> o(m), d(m), rad(m, make_float4(1.0f, 0.0f, 0.0f, 0.0f)), res(m);
    for (uint32_t first = 0; first < count; first += cap) {
        const uint32_t n = std::min(cap, count - first);
        for (uint32_t i = 0; i < n; i++) {
            const nx_ray& r = rays[first + i];
            float idx;
            std::memcpy(&idx, &i, 4);
            o[i] = make_float4(r.origin[0], r.origin[1], r.origin[2], tmax[first + i]);
            d[i] = make_float4(r.direction[0], r.direction[1], r.direction[2], idx);
        }
        // the kernel's tail adds the request's radiance to the path's pixel when unoccluded: radiance 1 into a zeroed buffer
        const HookLayout l = hook_layout(c, n);
        if ((rc = hook_copy(c, l, c->shRayO.p, o.data(), 16, true)) != NXHIP_OK) return rc;
        if ((rc = hook_copy(c, l, c->shRayD.p, d.data(), 16, true)) != NXHIP_OK) return rc;
        if ((rc = hook_copy(c, l, c->shRadiance.p, rad.data(), 16, true)) != NXHIP_OK) return rc;
        NX_HIP(hipMemsetAsync(c->h.radiance, 0, (size_t)n * 16, c->stream));  // the buffer the kernel adds into (own or bound)
        rc = run_trace_chunk(c, true, n);
        if (rc != NXHIP_OK) return rc;
        NX_HIP(hipMemcpyAsync(res.data(), c->h.radiance, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
        NX_SYNC_ALL(c);
        for (uint32_t i = 0; i < n; i++) occluded[first + i] = res[i].x == 1.0f ? 0 : 1;
    }
    return NXHIP_OK;
} catch (const std::exception& e) {  // nothing may unwind through the C boundary
    set_error(std::string("nxhip_trace_shadow_batch: ") + e.what());
    return NXHIP_ERR_INVALID;
}

int nxhip_enable_trace_stats(nxhip_ctx* c, int enable)
{
    NX_CHECK_CTX(c);
    if ((enable != 0) != c->statsEnabled) invalidate_graph(c);
    c->statsEnabled = enable != 0;
    return NXHIP_OK;
}

int nxhip_read_trace_stats(nxhip_ctx* c, nxhip_trace_stats* closest, nxhip_trace_stats* shadow, int reset)
{
    NX_CHECK_CTX(c);
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    TraceStatsDev h[2];
    NX_HIP(hipMemcpy(h, c->traceStats.p, sizeof h, hipMemcpyDeviceToHost));
    static_assert(sizeof(nxhip_trace_stats) == sizeof(TraceStatsDev), "stats layouts must match");
    if (closest) std::memcpy(closest, &h[0], sizeof h[0]);
    if (shadow) std::memcpy(shadow, &h[1], sizeof h[1]);
    if (reset) NX_HIP(hipMemset(c->traceStats.p, 0, sizeof h));
    return NXHIP_OK;
}

static int bsdf_hook(nxhip_ctx* c, const nx_material* material, const nx_bsdf_query* queries, uint32_t count, nx_bsdf_result* results, int sample)
{
    NX_CHECK_CTX(c);
    if (!material || (!queries && count) || (!results && count)) return fail_invalid("nxhip_bsdf_*_batch: null buffer");
    if (material->type < NX_MAT_DIFFUSE || material->type > NX_MAT_CONDUCTOR) return fail_invalid("nxhip_bsdf_*_batch: unknown material type");
    if (count == 0) return NXHIP_OK;
    NX_HIP(hipSetDevice(c->device));
    DevBuf dMat, dQ, dR;
    NX_ALLOC(dMat, sizeof(nx_material));
    NX_ALLOC(dQ, (size_t)count * sizeof(nx_bsdf_query));
    NX_ALLOC(dR, (size_t)count * sizeof(nx_bsdf_result));
    NX_HIP(hipMemcpy(dMat.p, material, sizeof(nx_material), hipMemcpyHostToDevice));
    NX_HIP(hipMemcpy(dQ.p, queries, (size_t)count * sizeof(nx_bsdf_query), hipMemcpyHostToDevice));
    const nx_material* pm = dMat.as<nx_material>();
    const nx_bsdf_query* pq = dQ.as<nx_bsdf_query>();
    nx_bsdf_result* pr = dR.as<nx_bsdf_result>();
    void* args[5] = {(void*)&pm, (void*)&pq, (void*)&count, (void*)&sample, (void*)&pr};
    NX_HIP(hipLaunchKernel(bsdf_hook_kernel_ptr(), dim3(c->wideBlocks), dim3(kWideBlockThreads), args, 0, c->stream));
    NX_SYNC_ALL(c);
    NX_HIP(hipMemcpy(results, dR.p, (size_t)count * sizeof(nx_bsdf_result), hipMemcpyDeviceToHost));
    return NXHIP_OK;
}

int nxhip_bsdf_sample_batch(nxhip_ctx* c, const nx_material* material, const nx_bsdf_query* queries, uint32_t count, nx_bsdf_result* results)
{
    return bsdf_hook(c, material, queries, count, results, 1);
}

int nxhip_bsdf_eval_batch(nxhip_ctx* c, const nx_material* material, const nx_bsdf_query* queries, uint32_t count, nx_bsdf_result* results)
{
    return bsdf_hook(c, material, queries, count, results, 0);
}

int nxhip_tex2d_batch(nxhip_ctx* c, int kind, int textureId, const float* uv, uint32_t count, float* rgba)
{
    NX_CHECK_CTX(c);
    if ((!uv || !rgba) && count) return fail_invalid("nxhip_tex2d_batch: null buffer");
    if (kind < 0 || kind > 2) return fail_invalid("nxhip_tex2d_batch: kind must be 0, 1 or 2");
    if (kind == 0 && (textureId < 0 || (size_t)textureId >= c->diffuseMaps.size())) return fail_invalid("nxhip_tex2d_batch: no such diffuse map");
    if (kind == 1 && (textureId < 0 || (size_t)textureId >= c->emissiveMaps.size())) return fail_invalid("nxhip_tex2d_batch: no such emissive map");
    if (kind == 2 && !c->hdrMap.texels.p) return fail_invalid("nxhip_tex2d_batch: no environment map has been uploaded");
    if (count == 0) return NXHIP_OK;
    NX_HIP(hipSetDevice(c->device));
    const TextureHost& th = kind == 0 ? c->diffuseMaps[textureId] : kind == 1 ? c->emissiveMaps[textureId] : c->hdrMap;
    TextureDev t{th.texels.as<uint32_t>(), th.width, th.height};
    DevBuf dUv, dOut;
    NX_ALLOC(dUv, (size_t)count * 8);
    NX_ALLOC(dOut, (size_t)count * 16);
    NX_HIP(hipMemcpy(dUv.p, uv, (size_t)count * 8, hipMemcpyHostToDevice));
    const float* lut = c->srgbLut.as<float>();
    const float* pu = dUv.as<float>();
    float4* po = dOut.as<float4>();
    void* args[5] = {(void*)&t, (void*)&lut, (void*)&pu, (void*)&count, (void*)&po};
    NX_HIP(hipLaunchKernel(tex2d_hook_kernel_ptr(), dim3(c->wideBlocks), dim3(kWideBlockThreads), args, 0, c->stream));
    NX_SYNC_ALL(c);
    NX_HIP(hipMemcpy(rgba, dOut.p, (size_t)count * 16, hipMemcpyDeviceToHost));
    return NXHIP_OK;
}

int nxhip_fmath_batch(nxhip_ctx* c, int op, const double* a, const double* b, uint32_t count, double* out)
{
    NX_CHECK_CTX(c);
    if (op < 0 || op >= NXF_OP_COUNT) return fail_invalid("nxhip_fmath_batch: op must be one of NXF_OP_*");
    if ((!a || !out) && count) return fail_invalid("nxhip_fmath_batch: null buffer");
    if (count == 0) return NXHIP_OK;
    NX_HIP(hipSetDevice(c->device));
    DevBuf dA, dB, dOut;
    NX_ALLOC(dA, (size_t)count * 8);
    NX_ALLOC(dOut, (size_t)count * 8);
    NX_HIP(hipMemcpy(dA.p, a, (size_t)count * 8, hipMemcpyHostToDevice));
    if (b) {
        NX_ALLOC(dB, (size_t)count * 8);
        NX_HIP(hipMemcpy(dB.p, b, (size_t)count * 8, hipMemcpyHostToDevice));
    }
    const double* pa = dA.as<double>();
    const double* pb = b ? dB.as<double>() : nullptr;
    double* po = dOut.as<double>();
    void* args[5] = {(void*)&op, (void*)&pa, (void*)&pb, (void*)&count, (void*)&po};
    NX_HIP(hipLaunchKernel(fmath_hook_kernel_ptr(), dim3(c->wideBlocks), dim3(kWideBlockThreads), args, 0, c->stream));
    NX_SYNC_ALL(c);
    NX_HIP(hipMemcpy(out, dOut.p, (size_t)count * 8, hipMemcpyDeviceToHost));
    return NXHIP_OK;
}

int nxhip_enable_kernel_timing(nxhip_ctx* c, int enable)
{
    NX_CHECK_CTX(c);
    if (enable < 0 || enable > 3) return fail_invalid("nxhip_enable_kernel_timing: mode must be 0, 1, 2 or 3");
    if (enable != c->timingMode) invalidate_graph(c);
    c->timingMode = enable;
    c->graphTimersPending = false;
    c->timingEnabled = enable != 0;
    return NXHIP_OK;
}

int nxhip_read_graph_timeline(nxhip_ctx* c, int32_t* klass, float* startMs, float* durationMs, uint32_t capacity, uint32_t* count)
{
    NX_CHECK_CTX(c);
    if (!count) return fail_invalid("nxhip_read_graph_timeline: null count");
    if (capacity && (!klass || !startMs || !durationMs)) return fail_invalid("nxhip_read_graph_timeline: null destination");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    *count = (uint32_t)c->graphTimers.size();
    if (c->graphTimers.empty() || !c->lastRendered) return NXHIP_OK;
    for (size_t i = 0; i < c->graphTimers.size() && i < capacity; i++) {
        float s = 0.0f, d = 0.0f;
        NX_HIP(hipEventElapsedTime(&s, c->graphTimers[0].start, c->graphTimers[i].start));
        NX_HIP(hipEventElapsedTime(&d, c->graphTimers[i].start, c->graphTimers[i].stop));
        klass[i] = c->graphTimerClass[i];
        startMs[i] = s;
        durationMs[i] = d;
    }
    return NXHIP_OK;
}

int nxhip_read_kernel_times(nxhip_ctx* c, nxhip_kernel_times* out, int reset)
{
    NX_CHECK_CTX(c);
    if (!out) return fail_invalid("null destination");
    NX_HIP(hipSetDevice(c->device));
    NX_SYNC_ALL(c);
    for (size_t i = 0; i < c->timerPool.size(); i++) {
        float ms = 0.0f;
        NX_HIP(hipEventElapsedTime(&ms, c->timerPool[i].start, c->timerPool[i].stop));
        c->times.ms[c->timerClass[i]] += ms;
        (void)hipEventDestroy(c->timerPool[i].start);
        (void)hipEventDestroy(c->timerPool[i].stop);
    }
    c->timerPool.clear();
    c->timerClass.clear();
    if (c->graphTimersPending) {  // mode 3: the events hold the last replay of a back-to-back series
        for (size_t i = 0; i < c->graphTimers.size(); i++) {
            float ms = 0.0f;
            NX_HIP(hipEventElapsedTime(&ms, c->graphTimers[i].start, c->graphTimers[i].stop));
            c->times.ms[c->graphTimerClass[i]] += ms;
            c->times.launches[c->graphTimerClass[i]]++;
        }
        c->graphTimersPending = false;
    }
    *out = c->times;
    if (reset) std::memset(&c->times, 0, sizeof c->times);
    return NXHIP_OK;
}

}  // extern "C"
