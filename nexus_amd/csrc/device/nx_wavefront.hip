// nx_wavefront.hip — the wide wavefront kernels: begin-frame, generate, logic, shade (+NEE), accumulate.
//
// What they compute is the reference's GenerateKernel, LogicKernel, Shade<BSDF> + NextEventEstimation<BSDF> and
// AccumulateKernel (/root/reference/Nexus/src/Cuda/PathTracer/PathTracer.cu:85-122, 136-210, 213-458, 480-496).
// How they run is MI355X-first:
//   * persistent grid-stride launches (<= 2048 workgroups of 256) instead of N/64+1 blocks that mostly exit:
//     late bounces with few live paths cost a launch, not tens of thousands of empty workgroups;
//   * float4-packed SoA queues: every lane moves 16 B per access, fully coalesced;
//   * queue slots are handed out once per workgroup (ballots + popcount ranks + one atomic) — NX_COMPACT_FAST — or by a
//     grid-wide ordered scan (tiles by ticket, decoupled look-back) — NX_COMPACT_ORDERED, which reproduces the
//     reference's serial slot order exactly (ascending thread index, material kernels in graph order) on all CUs and
//     is what the parity tests use; the shading math is the same code in both modes;
//   * a path's shadow-ray and continuation-ray payloads are built in registers and written after the slot
//     allocation, outside divergent control flow;
//   * the frame number lives on the device and is advanced by begin_frame_kernel, so a frame is one
//     hipGraph replay with no host-side writes in between.
#define NX_KERNEL_TU 1
#include "nx_bsdf.h"
#include "nx_device.h"
#include "nx_math.h"
#include "nx_queue.h"
#include "nx_texture.h"
#include "nx_traverse.h"

namespace nxd {

constexpr int kWideBlock = 256;     // generate / accumulate
#ifndef NX_SHADE_BLOCK
#define NX_SHADE_BLOCK 256
#endif
#ifndef NX_LOGIC_BLOCK
#define NX_LOGIC_BLOCK 1024
#endif
constexpr int kLogicBlock = NX_LOGIC_BLOCK;   // one slot atomic per tile of kLogicItems x this many items
// Items per logic thread.  Two: a tile of 2 048 items pays ONE round of slot allocation (barriers, the returning atomic per queue
// or, ordered, the ticket and four look-backs) where 1 024-item tiles paid two — logic kernel -4.5 % racing, ordered mode +2.7 % on
// the driver command — at 64 VGPRs with 20 spilled (two workgroups per CU as before; with a 4-wave register budget and no spills
// only one workgroup fits a CU: +18 % on the kernel).
#ifndef NX_LOGIC_ITEMS
#define NX_LOGIC_ITEMS 2
#endif
constexpr int kLogicItems = NX_LOGIC_ITEMS;
constexpr int kShadeBlock = NX_SHADE_BLOCK;

// ------------------------------------------------------------------------------------------------------
// slot allocation: K queues at once, one round of workgroup-wide cooperation per tile.
//   FAST    : per-wave ballots -> per-workgroup totals in LDS -> ONE atomicAdd per workgroup and counter (a single
//             global counter sustains only ~88 returning atomics per microsecond on MI355X, so per-wave atomics
//             serialise a 2M-item pass; per-workgroup ones do not).  Slot order = order of the atomics: racy, like the
//             reference's per-thread atomicAdd (PathTracer.cu:143, 326).
//   ORDERED : the reference's SERIAL slot order — ascending item index within a kernel, the material kernels of a bounce
//             continuing each other's queues in graph order (PathTracer.cpp:114-124) — on the whole chip.  Tiles are handed
//             out by ticket (so a tile's predecessors are always held by running workgroups), and a tile's base slot is the
//             sum of its predecessors' counts, found by decoupled look-back: every tile publishes its count, then reads
//             the status words of the tiles before it 64 at a time until it meets one that already knows its inclusive
//             prefix, and publishes its own.  Tile 0 starts from the counter word (what the previous kernel of the chain
//             left there), the last tile stores the new total.  Round 1-3 did this in ONE 1024-thread workgroup.
// Must be called by every thread of the workgroup (uniform control flow).

constexpr int kMaxWavesPerBlock = 1024 / kWave;

NXD unsigned long long scan_word(uint32_t serial, uint32_t state, int value) { return ((unsigned long long)((serial << 2) | state) << 32) | (unsigned long long)(uint32_t)value; }
NXD int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Decoupled look-back of the ordered compaction, ONE text for both allocators below (one wave per queue calls it): the slots taken
// by all tiles before tile t of this launch (and by the kernels before it in the chain: tile 0 starts from the counter word), and
// this tile's count published for its successors.  `st`: the queue's status words ([tile * kScanWords]), `counter`: the queue's
// counter word, `serial`: the launch's tag (words of other launches read as "not there yet").
NXD int scan_look_back(NX_G unsigned long long* const st, const int* const counter, NX_G FrameState* const frame, const uint32_t serial, const int t, const int total)
{
    const int lane = threadIdx.x & (kWave - 1);
    if (t == 0) {
        const int base = *counter;
        if (lane == 0) __hip_atomic_store(&st[0], scan_word(serial, kScanPrefix, base + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return base;
    }
    if (lane == 0) __hip_atomic_store(&st[(size_t)t * kScanWords], scan_word(serial, kScanAggregate, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int excl = 0, nearest = t - 1;
    uint32_t spins = 0u;
    for (;;) {
        const int j = nearest - lane;  // lane 0 looks at the nearest predecessor not summed yet
        uint32_t tag = (serial << 2) | kScanPrefix;  // (tiles "before tile 0" never decide: tile 0 itself is a prefix)
        int value = 0;
        if (j >= 0) {
            const unsigned long long w = __hip_atomic_load(&st[(size_t)j * kScanWords], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tag = (uint32_t)(w >> 32);
            value = (int)(uint32_t)w;
        }
        const bool ready = (tag >> 2) == serial && (tag & 3u) != 0u;
        const unsigned long long notReady = __ballot(!ready), prefixes = __ballot(ready && (tag & 3u) == kScanPrefix);
        const int firstNot = notReady ? __ffsll((long long)notReady) - 1 : kWave;
        const int firstPrefix = prefixes ? __ffsll((long long)prefixes) - 1 : kWave;
        if (firstPrefix < firstNot) {  // every tile between here and a known prefix has published its count
            excl += wave_sum(lane <= firstPrefix ? value : 0);
            break;
        }
        // counts in front of the first tile that has not published yet are final: take them, then look again from there
        excl += wave_sum(lane < firstNot ? value : 0);
        nearest -= firstNot;
        if (firstNot < kWave) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 20)) {  // (about a second; see kErrScanStalled)
                if (lane == 0) atomicOr(&frame->errorWord, kErrScanStalled);
                break;
            }
        }
    }
    if (lane == 0) __hip_atomic_store(&st[(size_t)t * kScanWords], scan_word(serial, kScanPrefix, excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

template <bool ORDERED, int K>
struct SlotAllocator {
    int* sWave;  // [K][kMaxWavesPerBlock] per-wave counts of the current tile
    int* sBase;  // [K] base slot of the current tile
    int* sTile;  // ORDERED: the ticket of the current tile
    int* counter0;  // the K counter words are counter0 + k * counterStep (an address computation, not an array of pointers: an
    int counterStep;  // array indexed by anything but a constant makes the allocator a private-memory object in LDS or scratch)
    NX_G unsigned long long* status;  // ORDERED: tile status words, [tile][kScanWords]
    NX_G int* ticket;
    NX_G FrameState* frame;
    uint32_t serial;
    int size, lastTile;

    // `kind`: 0 logic, 1 + NX_MAT_* material kernel; `items`: size of the launch's input queue
    NXD void init(const DeviceState* S, int* const first, const int step, const int kind, const int bounce, const int items)
    {
        __shared__ int wave[K * kMaxWavesPerBlock];
        __shared__ int base[K];
        __shared__ int tile;
        sWave = wave;
        sBase = base;
        sTile = &tile;
        counter0 = first;
        counterStep = step;
        size = items;
        lastTile = (items + (int)blockDim.x - 1) / (int)blockDim.x - 1;
        if (ORDERED) {
            status = S->scanStatus;
            ticket = &S->counters->scanTicket[kind][bounce];
            frame = S->frame;
            serial = S->frame->scanEpoch * 1024u + (uint32_t)bounce * 8u + (uint32_t)kind;
        }
    }
    NXD int* counter_of(const int k) const
    {
        return counter0 + k * counterStep;
    }
    // first item of the workgroup's first / next tile (>= size: none left)
    NXD int first_tile() { return ORDERED ? take() : (int)(blockIdx.x * blockDim.x); }
    NXD int next_tile(const int tile) { return ORDERED ? take() : tile + (int)(gridDim.x * blockDim.x); }
    NXD int take()
    {
        __syncthreads();  // the previous tile's readers are done with sTile
        if (threadIdx.x == 0) *sTile = atomicAdd(ticket, 1);
        __syncthreads();
        const int t = *sTile;
        return t > lastTile ? size : t * (int)blockDim.x;
    }

    NXD int look_back(const int k, const int t, const int total) { return scan_look_back(status + k, counter_of(k), frame, serial, t, total); }

    // `tile`: first item of the tile (what first_tile / next_tile returned); `region`: the queue region this tile appends to
    // (uniform over the workgroup; ORDERED keeps one region); counters[] point at region 0's words
    NXD void alloc(const bool (&want)[K], int (&slot)[K], const int tile, const int region = 0)
    {
        const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
        const int nWaves = blockDim.x / kWave;
        unsigned long long mask[K];
        __syncthreads();  // the previous tile's readers are done with sWave / sBase
#pragma unroll
        for (int k = 0; k < K; k++) {
            mask[k] = __ballot(want[k]);
            if (lane == 0) sWave[k * kMaxWavesPerBlock + wave] = __popcll(mask[k]);
        }
        __syncthreads();
        if (ORDERED) {
            if (wave < K) {  // (a workgroup has at least K waves: 256 threads, K <= 4)
                const int total = wave_sum(lane < nWaves ? sWave[wave * kMaxWavesPerBlock + lane] : 0);
                const int t = tile / (int)blockDim.x;
                const int excl = look_back(wave, t, total);
                if (lane == 0) {
                    sBase[wave] = excl;
                    if (t == lastTile) *counter_of(wave) = excl + total;
                }
            }
        } else if (threadIdx.x < K) {
            int total = 0;
            for (int w = 0; w < nWaves; w++) total += sWave[threadIdx.x * kMaxWavesPerBlock + w];
            int* const word = counter_of((int)threadIdx.x) + region * kRegionStride;
            sBase[threadIdx.x] = total ? atomicAdd(word, total) : 0;
#ifdef NX_EXTRA_ATOMICS
            // experiment: are the logic / material kernels bound by the returning atomics on their queue
            // counters?  NX_EXTRA_ATOMICS more of them per tile and counter, adding zero
            for (int x = 0; x < NX_EXTRA_ATOMICS; x++)
                if (total && atomicAdd(word, 0) == -123456789) sBase[threadIdx.x] = 0;  // (returning, result unused)
#endif
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) {
            slot[k] = -1;
            if (want[k]) {
                int prefix = 0;
                for (int w = 0; w < wave; w++) prefix += sWave[k * kMaxWavesPerBlock + w];
                slot[k] = sBase[k] + prefix + __popcll(mask[k] & ((1ull << lane) - 1ull));
            }
        }
    }
};

// The same allocator for tiles of U sub-tiles (the logic kernel: two items per thread; the material kernels keep SlotAllocator above,
// whose register allocation the generalised text disturbed: 18-21 -> 19-35 spilled VGPRs).  U: sub-tiles per tile.  A tile is U * blockDim.x consecutive items, thread t handles items t, t + blockDim.x, ... of it; the
// slots of sub-tile u lie in front of those of sub-tile u + 1 (ascending item index, as ORDERED needs).  One ticket, one look-back
// and one atomic per queue cover the whole tile: what a tile costs beside its items is paid once per U * blockDim.x of them.
template <bool ORDERED, int K, int U>
struct TileAllocator {
    int* sWave;  // [U * K][kMaxWavesPerBlock] per-wave counts of the current tile
    int* sBase;  // [U * K] base slot of the current tile's sub-tile u in queue k
    int* sTile;  // ORDERED: the ticket of the current tile
    int* counter0;  // the K counter words are counter0 + k * counterStep (an address computation, not an array of pointers: an
    int counterStep;  // array indexed by anything but a constant makes the allocator a private-memory object in LDS or scratch)
    NX_G unsigned long long* status;  // ORDERED: tile status words, [tile][kScanWords]
    NX_G int* ticket;
    NX_G FrameState* frame;
    uint32_t serial;
    int size, lastTile, tileItems;

    // `kind`: 0 logic, 1 + NX_MAT_* material kernel; `items`: size of the launch's input queue
    NXD void init(const DeviceState* S, int* const first, const int step, const int kind, const int bounce, const int items)
    {
        __shared__ int wave[U * K * kMaxWavesPerBlock];
        __shared__ int base[U * K];
        __shared__ int tile;
        sWave = wave;
        sBase = base;
        sTile = &tile;
        counter0 = first;
        counterStep = step;
        size = items;
        tileItems = U * (int)blockDim.x;
        lastTile = (items + tileItems - 1) / tileItems - 1;
        if (ORDERED) {
            status = S->scanStatus;
            ticket = &S->counters->scanTicket[kind][bounce];
            frame = S->frame;
            serial = S->frame->scanEpoch * 1024u + (uint32_t)bounce * 8u + (uint32_t)kind;
        }
    }
    NXD int* counter_of(const int k) const
    {
        return counter0 + k * counterStep;
    }
    // first item of the workgroup's first / next tile (>= size: none left)
    NXD int first_tile() { return ORDERED ? take() : (int)blockIdx.x * tileItems; }
    NXD int next_tile(const int tile) { return ORDERED ? take() : tile + (int)gridDim.x * tileItems; }
    NXD int take()
    {
        __syncthreads();  // the previous tile's readers are done with sTile
        if (threadIdx.x == 0) *sTile = atomicAdd(ticket, 1);
        __syncthreads();
        const int t = *sTile;
        return t > lastTile ? size : t * tileItems;
    }

    NXD int look_back(const int k, const int t, const int total) { return scan_look_back(status + k, counter_of(k), frame, serial, t, total); }

    // `tile`: first item of the tile (what first_tile / next_tile returned); `region`: the queue region this tile appends to
    // (uniform over the workgroup; ORDERED keeps one region); counters[] point at region 0's words.  want[u][k]: this thread's
    // item of sub-tile u goes to queue k.
    NXD void alloc(const bool (&want)[U][K], int (&slot)[U][K], const int tile, const int region = 0)
    {
        const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
        const int nWaves = blockDim.x / kWave;
        unsigned long long mask[U][K];
        __syncthreads();  // the previous tile's readers are done with sWave / sBase
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < K; k++) {
                mask[u][k] = __ballot(want[u][k]);
                if (lane == 0) sWave[(u * K + k) * kMaxWavesPerBlock + wave] = __popcll(mask[u][k]);
            }
        __syncthreads();
        if (ORDERED) {
            if (wave < K) {  // (a workgroup has at least K waves: 256 threads, K <= 4)
                int subTotal[U], total = 0;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    subTotal[u] = wave_sum(lane < nWaves ? sWave[(u * K + wave) * kMaxWavesPerBlock + lane] : 0);
                    total += subTotal[u];
                }
                const int t = tile / tileItems;
                const int excl = look_back(wave, t, total);
                if (lane == 0) {
                    int run = excl;
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        sBase[u * K + wave] = run;
                        run += subTotal[u];
                    }
                    if (t == lastTile) *counter_of(wave) = excl + total;
                }
            }
        } else if (threadIdx.x < K) {
            int subTotal[U], total = 0;
#pragma unroll
            for (int u = 0; u < U; u++) {
                subTotal[u] = 0;
                for (int w = 0; w < nWaves; w++) subTotal[u] += sWave[(u * K + (int)threadIdx.x) * kMaxWavesPerBlock + w];
                total += subTotal[u];
            }
            int* const word = counter_of((int)threadIdx.x) + region * kRegionStride;
            int run = total ? atomicAdd(word, total) : 0;
#ifdef NX_EXTRA_ATOMICS
            // experiment: are the logic / material kernels bound by the returning atomics on their queue
            // counters?  NX_EXTRA_ATOMICS more of them per tile and counter, adding zero
            for (int x = 0; x < NX_EXTRA_ATOMICS; x++)
                if (total && atomicAdd(word, 0) == -123456789) run = 0;  // (returning, result unused)
#endif
#pragma unroll
            for (int u = 0; u < U; u++) {
                sBase[u * K + (int)threadIdx.x] = run;
                run += subTotal[u];
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < K; k++) {
                slot[u][k] = -1;
                if (want[u][k]) {
                    int prefix = 0;
                    for (int w = 0; w < wave; w++) prefix += sWave[(u * K + k) * kMaxWavesPerBlock + w];
                    slot[u][k] = sBase[u * K + k] + prefix + __popcll(mask[u][k] & ((1ull << lane) - 1ull));
                }
            }
    }
};

NXD uint32_t global_pixel(const DeviceState* S, uint32_t local) { return S->pixelMap ? S->pixelMap[local] : local; }

// frame number and local pixel of a path; frameLast = number of the last frame of the current pass
struct PathId { uint32_t frame, pixel; };
NXD PathId path_id(const DeviceState* S, uint32_t pathIdx, uint32_t frameLast)
{
    const uint32_t slice = S->framesPerPass > 1u ? pathIdx / S->localCount : 0u;
    return PathId{frameLast - (S->framesPerPass - 1u) + slice, pathIdx - slice * S->localCount};
}

NXD uint32_t seed_for(const DeviceState* S, uint32_t slot, uint32_t pathIdx, uint32_t bounce, uint32_t stage, uint32_t frameLast)
{
    const PathId id = path_id(S, pathIdx, frameLast);
    if (S->rngMode == NX_RNG_PIXEL_KEYED) return rng_init_keyed(global_pixel(S, id.pixel), bounce, id.frame, stage);
    return rng_init_index(slot, S->camera.resolution[0], id.frame);
}

// ------------------------------------------------------------------------------------------------------
// begin pass: the pass size (frames batched into this pass) and the number of its last frame arrive as kernel arguments and
// are published to the other kernels through the device state, so neither a pass of a different size nor several passes
// in flight (each in its own slot, numbered by the host) need a host-side state upload or a synchronisation.  Also: zero
// every counter, traceSize[0] = localCount * frames (GenerateKernel's thread 0 in the reference, PathTracer.cu:112-113; the
// memset of PathTracer.cpp:263; the frame counter of PathTracer.cpp:250).

__global__ void __launch_bounds__(kWideBlock) begin_frame_kernel(DeviceState* __restrict__ S, const uint32_t frames, const uint32_t frameLast, const uint32_t scanEpoch)
{
    int* c = reinterpret_cast<int*>(S->counters);
    constexpr int n = (int)(sizeof(Counters) / sizeof(int));
    for (int i = threadIdx.x; i < n; i += blockDim.x) c[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        S->framesPerPass = frames;
        S->pathCount = S->localCount * frames;
        S->frame->frameNumber = frameLast;
        S->frame->scanEpoch = scanEpoch;
        // a pending pixel query (D_PixelQuery, PathTracer.cuh:54-58) starts the pass as "nothing hit": the bounce-1 material kernel
        // writes the instance a primary ray of that pixel hits (the reference's logic kernel writes -1 on a miss, PathTracer.cu:160;
        // the SCAN pipeline has no kernel that looks at misses when the background is black)
        if (S->frame->pixelQueryPixel >= 0) S->frame->pixelQueryInstance = -1;
    }
    if (threadIdx.x < kQueueShards) {
        // the primary rays: path i goes to region i / piece (generate_kernel)
        const uint32_t n = S->localCount * frames, piece = dense_piece(n, S->queueShards), k = threadIdx.x;
        S->counters->region[k].traceSize[0] = k < S->queueShards ? (int)min(piece, n - min(n, k * piece)) : 0;
    }
}

// The ray-batch hooks (nxhip_trace_batch / nxhip_trace_shadow_batch): `n` rays in the dense numbering of dense_piece — the sizes of
// the regions in use and zeroed fetch heads at bounce slot `slot`.
__global__ void hook_sizes_kernel(DeviceState* __restrict__ S, const uint32_t n, const int anyHit, const int slot)
{
    const uint32_t k = threadIdx.x;
    if (k >= (uint32_t)kQueueShards) return;
    const uint32_t shards = S->queueShards, piece = dense_piece(n, shards);
    const int32_t size = k < shards ? (int32_t)min(piece, n - min(n, k * piece)) : 0;
    NX_G RegionCounters* r = &S->counters->region[k];
    if (anyHit) { r->traceShadowSize[slot] = size; r->shadowHead[slot] = 0; }
    else { r->traceSize[slot] = size; r->traceHead[slot] = 0; }
    if (k == 0) { S->counters->thinCount[0][slot] = 0; S->counters->thinCount[1][slot] = 0; }  // (the hooks may run with the thin hand-over on: nxhip_debug_set_thin)
}

// ------------------------------------------------------------------------------------------------------
// GenerateKernel — PathTracer.cu:85-122

__global__ void __launch_bounds__(kWideBlock) generate_kernel(const DeviceState* __restrict__ S)
{
    const uint32_t n = S->pathCount;
    const uint32_t frameLast = S->frame->frameNumber;
    const uint32_t resX = S->camera.resolution[0], resY = S->camera.resolution[1];
    const f3 camPos = ld3(S->camera.position), camRight = ld3(S->camera.right), camUp = ld3(S->camera.up);
    const f3 llc = ld3(S->camera.lowerLeftCorner), vpX = ld3(S->camera.viewportX), vpY = ld3(S->camera.viewportY);
    const float lensRadius = S->camera.lensRadius;
    const uint32_t piece = dense_piece(n, S->queueShards), cap = S->queueShardCap;
    for (uint32_t index = blockIdx.x * blockDim.x + threadIdx.x; index < n; index += gridDim.x * blockDim.x) {
        const PathId id = path_id(S, index, frameLast);
        const uint32_t g = global_pixel(S, id.pixel);
        const uint32_t j = g / resX;
        const uint32_t i = g - j * resX;
        uint32_t rng = rng_init_pixel(i, j, resX, id.frame);
        const float x = ((float)i + rng_next(rng)) / (float)resX;
        const float y = ((float)j + rng_next(rng)) / (float)resY;
        const f2 disk = unit_disk(rng);
        const float rdx = lensRadius * disk.x, rdy = lensRadius * disk.y;
        const f3 offset = camRight * rdx + camUp * rdy;
        const f3 origin = camPos + offset;
        const f3 direction = normalize3((((llc + vpX * x) + vpY * y) - camPos) - offset);
        // the path's radiance starts at zero here (a coalesced 16-byte store) so that the first hit adds to it only when it
        // emits, like every later one.  (The path's previous vertex is kept by the logic step: keep_previous_vertex.)
        S->radiance[index] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        // throughput / lastPdf start as (1, 1, 1, 1e10): the bounce-1 logic and shade kernels use those constants instead of
        // reading them back, and the logic kernel stores them for every path that survives its first hit
        const uint32_t region = index / piece, slot = region * cap + (index - region * piece);
        // w: kRaySurvives — the bounce-1 logic step draws against a throughput of 1, and rng_next() < 1 always (PathTracer.cu:167-175
        // with the implicit throughput of :149)
        // ... and, with entry points on, the number of the entry state of the path's run of 64 (nx_entry.hip)
        const uint32_t entryBits = S->entry ? ((id.pixel >> 6) + 1u) << kRayEntryShift : 0u;
        S->trace.rays[0].rayO[slot] = make_float4(origin.x, origin.y, origin.z, __uint_as_float(kRaySurvives | entryBits));
        S->trace.rays[0].rayD[slot] = make_float4(direction.x, direction.y, direction.z, __uint_as_float(index));
    }
}

// ------------------------------------------------------------------------------------------------------
// SampleBackground — PathTracer.cu:65-83

// (u, v) of a direction on the latitude / longitude map — PathTracer.cu:65-83.  Computed ONCE per direction and handed to the
// colour lookup and to the sampler's density lookup alike: the arc functions are the expensive part (include/nexus_fmath.h
// evaluates them in binary64), and a miss under environment sampling needs both lookups for the same direction.
struct EnvUv { float u, v; };
NXD EnvUv env_uv(f3 d)
{
    const float theta = nxf_atan2f(d.z, d.x);
    const float phi = nxf_asinf(d.y);
    return EnvUv{(float)((theta + kPiD) * kInvPi * 0.5), (float)(1.0f - (phi + kPiD * 0.5f) * kInvPi)};
}
NXD f3 env_colour(const DeviceState* S, EnvUv uv)
{
    const float4 c = tex2d(S->hdrMap, S->srgbLut, uv.u, uv.v);
    return mk3(c.x, c.y, c.z);
}
NXD f3 sample_background(const DeviceState* S, f3 d)
{
    if (S->hdrMap.texels) return env_colour(S, env_uv(d));
    return ld3(S->settings.backgroundColor) * S->settings.backgroundIntensity;
}

// ------------------------------------------------------------------------------------------------------
// Environment importance sampling — an extension (the reference adds the environment on a miss only, PathTracer.cu:152-164,
// so an HDR map with a small bright sun converges very slowly): the NEE may pick the environment as one more light and
// draws its direction from the map's luminance distribution; a BSDF-sampled ray that misses is MIS-weighted against it.

// the texel (u, v) falls in
NXD uint32_t env_texel(const DeviceState* S, EnvUv uv)
{
    const int W = (int)S->hdrMap.width, H = (int)S->hdrMap.height;
    const int x = min(max((int)(uv.u * (float)W), 0), W - 1), y = min(max((int)(uv.v * (float)H), 0), H - 1);
    return (uint32_t)y * (uint32_t)W + (uint32_t)x;
}

// pdf per solid angle of the environment sampler for the unit direction d whose map coordinates are uv (light-selection
// probability excluded)
NXD float env_pdf(const DeviceState* S, f3 d, EnvUv uv)
{
    const float cosLat = sqrtf(fmaxf(1.0f - d.y * d.y, 1.0e-12f));
    return S->envDensity[env_texel(S, uv)] / cosLat;
}

// first index whose cdf exceeds r (n - 1 when none does).  `guide` brackets the answer: entry b is the first index whose cdf
// exceeds b / kEnvGuide, r lies in bucket floor(r * kEnvGuide), so the search starts from [guide[b], guide[b + 1]] instead
// of [0, n - 1] — the same index after 2-5 dependent loads instead of 10-11.
NXD int cdf_find(const NX_G float* cdf, const NX_G uint32_t* guide, int n, float r)
{
    const int b = min(kEnvGuide - 1, max(0, (int)(r * (float)kEnvGuide)));
    int lo = (int)guide[b], hi = min(n - 1, (int)guide[b + 1]);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] > r) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

NXD f3 env_sample(const DeviceState* S, float r1, float r2)
{
    const int W = (int)S->hdrMap.width, H = (int)S->hdrMap.height;
    const int y = cdf_find(S->envMarginalCdf, S->envMarginalGuide, H, r1);
    const float ylo = y ? S->envMarginalCdf[y - 1] : 0.0f;
    const float fy = (r1 - ylo) / (S->envMarginalCdf[y] - ylo);
    const NX_G float* row = S->envRowCdf + (size_t)y * (size_t)W;
    const int x = cdf_find(row, S->envRowGuide + (size_t)y * (size_t)(kEnvGuide + 1), W, r2);
    const float xlo = x ? row[x - 1] : 0.0f;
    const float fx = (r2 - xlo) / (row[x] - xlo);
    const float u = ((float)x + fx) / (float)W, v = ((float)y + fy) / (float)H;
    const float phi = (1.0f - v) * 3.14159265f - 1.57079633f, theta = u * 6.28318531f - 3.14159265f;
    const float c = nxf_cosf(phi);
    return mk3(c * nxf_cosf(theta), nxf_sinf(phi), c * nxf_sinf(theta));
}

// lights the NEE chooses among: the mesh lights, plus the environment when it is importance sampled
NXD uint32_t nee_light_count(const DeviceState* S) { return S->lightCount + ((S->envSampling && S->hdrMap.texels) ? 1u : 0u); }

// ------------------------------------------------------------------------------------------------------
// LogicKernel — PathTracer.cu:136-210

// LogicKernel's decision for one path (PathTracer.cu:136-210), without the queue traffic: a miss ends the path and adds the
// (MIS-weighted) environment `bg`; a hit survives Russian roulette with probability max(throughput) — `survived`, the
// throughput divided by it in `throughputOut` — and is shaded as the returned material type, or ends (-1).
template <class HitInst>
NXD int logic_path(const DeviceState* S, const int bounce, const uint32_t frame, const uint32_t seedSlot, const uint32_t pixelIdx, const float hitT, const f3 dir,
                   const float4 tp, HitInst hitInst, bool& miss, f3& bg, bool& survived, f3& throughputOut, uint32_t& inst, bool& needsPrevVertex)
{
    needsPrevVertex = false;
    const f3 throughput = mk3(tp.x, tp.y, tp.z);
    int type = -1;
    miss = false;
    survived = false;
    if (hitT == 1e30f) {
        miss = true;
        EnvUv uv{0.0f, 0.0f};
        if (S->hdrMap.texels) {
            uv = env_uv(dir);
            bg = throughput * env_colour(S, uv);
        } else {
            bg = throughput * sample_background(S, dir);
        }
        if (S->envSampling && S->hdrMap.texels && bounce > 1 && S->settings.useMIS) {
            // the NEE samples the environment too: weight the BSDF-sampled miss against it (extension)
            const float envPdf = env_pdf(S, dir, uv) / (float)nee_light_count(S);
            if (pdf_valid(envPdf)) bg = bg * power_heuristic(tp.w, envPdf);
        }
    } else {
        uint32_t rng = seed_for(S, seedSlot, pixelIdx, (uint32_t)bounce, 0u, frame);
        const float probability = maxcomp3(throughput);
        if (rng_next(rng) < probability) {
            survived = true;
            throughputOut = throughput / probability;
            inst = hitInst();
            // type and, behind it, the flag nxhip_set_materials derived: the material can emit (an emissive hit weighs itself
            // against the light sampler by the distance from the path's previous vertex) or let a path pass through (which
            // keeps the path state, PathTracer.cu:372-386) — see keep_previous_vertex below.  Read from the instance's shading
            // record (one dependent load; the reference's instance -> material is two)
            const uint32_t typeAndFlag = *(const NX_G uint32_t*)((const NX_G char*)&S->shadeInst[inst].material + kMaterialTypeOffset);
            type = (int)(int8_t)(typeAndFlag & 0xffu);
            if (type < 0 || type > 3) type = -1;
            needsPrevVertex = ((typeAndFlag >> 8) & 1u) != 0u;
        }
    }
    return type;
}

// The previous vertex of a path (D_PathStateSOA::rayOrigin, PathTracer.cuh:19-30) is the origin of its last SAMPLED ray.  The
// material kernels used to store it per path and bounce (a scattered 16-byte store each: 7 % of their time) although it is
// read only by the rare emissive hit under MIS.  Now the logic step stores it, and only for hits that can need it: the
// origin of the ray that produced the hit — unless that ray was a pass-through continuation (flagged in the ray's w), whose
// predecessor hit a pass-through-capable material and therefore stored the right vertex itself.  Same values as before.
NXD void keep_previous_vertex(const DeviceState* S, const uint32_t pixelIdx, const float4 rayOrigin)
{
    if ((__float_as_uint(rayOrigin.w) & kRayPassThrough) == 0u) S->rayOrigin[pixelIdx] = make_float4(rayOrigin.x, rayOrigin.y, rayOrigin.z, 0.0f);
}

template <bool ORDERED, int U>
// 8 waves per SIMD (60 VGPRs, no spills): two of the 1024-thread workgroups fit a CU instead of one, so the barriers of the
// slot allocation in one overlap with the streaming of the other (logic kernel -16 %, bench +1 %)
#ifndef NX_LOGIC_WAVES
#define NX_LOGIC_WAVES 8
#endif
__global__ void __launch_bounds__(kLogicBlock, NX_LOGIC_WAVES) logic_kernel(const DeviceState* __restrict__ S, const int bounce)
{
    // U items per thread: a tile is U * 1 024 items behind ONE round of slot allocation
    Counters* C = S->counters;
    const QueueView in = queue_view(&C->region[0].traceSize[bounce - 1], S->queueShardCap);
    const int size = in.total;
    // the grid is sized for the largest queue: a workgroup with no tile to process leaves before the allocator's
    // barriers (late bounces carry a few thousand items; an all-empty launch used to cost 30 us)
    if ((int)(blockIdx.x * blockDim.x) * U >= size) return;
    const uint32_t frame = S->frame->frameNumber;
    TileAllocator<ORDERED, 4, U> slots;
    // queue k of this kernel: the material queue of type k
    slots.init(S, &C->region[0].materialSize[0][bounce], kMaxBounceSlots, 0, bounce, size);
    const ProducerRegions out = producer_regions(S, size, U * (int)blockDim.x);
    for (int tile = slots.first_tile(); tile < size; tile = slots.next_tile(tile)) {
        int type[U];
        float4 hit[U], dirPix[U], tpOut[U];
        uint32_t inst[U], pixelIdx[U];
        bool want[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int index = tile + u * (int)blockDim.x + (int)threadIdx.x;
            type[u] = -1;
            hit[u] = make_float4(0, 0, 0, 0); dirPix[u] = make_float4(0, 0, 0, 0); tpOut[u] = make_float4(0, 0, 0, 0);
            inst[u] = 0; pixelIdx[u] = 0;
            if (index < size) {
                const int at = in.slot(index);  // where item `index` of the trace queue lives
                hit[u] = S->trace.hit[at];
                dirPix[u] = S->trace.rays[0].rayD[at];
                pixelIdx[u] = __float_as_uint(dirPix[u].w);
                const float4 tp = bounce == 1 ? make_float4(1.0f, 1.0f, 1.0f, 1.0e10f) : S->trace.rays[0].tp[at];
                // the hit's instance is loaded with the rest of the entry, not behind the roulette decision that first needs it: a
                // load inside the branch is one more dependent round trip per tile for the paths that survive (4 B per item more for
                // those that do not; logic kernel -4 %)
                const uint32_t hitInstance = S->trace.hitInst[at];
                bool miss, survived, needsPrevVertex;
                f3 bg = mk3(0.0f), t = mk3(0.0f);
                type[u] = logic_path(S, bounce, frame, (uint32_t)index, pixelIdx[u], hit[u].x, mk3(dirPix[u].x, dirPix[u].y, dirPix[u].z), tp, [&]() { return hitInstance; }, miss, bg, survived, t,
                                     inst[u], needsPrevVertex);
                if (needsPrevVertex) keep_previous_vertex(S, pixelIdx[u], S->trace.rays[0].rayO[at]);
                if (miss) {
                    // A background contribution of exactly +0 in all three components (a black environment: the reference's default
                    // backgroundIntensity 0) leaves the pixel's radiance as it is — generate_kernel zeroed it, later additions never
                    // produce a negative zero from non-negative emission — so the scattered 16-byte read-modify-write is skipped:
                    // it was a third of this kernel's memory traffic on such scenes.  (Anything else, -0 included, is added.)
                    if ((__float_as_uint(bg.x) | __float_as_uint(bg.y) | __float_as_uint(bg.z)) != 0u) {
                        float4 r = bounce == 1 ? make_float4(0, 0, 0, 0) : S->radiance[pixelIdx[u]];
                        r.x += bg.x; r.y += bg.y; r.z += bg.z;
                        if (bounce == 1) r = make_float4(bg.x, bg.y, bg.z, 0.0f);
                        S->radiance[pixelIdx[u]] = r;
                    }
                    if (bounce == 1 && pixelIdx[u] < S->localCount && S->frame->pixelQueryPixel == (int)global_pixel(S, pixelIdx[u])) S->frame->pixelQueryInstance = -1;
                }
                // (at bounce 1 every hit survives with throughput 1 and the material kernels use that constant instead of reading
                //  it back)
                if (survived) tpOut[u] = make_float4(t.x, t.y, t.z, tp.w);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) want[u][k] = type[u] == k;
        }
        int slot[U][4];
        const int region = out.of_tile(tile), regionBase = region * (int)S->queueShardCap;
        slots.alloc(want, slot, tile, region);
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (type[u] >= 0) {
                const MaterialQueue mq = S->material[type[u]];
                const int sl = regionBase + (type[u] == 0 ? slot[u][0] : (type[u] == 1 ? slot[u][1] : (type[u] == 2 ? slot[u][2] : slot[u][3])));
                // the hit distance is of no use to the material kernels (they work from u, v): its slot carries the path index,
                // which saves a third array (4 B written and read per path, one load and one store instruction each)
                mq.hit[sl] = make_float4(__uint_as_float(pixelIdx[u]), hit[u].y, hit[u].z, hit[u].w);
                mq.dirInst[sl] = make_float4(dirPix[u].x, dirPix[u].y, dirPix[u].z, __uint_as_float(inst[u]));
                if (bounce != 1) mq.tp[sl] = tpOut[u];  // the path's throughput after Russian roulette and its last pdf go with it
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Shade<BSDF> + NextEventEstimation<BSDF> — PathTracer.cu:311-458, 213-308

NXD MatParams load_params(const nx_material& m)
{
    MatParams p;
    p.albedo = ld3(m.diffuse.albedo);
    p.roughness = (m.type == NX_MAT_CONDUCTOR) ? m.conductor.roughness : m.plastic.roughness;
    p.ior = m.plastic.ior;
    p.cIor = ld3(m.conductor.ior);
    p.cK = ld3(m.conductor.k);
    return p;
}

NXD float tri_area(f3 p0, f3 p1, f3 p2) { return 0.5f * length3(cross3(p1 - p0, p2 - p0)); }

struct ShadowPayload {
    f3 origin, direction, radiance;
    float distance;
};

template <int TYPE>
NXD bool next_event_estimation(const DeviceState* S, f3 wi, const MatParams& mp, f3 hitPoint, f3 normal, f3 hitGNormal, f3 throughput,
                               uint32_t& rng, ShadowPayload& out)
{
    // no lights: the reference indexes an empty array here (undefined); defined as "no light sample, no random numbers
    // drawn", identically in the CPU restatement used by the tests
    const uint32_t nLights = nee_light_count(S);
    if (nLights == 0u) return false;
    const uint32_t pick = uniform_index(nLights, rng);
    if (pick >= S->lightCount) {
        // the environment (extension): direction from the map's luminance distribution, shadow ray to infinity
        const float r1 = rng_next(rng), r2 = rng_next(rng);
        const f3 shDir = env_sample(S, r1, r2);
        // (the BSDF first: a direction it rejects — most of them on a near-specular surface — needs no density lookup)
        const float4 q = rotation_to_z(normal);
        const f3 wo = rotate_point(q, shDir);
        f3 sampleThroughput;
        float bsdfPdf;
        if (!Bsdf<TYPE>::eval(mp, wi, wo, sampleThroughput, bsdfPdf)) return false;
        const EnvUv uv = env_uv(shDir);
        const float lightPdf = env_pdf(S, shDir, uv) / (float)nLights;
        if (!pdf_valid(lightPdf)) return false;
        const float weight = power_heuristic(lightPdf, bsdfPdf);
        out.radiance = (((throughput * weight) * sampleThroughput) * env_colour(S, uv)) / lightPdf;
        out.origin = offset_ray(hitPoint, hitGNormal * sgnE(dot3(shDir, normal)));
        out.direction = shDir;
        out.distance = 1e30f;
        return true;
    }
    const nx_light light = S->lights[pick];
    if (light.type != NX_LIGHT_MESH) return false;
    // the light's instance: its shading record (transform, triangles, material behind one load)
    const NX_G ShadeInst* inst = &S->shadeInst[light.mesh.meshId];
    const uint32_t lightTriCount = inst->triCount;
    const uint32_t triangleIdx = uniform_index(lightTriCount, rng);
    const f2 uv = uniform_triangle(rng);
    const NX_G nx_triangle* tri = shade_tri(inst->tris, triangleIdx);
    const f3 tp0 = ld3(tri->pos0), tp1 = ld3(tri->pos1), tp2 = ld3(tri->pos2);
    const NX_G float* T = inst->transform;
    const NX_G float* IT = inst->invTransform;

    f3 p = mat_point(T, bary3(tp0, tp1, tp2, uv.x, uv.y));
    const f3 lightGNormal = normalize3(mat_vec_transposed(IT, cross3(tp1 - tp0, tp2 - tp0)));
    const f3 lightNormal = normalize3(mat_vec_transposed(IT, bary3(ld3(tri->normal0), ld3(tri->normal1), ld3(tri->normal2), uv.x, uv.y)));

    f3 toLight = p - hitPoint;
    float offsetDirection = sgnE(dot3(toLight, normal));
    out.origin = offset_ray(hitPoint, hitGNormal * offsetDirection);
    offsetDirection = sgnE(dot3(-toLight, lightNormal));
    p = offset_ray(p, lightGNormal * offsetDirection);

    toLight = p - out.origin;
    out.distance = length3(toLight);
    out.direction = toLight / out.distance;

    const float4 q = rotation_to_z(normal);
    const f3 wo = rotate_point(q, out.direction);
    const float cosThetaO = fabsf(dot3(lightNormal, out.direction));
    const float dSquared = dot3(toLight, toLight);
    const float area = tri_area(mat_point(T, tp0), mat_point(T, tp1), mat_point(T, tp2));
    float lightPdf = 1.0f / ((float)(nLights * lightTriCount) * area);
    lightPdf *= dSquared / cosThetaO;
    if (!pdf_valid(lightPdf)) return false;

    const NX_G nx_material* lightMaterial = &inst->material;
    f3 sampleThroughput;
    float bsdfPdf;
    if (!Bsdf<TYPE>::eval(mp, wi, wo, sampleThroughput, bsdfPdf)) return false;
    const float weight = power_heuristic(lightPdf, bsdfPdf);

    f3 emissive;
    if (lightMaterial->emissiveMapId != -1) {
        const f2 t = bary2(tri->texCoord0, tri->texCoord1, tri->texCoord2, uv.x, uv.y);
        const float4 c = tex2d(S->emissiveMaps[lightMaterial->emissiveMapId], S->srgbLut, t.x, t.y);
        emissive = mk3(c.x, c.y, c.z);
    } else {
        emissive = ld3(lightMaterial->emissive);
    }
    out.radiance = ((((throughput * weight) * sampleThroughput) * emissive) * lightMaterial->intensity) / lightPdf;
    return true;
}

// Shade<BSDF> for one path (PathTracer.cu:311-458), without the queue traffic: shade_kernel wraps it with the material-queue
// loads and the slot allocation, tail_kernel calls it in its per-path loop.  prevOrigin() returns the path's previous vertex
// (only read for an emissive hit under MIS).  emit(radiance, instance) receives what the hit emits towards the path; what comes out: the shadow ray of its
// light sample, the continuation ray and the path state that goes with it.  (Separate references, not a struct: the
// compiler kept a struct of these in scratch memory, +30 % on the material kernels.)
template <int TYPE, class PrevOrigin, class Emit>
NXD void shade_path(const DeviceState* S, const int bounce, const uint32_t frame, const uint32_t seedSlot, const uint32_t pixelIdx, const float hu, const float hv,
                    const uint32_t triIdx, const uint32_t instanceIdx, const f3 rayDirection, const float4 tpdf, PrevOrigin prevOrigin, Emit emit,
                    bool& wantShadow, ShadowPayload& sh, bool& wantTrace, bool& updatePath, f3& nextOrigin, f3& nextDir, f3& nextThroughput, float& nextPdf)
{
    wantShadow = false; wantTrace = false; updatePath = false;
    nextOrigin = mk3(0.0f); nextDir = mk3(0.0f); nextThroughput = mk3(0.0f);
    nextPdf = 0.0f;
    f3 throughput = mk3(tpdf.x, tpdf.y, tpdf.z);
    uint32_t rng = seed_for(S, seedSlot, pixelIdx, (uint32_t)bounce, 1u, frame);

    const NX_G ShadeInst* inst = &S->shadeInst[instanceIdx];
#ifdef NX_SHADE_SEQ_TRI
    // BOUND EXPERIMENT (wrong results): every item reads the shading triangle of its QUEUE SLOT instead of its hit's —
    // consecutive items, consecutive 96-byte records — so that what is left of the material launch is everything but the random gather
    const NX_G nx_triangle* tri = shade_tri(inst->tris, seedSlot % max(inst->triCount, 1u));
#else
    const NX_G nx_triangle* tri = shade_tri(inst->tris, triIdx);
#endif
    const nx_material material = inst->material;
    MatParams mp = load_params(material);
    const NX_G float* T = inst->transform;
    const NX_G float* IT = inst->invTransform;
    const f3 tp0 = ld3(tri->pos0), tp1 = ld3(tri->pos1), tp2 = ld3(tri->pos2);

    const f3 p = mat_point(T, bary3(tp0, tp1, tp2, hu, hv));
    f3 normal = bary3(ld3(tri->normal0), ld3(tri->normal1), ld3(tri->normal2), hu, hv);
    const f2 texUv = bary2(tri->texCoord0, tri->texCoord1, tri->texCoord2, hu, hv);
    normal = normalize3(mat_vec_transposed(IT, normal));
    f3 gNormal = normalize3(mat_vec_transposed(IT, cross3(tp1 - tp0, tp2 - tp0)));

    f3 emissive = ld3(material.emissive);
    if (material.emissiveMapId != -1) {
        const float4 c = tex2d(S->emissiveMaps[material.emissiveMapId], S->srgbLut, texUv.x, texUv.y);
        emissive = mk3(c.x, c.y, c.z);
    }
    const bool useMIS = S->settings.useMIS != 0;
    const bool allowMIS = bounce > 1 && useMIS;
    f3 radiance = mk3(0.0f);
    if (maxcomp3(emissive * material.intensity) > 0.0f) {
        float weight = 1.0f;
        if (allowMIS) {
            const float lastPdf = tpdf.w;
            const float cosThetaO = fabsf(dot3(normal, rayDirection));
            const float4 ro = prevOrigin();
            const float dSquared = squaref(length3(p - mk3(ro.x, ro.y, ro.z)));
            const float area = tri_area(mat_point(T, tp0), mat_point(T, tp1), mat_point(T, tp2));
            float lightPdf = 1.0f / ((float)(nee_light_count(S) * inst->triCount) * area);
            lightPdf *= dSquared / cosThetaO;
            if (!pdf_valid(lightPdf)) weight = 0.0f;
            else weight = power_heuristic(lastPdf, lightPdf);
        }
        radiance = ((emissive * weight) * material.intensity) * throughput;
    }
    emit(radiance, instanceIdx);  // (here, before the sampling code: the three values need not live across it)

    if (bounce != (int)S->settings.pathLength) {
        float4 color = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        if (material.diffuseMapId != -1) {
            color = tex2d(S->diffuseMaps[material.diffuseMapId], S->srgbLut, texUv.x, texUv.y);
            mp.albedo = mk3(color.x, color.y, color.z);
        }
        if (dot3(gNormal, rayDirection) > 0.0f && TYPE != NX_MAT_DIELECTRIC) { normal = -normal; gNormal = -gNormal; }

        const float4 q = rotation_to_z(normal);
        const f3 wi = rotate_point(q, -rayDirection);
        f3 wo;
        if (rng_next(rng) > material.opacity || (material.diffuseMapId != -1 && rng_next(rng) > color.w)) {
            // texture / opacity pass-through: continue straight, path state untouched
            wo = normalize3(rotate_point(invert_rotation(q), -wi));
            const float od = sgnE(dot3(wo, normal));
            nextOrigin = offset_ray(p, gNormal * od);
            nextDir = wo;
            wantTrace = true;
        } else {
            if (useMIS) wantShadow = next_event_estimation<TYPE>(S, wi, mp, p, normal, gNormal, throughput, rng, sh);
            float pdf;
            f3 sampleThroughput;
            if (Bsdf<TYPE>::sample(mp, wi, rng, wo, sampleThroughput, pdf)) {
                wo = normalize3(rotate_point(invert_rotation(q), wo));
                const float od = sgnE(dot3(wo, normal));
                nextOrigin = offset_ray(p, gNormal * od);
                nextDir = wo;
                nextThroughput = throughput * sampleThroughput;
                nextPdf = pdf;
                wantTrace = true;
                updatePath = true;
            }
        }
    }
}

// Workgroups of 256 at 5 waves per SIMD (96 VGPRs, 5-11 spilled once per path), 10 workgroups per CU: equal to 512 threads
// at 4 waves (126 VGPRs) for one large pass, +4.5 % for one-frame passes in flight, where a smaller register footprint lets
// the material kernels of one slot share SIMDs with the trace waves of another.  6 and 8 waves per SIMD spill 30-87 VGPRs and
// double the kernel's time: it is sensitive to memory traffic, not short of waves.
template <int TYPE, bool ORDERED>
#ifndef NX_SHADE_WAVES
#define NX_SHADE_WAVES 5
#endif
#ifndef NX_SHADE_WAVES_ORDERED
#define NX_SHADE_WAVES_ORDERED 4
#endif
// (ordered: one 1 024-thread workgroup per CU is 4 waves per SIMD whatever the register budget says, so it may as well be 128)
__global__ void __launch_bounds__(ORDERED ? kShadeBlockOrderedThreads : kShadeBlock, ORDERED ? NX_SHADE_WAVES_ORDERED : NX_SHADE_WAVES) shade_kernel(const DeviceState* __restrict__ S, const int bounce)
{
    Counters* C = S->counters;
    const QueueView in = queue_view(&C->region[0].materialSize[TYPE][bounce], S->queueShardCap);
    const int size = in.total;
    if ((int)(blockIdx.x * blockDim.x) >= size) return;  // no tile for this workgroup (see logic_kernel)
    const uint32_t frame = S->frame->frameNumber;
    const MaterialQueue mq = S->material[TYPE];
    SlotAllocator<ORDERED, 2> slots;  // 0: shadow requests, 1: continuation rays
    static_assert(offsetof(RegionCounters, traceSize) + kMaxBounceSlots * sizeof(int32_t) == offsetof(RegionCounters, traceShadowSize), "traceShadowSize follows traceSize");
    slots.init(S, &C->region[0].traceShadowSize[bounce], -kMaxBounceSlots, 1 + TYPE, bounce, size);
    const ProducerRegions out = producer_regions(S, size, (int)blockDim.x);
    for (int tile = slots.first_tile(); tile < size; tile = slots.next_tile(tile)) {
        const int requestIdx = tile + (int)threadIdx.x;
        bool wantShadow = false, wantTrace = false, updatePath = false;
        ShadowPayload sh;
        f3 nextOrigin = mk3(0.0f), nextDir = mk3(0.0f), nextThroughput = mk3(0.0f);
        float nextPdf = 0.0f;
        uint32_t pixelIdx = 0;
        float4 tpdf = make_float4(0, 0, 0, 0);

        if (requestIdx < size) {
            const int at = in.slot(requestIdx);
            const float4 hit = mq.hit[at];
            const float4 dirInst = mq.dirInst[at];
            pixelIdx = __float_as_uint(hit.x);
            const uint32_t instanceIdx = __float_as_uint(dirInst.w);
            tpdf = bounce == 1 ? make_float4(1.0f, 1.0f, 1.0f, 1.0e10f) : mq.tp[at];
            shade_path<TYPE>(S, bounce, frame, (uint32_t)requestIdx, pixelIdx, hit.y, hit.z, __float_as_uint(hit.w), instanceIdx, mk3(dirInst.x, dirInst.y, dirInst.z), tpdf,
                             [&]() { return S->rayOrigin[pixelIdx]; },
                             [&](f3 emitted, uint32_t instIdx) {
                                 if (bounce == 1 && bounce != (int)S->settings.pathLength && pixelIdx < S->localCount && S->frame->pixelQueryPixel == (int)global_pixel(S, pixelIdx))
                                     S->frame->pixelQueryInstance = (int)instIdx;
                                 if (emitted.x != 0.0f || emitted.y != 0.0f || emitted.z != 0.0f) {
                                     // (most hits emit nothing: adding zeros would cost a 16-byte read and write per path and bounce)
                                     float4 r = S->radiance[pixelIdx];
                                     r.x += emitted.x; r.y += emitted.y; r.z += emitted.z;
                                     S->radiance[pixelIdx] = r;
                                 }
                             },
                             wantShadow, sh, wantTrace, updatePath, nextOrigin, nextDir, nextThroughput, nextPdf);
        }
        const bool want[2] = {wantShadow, wantTrace};
        int slot[2];
        const int region = out.of_tile(tile), regionBase = region * (int)S->queueShardCap;
        slots.alloc(want, slot, tile, region);
        const int shadowSlot = regionBase + slot[0], traceSlot = regionBase + slot[1];
        if (wantShadow) {
            S->shadow.rayO[shadowSlot] = make_float4(sh.origin.x, sh.origin.y, sh.origin.z, sh.distance);
            S->shadow.rayD[shadowSlot] = make_float4(sh.direction.x, sh.direction.y, sh.direction.z, __uint_as_float(pixelIdx));
            S->shadow.radiance[shadowSlot] = make_float4(sh.radiance.x, sh.radiance.y, sh.radiance.z, 0.0f);
        }
        if (wantTrace) {
            // w: kRayPassThrough for a pass-through continuation (the path's previous vertex stays what it was: keep_previous_vertex)
            S->trace.rays[0].rayO[traceSlot] = make_float4(nextOrigin.x, nextOrigin.y, nextOrigin.z, __uint_as_float(updatePath ? 0u : kRayPassThrough));
            S->trace.rays[0].rayD[traceSlot] = make_float4(nextDir.x, nextDir.y, nextDir.z, __uint_as_float(pixelIdx));
            // the path state that goes with the ray: the new one, or — a pass-through — the one the path arrived with
            S->trace.rays[0].tp[traceSlot] = updatePath ? make_float4(nextThroughput.x, nextThroughput.y, nextThroughput.z, nextPdf) : tpdf;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// The SCAN pipeline's material kernels: Shade<BSDF> straight from the trace queue, no logic kernel, no material queues.
//
// The reference's logic kernel (PathTracer.cu:136-210) reads every trace result, decides miss / Russian roulette / material
// type and COPIES the survivors into one of four material queues (48 B per path and bounce, :183-206) for the Shade kernels to
// read back.  Here the decision is already in the hit record when the closest-hit launch ends — the roulette draw is made by the
// ray's producer and rides in the ray (kRaySurvives), the material type rides in the instance record (InstTrav::instIdx), the
// trace kernel's flush combines the two into a code beside the hit's instance (nx_trace.hip) — so a material kernel only has to
// FIND its items: a workgroup walks its share of the trace queue 1 024 rays at a time (one 16-byte load of four instance words
// per thread), collects the slots whose code is its type in an LDS ring (ballots + one LDS atomic per wave), and shades them
// 256 at a time with every lane busy, reading hit, direction and path state at the ray's own slot (ascending within a batch).
// Per path and bounce that is 4 B scanned per material kernel in the graph + 52 B gathered, against the 52 B in / 48 B out of
// the logic kernel + 48 B in of the material kernel; and one launch less per bounce.  The rays alternate between two sets by
// bounce parity (TraceQueue), because a kernel of bounce b reads the rays of b - 1 while it writes those of b.
//
// Regions: workgroup w serves region w % 8 of the input queue (its blockIdx % 8 is its XCD: the XCD whose trace waves call that
// region home) and appends to the same region of the output queues, so a region never holds more than it did the bounce before
// (the primary pieces are even), and no batch mixes regions.  Fast compaction only (slot order = order of the atomics); the
// ordered mode keeps the classic pipeline, which is the reference's serial-slot semantics.
constexpr int kScanRing = 2048;  // LDS ring of found slots: fewer than 256 left over + 1 024 new ones at most
static_assert(kScanRing >= kShadeBlock + 4 * kShadeBlock && (kScanRing & (kScanRing - 1)) == 0, "scan geometry");

// One material type's share of a workgroup's work.  Tiles of `tileRays` rays (1 024: four instance words per thread in one 16-byte
// load; 256 when the region holds too few rays to give every workgroup a large tile — late bounces, whose launches are as long as
// their longest workgroup): tile `firstTile` is this workgroup's by its rank (-1: none, it comes over from another type), tiles
// from `staticTiles` on are handed out by ticket — one returning atomic per tile on the region's own word.  The share of a tile
// that is this type's to shade varies from none to all with what the camera sees there, so a static split leaves most workgroups
// waiting for the few whose tiles were full (measured: +40 % on the kernels).
template <int TYPE>
NXD void shade_scan_type(const DeviceState* __restrict__ S, const int bounce, const int region, const int inRegion, const int per, const int firstTile, const int staticTiles,
                         int* const sRing, int* const sHead, int* const sTicket)
{
    // TYPE == kScanMiss: the rays that missed (code kHitCodeMiss) — what is left of the logic step when a miss can contribute: the
    // environment (flat colour or map, MIS-weighted against the environment sampler) added to the path's radiance, PathTracer.cu:152-164
    constexpr bool kMiss = TYPE == kScanMiss;
    constexpr uint32_t kCode = kMiss ? kHitCodeMiss : (uint32_t)(TYPE + 1);
    Counters* C = S->counters;
    const int tileRays = per * kShadeBlock;
    const int tiles = (inRegion + tileRays - 1) / tileRays;
    int* const ticket = &C->region[region].scanTile[TYPE][bounce];
    int t = firstTile;
    if (t < 0 && tiles <= staticTiles) return;  // every tile has a workgroup of this type's own
    __syncthreads();  // the previous type is done with the ticket word and the ring
    if (threadIdx.x == 0) {
        *sHead = 0;
        if (t < 0) *sTicket = staticTiles + atomicAdd(ticket, 1);
    }
    __syncthreads();
    if (t < 0) {
        t = *sTicket;
        if (t >= tiles) return;
    }
    const uint32_t frame = S->frame->frameNumber;
    const int regionBase = region * (int)S->queueShardCap;
    const TraceRays in = S->trace.rays[(bounce - 1) & 1], out = S->trace.rays[bounce & 1];
    const NX_G uint32_t* const codes = S->trace.hitInst + regionBase;
    SlotAllocator<false, 2> slots;  // 0: shadow requests, 1: continuation rays
    static_assert(offsetof(RegionCounters, traceSize) + kMaxBounceSlots * sizeof(int32_t) == offsetof(RegionCounters, traceShadowSize), "traceShadowSize follows traceSize");
    slots.init(S, &C->region[0].traceShadowSize[bounce], -kMaxBounceSlots, 1 + (kMiss ? 0 : TYPE), bounce, inRegion);
    const int lane = threadIdx.x & (kWave - 1);
    const unsigned long long laneLt = (1ull << lane) - 1ull;
    int tail = 0, head = 0;  // ring positions [tail, head) hold found slots not shaded yet (uniform)

    // shades `take` <= 256 slots from the ring's tail
    const auto shade_batch = [&](const int take) {
        const bool have = (int)threadIdx.x < take;
        const int at = regionBase + (have ? sRing[(tail + (int)threadIdx.x) & (kScanRing - 1)] : 0);
        tail += take;
        if constexpr (kMiss) {
            if (have) {
                const float4 dirPix = in.rayD[at];
                const uint32_t pixelIdx = __float_as_uint(dirPix.w);
                const float4 tp = bounce == 1 ? make_float4(1.0f, 1.0f, 1.0f, 1.0e10f) : in.tp[at];
                bool miss, survived, needsPrevVertex;
                f3 bg = mk3(0.0f), t = mk3(0.0f);
                uint32_t inst = 0;
                logic_path(S, bounce, frame, (uint32_t)at, pixelIdx, 1e30f, mk3(dirPix.x, dirPix.y, dirPix.z), tp, [&]() { return 0u; }, miss, bg, survived, t, inst, needsPrevVertex);
                if ((__float_as_uint(bg.x) | __float_as_uint(bg.y) | __float_as_uint(bg.z)) != 0u) {  // (as the logic kernel: +0 changes nothing)
                    float4 r = bounce == 1 ? make_float4(0, 0, 0, 0) : S->radiance[pixelIdx];
                    r.x += bg.x; r.y += bg.y; r.z += bg.z;
                    if (bounce == 1) r = make_float4(bg.x, bg.y, bg.z, 0.0f);
                    S->radiance[pixelIdx] = r;
                }
            }
            return;
        } else {
        bool wantShadow = false, wantTrace = false, updatePath = false;
        ShadowPayload sh;
        f3 nextOrigin = mk3(0.0f), nextDir = mk3(0.0f), nextThroughput = mk3(0.0f);
        float nextPdf = 0.0f;
        uint32_t pixelIdx = 0;
        float4 tpdf = make_float4(0, 0, 0, 0);
        if (have) {
            const uint32_t instanceIdx = S->trace.hitInst[at] & kHitInstMask;
            const float4 hit = S->trace.hit[at];
            const float4 dirPix = in.rayD[at];
            pixelIdx = __float_as_uint(dirPix.w);
            // the rest of the logic step (PathTracer.cu:167-175): the survivor's throughput is divided by the roulette's
            // probability (at bounce 1 both are 1)
            tpdf = make_float4(1.0f, 1.0f, 1.0f, 1.0e10f);
            if (bounce != 1) {
                const float4 tp = in.tp[at];
                const f3 throughput = mk3(tp.x, tp.y, tp.z);
                const f3 tq = throughput / maxcomp3(throughput);
                tpdf = make_float4(tq.x, tq.y, tq.z, tp.w);
            }
            // ... and the path's previous vertex, for hits that can need it (keep_previous_vertex)
            const uint32_t typeAndFlag = *(const NX_G uint32_t*)((const NX_G char*)&S->shadeInst[instanceIdx].material + kMaterialTypeOffset);
            if ((typeAndFlag >> 8) & 1u) keep_previous_vertex(S, pixelIdx, in.rayO[at]);
            shade_path<TYPE>(S, bounce, frame, (uint32_t)at, pixelIdx, hit.y, hit.z, __float_as_uint(hit.w), instanceIdx, mk3(dirPix.x, dirPix.y, dirPix.z), tpdf,
                             [&]() { return S->rayOrigin[pixelIdx]; },
                             [&](f3 emitted, uint32_t instIdx) {
                                 if (bounce == 1 && bounce != (int)S->settings.pathLength && pixelIdx < S->localCount && S->frame->pixelQueryPixel == (int)global_pixel(S, pixelIdx))
                                     S->frame->pixelQueryInstance = (int)instIdx;
                                 if (emitted.x != 0.0f || emitted.y != 0.0f || emitted.z != 0.0f) {
                                     float4 r = S->radiance[pixelIdx];
                                     r.x += emitted.x; r.y += emitted.y; r.z += emitted.z;
                                     S->radiance[pixelIdx] = r;
                                 }
                             },
                             wantShadow, sh, wantTrace, updatePath, nextOrigin, nextDir, nextThroughput, nextPdf);
        }
        const bool want[2] = {wantShadow, wantTrace};
        int slot[2];
        slots.alloc(want, slot, 0, region);
        const int shadowSlot = regionBase + slot[0], traceSlot = regionBase + slot[1];
        if (wantShadow) {
            S->shadow.rayO[shadowSlot] = make_float4(sh.origin.x, sh.origin.y, sh.origin.z, sh.distance);
            S->shadow.rayD[shadowSlot] = make_float4(sh.direction.x, sh.direction.y, sh.direction.z, __uint_as_float(pixelIdx));
            S->shadow.radiance[shadowSlot] = make_float4(sh.radiance.x, sh.radiance.y, sh.radiance.z, 0.0f);
        }
        if (wantTrace) {
            // the path state that goes with the ray: the new one, or — a pass-through — the one the path arrived with
            const float4 tpNext = updatePath ? make_float4(nextThroughput.x, nextThroughput.y, nextThroughput.z, nextPdf) : tpdf;
            // the next logic step's Russian roulette, drawn here (kRaySurvives): its random number is keyed by the pixel — or, with
            // slot-keyed numbers, by the slot the ray goes to — the next bounce and the frame, its probability is the throughput
            // just computed
            uint32_t rng = seed_for(S, (uint32_t)traceSlot, pixelIdx, (uint32_t)bounce + 1u, 0u, frame);
            const bool survives = rng_next(rng) < maxcomp3(mk3(tpNext.x, tpNext.y, tpNext.z));
            out.rayO[traceSlot] = make_float4(nextOrigin.x, nextOrigin.y, nextOrigin.z, __uint_as_float((updatePath ? 0u : kRayPassThrough) | (survives ? kRaySurvives : 0u)));
            out.rayD[traceSlot] = make_float4(nextDir.x, nextDir.y, nextDir.z, __uint_as_float(pixelIdx));
            out.tp[traceSlot] = tpNext;
        }
        }
    };

    for (;;) {
        // ---- find: which of this tile's rays does this kernel shade
        if (per == 4) {
            const int r0 = t * tileRays + 4 * (int)threadIdx.x;
            uint4 w = make_uint4(0u, 0u, 0u, 0u);
            if (r0 + 3 < inRegion) w = *(const NX_G uint4*)(codes + r0);
            else {
                if (r0 < inRegion) w.x = codes[r0];
                if (r0 + 1 < inRegion) w.y = codes[r0 + 1];
                if (r0 + 2 < inRegion) w.z = codes[r0 + 2];
            }
            const bool m0 = (w.x >> kHitCodeShift) == kCode, m1 = (w.y >> kHitCodeShift) == kCode;
            const bool m2 = (w.z >> kHitCodeShift) == kCode, m3 = (w.w >> kHitCodeShift) == kCode;
            const unsigned long long b0 = __ballot(m0), b1 = __ballot(m1), b2 = __ballot(m2), b3 = __ballot(m3);
            const int n0 = __popcll(b0), n1 = __popcll(b1), n2 = __popcll(b2), n3 = __popcll(b3);
            const int total = n0 + n1 + n2 + n3;
            if (total) {
                int base = 0;
                if (lane == 0) base = atomicAdd(sHead, total);  // (an LDS atomic: the ring's head)
                base = __builtin_amdgcn_readfirstlane(base);
                if (m0) sRing[(base + __popcll(b0 & laneLt)) & (kScanRing - 1)] = r0;
                if (m1) sRing[(base + n0 + __popcll(b1 & laneLt)) & (kScanRing - 1)] = r0 + 1;
                if (m2) sRing[(base + n0 + n1 + __popcll(b2 & laneLt)) & (kScanRing - 1)] = r0 + 2;
                if (m3) sRing[(base + n0 + n1 + n2 + __popcll(b3 & laneLt)) & (kScanRing - 1)] = r0 + 3;
            }
        } else {
            const int r0 = t * tileRays + (int)threadIdx.x;
            const bool m0 = r0 < inRegion && (codes[r0] >> kHitCodeShift) == kCode;
            const unsigned long long b0 = __ballot(m0);
            if (b0) {
                int base = 0;
                if (lane == 0) base = atomicAdd(sHead, __popcll(b0));
                base = __builtin_amdgcn_readfirstlane(base);
                if (m0) sRing[(base + __popcll(b0 & laneLt)) & (kScanRing - 1)] = r0;
            }
        }
        __syncthreads();
        head = *sHead;
        // the next tile: by ticket (issued here, needed only after this tile's batches)
        if (threadIdx.x == 0) *sTicket = tiles > staticTiles ? staticTiles + atomicAdd(ticket, 1) : tiles;
        // ---- shade: full batches; what is left over waits for the next tile's finds, or goes last
        while (head - tail >= kShadeBlock) shade_batch(kShadeBlock);
        __syncthreads();  // the ticket is there; and nobody appends to the ring before everybody has read its head
        t = *sTicket;
        if (t >= tiles) break;
    }
    if (head - tail > 0) shade_batch(head - tail);
    // the items this workgroup shaded, for nxhip_read_queue_sizes (the reference's per-type queue sizes, D_QueueSize)
    if (!kMiss && threadIdx.x == 0 && head) atomicAdd(&C->region[region].materialSize[kMiss ? 0 : TYPE][bounce], head);
}

// All material types of a bounce in ONE launch (`typeMask`: bit NX_MAT_* = that type has a kernel in this pass; bit kScanMiss = the
// misses contribute — an environment map or a background that is not black — and are handled here as a fifth type).  The region's
// workgroups start on different types (rank modulo the number of types) and move on to the next type when theirs has no tile
// left, so the types run side by side, the launch ends when the last tile of the last type does, and a bounce costs one material
// launch instead of one per type (the reference: four, PathTracer.cpp:116-120).  A single-bit mask is a per-type launch.
__global__ void __launch_bounds__(kShadeBlock, NX_SHADE_WAVES) shade_scan_kernel(const DeviceState* __restrict__ S, const int bounce, const int typeMask)
{
    __shared__ int sRing[kScanRing];
    __shared__ int sHead, sTicket;
    const int region = (int)(blockIdx.x & (kQueueShards - 1));
    const int inRegion = S->counters->region[region].traceSize[bounce - 1];  // rays of this region after trace(bounce - 1)
    const int rank = (int)(blockIdx.x >> 3), ranks = max(1, (int)(gridDim.x >> 3));
    const int nTypes = __popc((uint32_t)typeMask & 0x1fu);
    if (inRegion <= 0 || nTypes == 0) return;
    // this workgroup's place among those that start on the same type, and how many of them there are
    const int myStart = rank % nTypes, myIndex = rank / nTypes;
    // four rays per thread while that still gives every starter a tile, else one (see shade_scan_type)
    // (... while that gives every starter FOUR tiles: with fewer the launch is as long as its unluckiest workgroup — one frame per pass:
    //  material step 0.50 -> 0.45 ms per frame; the driver's 20-frame pass =)
#ifndef NX_SCAN_TILE_FACTOR
#define NX_SCAN_TILE_FACTOR 4
#endif
    const int per = inRegion >= NX_SCAN_TILE_FACTOR * 4 * kShadeBlock * ((ranks + nTypes - 1) / nTypes) ? 4 : 1;
    const int tiles = (inRegion + per * kShadeBlock - 1) / (per * kShadeBlock);
    if (myIndex >= tiles) return;  // more workgroups than tiles (late bounces): the surplus leaves before any barrier or atomic
    // the types in graph order of the reference (Diffuse, Plastic, Dielectric, Conductor: PathTracer.cpp:116-120), rotated so that
    // this workgroup begins with its own
    constexpr int kOrder[5] = {NX_MAT_DIFFUSE, NX_MAT_PLASTIC, NX_MAT_DIELECTRIC, NX_MAT_CONDUCTOR, kScanMiss};
    for (int step = 0; step < nTypes; step++) {
        const int want = (myStart + step) % nTypes;  // index among the types present
        int type = -1, seen = 0;
#pragma unroll
        for (int k = 0; k < 5; k++)
            if ((typeMask >> kOrder[k]) & 1) { if (seen == want) type = kOrder[k]; seen++; }
        const int starters = (ranks - want + nTypes - 1) / nTypes;  // workgroups whose first type this is: each takes the tile of its index
        const int first = step == 0 ? myIndex : -1;
        switch (type) {
        case NX_MAT_DIFFUSE: shade_scan_type<NX_MAT_DIFFUSE>(S, bounce, region, inRegion, per, first, min(starters, tiles), sRing, &sHead, &sTicket); break;
        case NX_MAT_PLASTIC: shade_scan_type<NX_MAT_PLASTIC>(S, bounce, region, inRegion, per, first, min(starters, tiles), sRing, &sHead, &sTicket); break;
        case NX_MAT_DIELECTRIC: shade_scan_type<NX_MAT_DIELECTRIC>(S, bounce, region, inRegion, per, first, min(starters, tiles), sRing, &sHead, &sTicket); break;
        case NX_MAT_CONDUCTOR: shade_scan_type<NX_MAT_CONDUCTOR>(S, bounce, region, inRegion, per, first, min(starters, tiles), sRing, &sHead, &sTicket); break;
        case kScanMiss: shade_scan_type<kScanMiss>(S, bounce, region, inRegion, per, first, min(starters, tiles), sRing, &sHead, &sTicket); break;
        default: break;
        }
    }
}

// The reference's queue sizes count the hits of a material type whether or not a kernel shades them (its conductor kernel's body is
// commented out, PathTracer.cu:475-478, but the logic kernel still fills that queue): with NX_CONDUCTOR_REFERENCE the SCAN
// pipeline has no conductor kernel to count its items, so this one does (4 B per ray; in the graph only in that mode).
__global__ void __launch_bounds__(kWideBlock) count_scan_kernel(const DeviceState* __restrict__ S, const int bounce, const int type)
{
    Counters* C = S->counters;
    const QueueView in = queue_view(&C->region[0].traceSize[bounce - 1], S->queueShardCap);
    int n = 0;
    for (int index = (int)(blockIdx.x * blockDim.x + threadIdx.x); index < in.total; index += (int)(gridDim.x * blockDim.x))
        n += (S->trace.hitInst[in.slot(index)] >> kHitCodeShift) == (uint32_t)(type + 1) ? 1 : 0;
    n = wave_sum(n);
    if ((threadIdx.x & (kWave - 1)) == 0 && n) atomicAdd(&C->region[0].materialSize[type][bounce], n);
}

// What is left of the logic kernel in the SCAN pipeline when a miss can contribute: the environment (flat colour or map, MIS-
// weighted against the environment sampler) added to the radiance of the paths that missed — PathTracer.cu:152-164.  In the
// graph only when the scene has an environment map or a background that is not black.
__global__ void __launch_bounds__(kWideBlock) miss_scan_kernel(const DeviceState* __restrict__ S, const int bounce)
{
    Counters* C = S->counters;
    const QueueView in = queue_view(&C->region[0].traceSize[bounce - 1], S->queueShardCap);
    const uint32_t frame = S->frame->frameNumber;
    const TraceRays rays = S->trace.rays[(bounce - 1) & 1];
    for (int index = (int)(blockIdx.x * blockDim.x + threadIdx.x); index < in.total; index += (int)(gridDim.x * blockDim.x)) {
        const int at = in.slot(index);
        if ((S->trace.hitInst[at] >> kHitCodeShift) != kHitCodeMiss) continue;
        const float4 dirPix = rays.rayD[at];
        const uint32_t pixelIdx = __float_as_uint(dirPix.w);
        const float4 tp = bounce == 1 ? make_float4(1.0f, 1.0f, 1.0f, 1.0e10f) : rays.tp[at];
        bool miss, survived, needsPrevVertex;
        f3 bg = mk3(0.0f), t = mk3(0.0f);
        uint32_t inst = 0;
        logic_path(S, bounce, frame, (uint32_t)at, pixelIdx, 1e30f, mk3(dirPix.x, dirPix.y, dirPix.z), tp, [&]() { return 0u; }, miss, bg, survived, t, inst, needsPrevVertex);
        if ((__float_as_uint(bg.x) | __float_as_uint(bg.y) | __float_as_uint(bg.z)) != 0u) {  // (as the logic kernel: +0 changes nothing)
            float4 r = bounce == 1 ? make_float4(0, 0, 0, 0) : S->radiance[pixelIdx];
            r.x += bg.x; r.y += bg.y; r.z += bg.z;
            if (bounce == 1) r = make_float4(bg.x, bg.y, bg.z, 0.0f);
            S->radiance[pixelIdx] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Tail kernel: the late bounces of a pass without the graph.  From bounce `firstBounce` on a pass carries a few per cent of
// its rays, yet every bounce still costs a trace level as long as its slowest ray plus a logic and four material launches.
// Here a wave takes 64 paths of the trace queue after trace(firstBounce - 1) and runs each lane's path to its end —
// logic, shade (the material kinds present in the wave, one after the other), the shadow ray of the light sample, the
// continuation ray — with traverse_wave for the rays.  Same functions, same order of radiance additions per pixel, and the
// random numbers are keyed by pixel, bounce and stage: the image is bit-identical to the level-by-level pipeline
// (tests/test_gpu_tail.py).  Pixel-keyed random numbers and the workgroup-aggregated compaction only.
#ifndef NX_TAIL_REFILL_BELOW
#define NX_TAIL_REFILL_BELOW 40
#endif
__global__ void __launch_bounds__(kTraceBlock) tail_kernel(const DeviceState* __restrict__ S, const int firstBounceArg)
{
    // firstBounce | kTraceScanFlag: the pass ran the SCAN pipeline so far — the rays of trace(firstBounce - 1) are in the set of
    // that bounce's parity and the hit records carry codes (which this kernel does not need: it makes the logic step's decisions
    // itself, with the same random numbers)
    const int firstBounce = firstBounceArg & 0xff;
    const TraceRays rays = S->trace.rays[(firstBounceArg & kTraceScanFlag) ? ((firstBounce - 1) & 1) : 0];
    __shared__ unsigned long long ldsStack[kLdsDepth * kTraceBlock];
    Counters* C = S->counters;
    const QueueView in = queue_view(&C->region[0].traceSize[firstBounce - 1], S->queueShardCap);
    const int size = in.total;
    if (size <= 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const unsigned long long laneLt = (1ull << lane) - 1ull;
    lds_u64* const stackLds = (lds_u64*)&ldsStack[threadIdx.x];
    const uint32_t frame = S->frame->frameNumber;

    // per-lane path state; a lane whose path has ended takes the next path of the queue when the wave refills
    bool alive = false, dirty = false;
    int bounce = firstBounce, index = 0;
    uint32_t pixelIdx = 0, inst = 0, tri = 0;
    float hitT = 1e30f, hu = 0.0f, hv = 0.0f;
    f3 dir = mk3(0.0f);
    float4 tp = make_float4(0, 0, 0, 0), ro = make_float4(0, 0, 0, 0), rad = make_float4(0, 0, 0, 0);
    bool exhausted = false;

    for (;;) {
        // ---- refill: paths are handed out one by one (the head counts paths), 64 - alive at a time
        unsigned long long aliveMask = __ballot(alive);
        if (!exhausted && __popcll(aliveMask) < NX_TAIL_REFILL_BELOW) {
            if (!alive && dirty) { S->radiance[pixelIdx] = rad; dirty = false; }
            const unsigned long long needMask = ~aliveMask;
            const int need = (int)__popcll(needMask);
            int base = 0;
            if (lane == 0) base = atomicAdd(&C->tailHead, need);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base + need >= size) exhausted = true;
            const int mine = base + (int)__popcll(needMask & laneLt);
            if (!alive && mine < size) {
                index = mine;
                const int at = in.slot(mine);
                const float4 hit = S->trace.hit[at];
                const float4 dirPix = rays.rayD[at];
                inst = S->trace.hitInst[at] & kHitInstMask;
                pixelIdx = __float_as_uint(dirPix.w);
                dir = mk3(dirPix.x, dirPix.y, dirPix.z);
                hitT = hit.x; hu = hit.y; hv = hit.z; tri = __float_as_uint(hit.w);
                tp = rays.tp[at];
                // the path's previous vertex: the origin of the ray that produced this hit, or, after a pass-through, what
                // the logic step of the pass-through surface kept
                ro = rays.rayO[at];
                if (__float_as_uint(ro.w) & kRayPassThrough) ro = S->rayOrigin[pixelIdx];
                rad = S->radiance[pixelIdx];
                bounce = firstBounce;
                alive = true;
            }
            aliveMask = __ballot(alive);
        }
        if (aliveMask == 0ull) break;

        // ---- logic and shade of every live path at its own bounce
        int type = -1;
        if (alive) {
            bool miss, survived, needsPrevVertex;
            f3 bg = mk3(0.0f), t = mk3(0.0f);
            type = logic_path(S, bounce, frame, (uint32_t)index, pixelIdx, hitT, dir, tp, [&]() { return inst; }, miss, bg, survived, t, inst, needsPrevVertex);
            if (miss) { rad.x += bg.x; rad.y += bg.y; rad.z += bg.z; dirty = true; }
            if (survived) { tp.x = t.x; tp.y = t.y; tp.z = t.z; }
            if (type < 0) alive = false;
        }
        bool wantShadowRay = false, wantTrace = false, updatePath = false;
        ShadowPayload sh;
        sh.origin = mk3(0.0f); sh.direction = mk3(0.0f); sh.radiance = mk3(0.0f); sh.distance = 0.0f;
        f3 nextOrigin = mk3(0.0f), nextDir = mk3(0.0f), nextThroughput = mk3(0.0f);
        float nextPdf = 0.0f;
        const auto prevOrigin = [&]() { return ro; };
        const auto emit = [&](f3 emitted, uint32_t) {
            if (emitted.x != 0.0f || emitted.y != 0.0f || emitted.z != 0.0f) {
                rad.x += emitted.x; rad.y += emitted.y; rad.z += emitted.z;
                dirty = true;
            }
        };
        if (__ballot(alive && type == NX_MAT_DIFFUSE) != 0ull) {
            if (alive && type == NX_MAT_DIFFUSE)
                shade_path<NX_MAT_DIFFUSE>(S, bounce, frame, (uint32_t)index, pixelIdx, hu, hv, tri, inst, dir, tp, prevOrigin, emit, wantShadowRay, sh, wantTrace, updatePath, nextOrigin, nextDir, nextThroughput, nextPdf);
        }
        if (__ballot(alive && type == NX_MAT_PLASTIC) != 0ull) {
            if (alive && type == NX_MAT_PLASTIC)
                shade_path<NX_MAT_PLASTIC>(S, bounce, frame, (uint32_t)index, pixelIdx, hu, hv, tri, inst, dir, tp, prevOrigin, emit, wantShadowRay, sh, wantTrace, updatePath, nextOrigin, nextDir, nextThroughput, nextPdf);
        }
        if (__ballot(alive && type == NX_MAT_DIELECTRIC) != 0ull) {
            if (alive && type == NX_MAT_DIELECTRIC)
                shade_path<NX_MAT_DIELECTRIC>(S, bounce, frame, (uint32_t)index, pixelIdx, hu, hv, tri, inst, dir, tp, prevOrigin, emit, wantShadowRay, sh, wantTrace, updatePath, nextOrigin, nextDir, nextThroughput, nextPdf);
        }
        if (__ballot(alive && type == NX_MAT_CONDUCTOR) != 0ull) {
            if (alive && type == NX_MAT_CONDUCTOR) {
                if (S->conductorMode == NX_CONDUCTOR_EXTENDED)
                    shade_path<NX_MAT_CONDUCTOR>(S, bounce, frame, (uint32_t)index, pixelIdx, hu, hv, tri, inst, dir, tp, prevOrigin, emit, wantShadowRay, sh, wantTrace, updatePath, nextOrigin, nextDir, nextThroughput, nextPdf);
                else alive = false;  // (the reference's graph has no conductor kernel: such a path ends unshaded)
            }
        }
        // ---- the light sample's shadow ray: traced now, added when unoccluded (the shadow launch of this bounce, BVH8Traversal.cuh:515)
        const bool wantShadow = alive && wantShadowRay;
        if (__ballot(wantShadow) != 0ull) {
            float t = wantShadow ? sh.distance : 0.0f, su, sv;
            uint32_t st, si;
            const bool occluded = traverse_wave<true>(S, stackLds, wantShadow, sh.origin, sh.direction, t, su, sv, st, si);
            if (wantShadow && !occluded) {
                rad.x += sh.radiance.x; rad.y += sh.radiance.y; rad.z += sh.radiance.z;
                dirty = true;
            }
        }
        // ---- the continuation ray (a path shaded at bounce == pathLength never asks for one)
        alive = alive && wantTrace;
        if (alive && updatePath) {
            ro = make_float4(nextOrigin.x, nextOrigin.y, nextOrigin.z, 0.0f);
            tp = make_float4(nextThroughput.x, nextThroughput.y, nextThroughput.z, nextPdf);
        }
        if (__ballot(alive) != 0ull) {
            if (alive) dir = nextDir;
            float nt = 1e30f, nu = 0.0f, nv = 0.0f;
            uint32_t ntri = 0xffffffffu, ninst = 0xffffffffu;
            traverse_wave<false>(S, stackLds, alive, nextOrigin, nextDir, nt, nu, nv, ntri, ninst);
            if (alive) { hitT = nt; hu = nu; hv = nv; tri = ntri; inst = ninst; bounce++; }
        }
    }
    if (dirty) S->radiance[pixelIdx] = rad;
}

// ------------------------------------------------------------------------------------------------------
// AccumulateKernel, Tonemap, LinearToGamma, ToColorUInt — PathTracer.cu:37-62, 480-496; Utils/Utils.h:51-54

NXD uint32_t tonemap_rgba8(f3 c)
{
    const float v[3] = {c.x, c.y, c.z};
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float x = v[k] * 0.6f;
        x = clampf((x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f), 0.0f, 1.0f);
        x = (float)nxf_pow((double)x, 0.45454545454);
        x = clampf(x, 0.0f, 1.0f);
        out |= (uint32_t)(uint8_t)(x * 255.0f) << (8 * k);
    }
    return out | (255u << 24);
}

// Running mean over `slices` consecutive frames per pixel, then tonemap.  src == nullptr: this context's own radiance
// (slices = framesPerPass, the pass's frame numbers); otherwise externally gathered radiance laid out
// [slices][sliceStride] whose element k belongs to full-image pixel dstMap[k] (nullptr: k), first frame = firstFrame.
__global__ void __launch_bounds__(kWideBlock) accumulate_kernel(const DeviceState* __restrict__ S, const float4* __restrict__ src, const uint32_t count,
                                                                 const uint32_t slicesIn, const uint32_t sliceStrideIn, const uint32_t firstFrameIn,
                                                                 const uint32_t* __restrict__ dstMap)
{
    const float4* in = src ? src : S->radiance;
    const uint32_t slices = src ? slicesIn : S->framesPerPass;
    const uint32_t sliceStride = src ? sliceStrideIn : S->localCount;
    const uint32_t firstFrame = src ? firstFrameIn : S->frame->frameNumber - (S->framesPerPass - 1u);
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const uint32_t i = dstMap ? dstMap[k] : k;
        float4 a = firstFrame == 1u ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : S->accumulation[i];
        for (uint32_t sl = 0; sl < slices; sl++) {
            const float4 r = in[(size_t)sl * sliceStride + k];
            const uint32_t frame = firstFrame + sl;
            if (frame == 1u) a = make_float4(r.x, r.y, r.z, 0.0f);
            else {
                const float f = (float)frame;
                a.x += (r.x - a.x) / f;
                a.y += (r.y - a.y) / f;
                a.z += (r.z - a.z) / f;
            }
        }
        S->accumulation[i] = a;
        S->rgba8[i] = tonemap_rgba8(mk3(a.x, a.y, a.z));
    }
}

// Multi-GPU image assembly on the root: element k of a rank's local-order accumulation tile belongs to full-image pixel
// dstMap[k] (nullptr: k).  A copy and a tonemap, no arithmetic on the accumulated values.
__global__ void __launch_bounds__(kWideBlock) compose_kernel(const float4* __restrict__ src, const uint32_t count, const uint32_t* __restrict__ dstMap,
                                                              float4* __restrict__ dstAccum, uint32_t* __restrict__ dstRgba8)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const uint32_t i = dstMap ? dstMap[k] : k;
        const float4 a = src[k];
        dstAccum[i] = a;
        if (dstRgba8) dstRgba8[i] = tonemap_rgba8(mk3(a.x, a.y, a.z));
    }
}

// ------------------------------------------------------------------------------------------------------
// Test hooks: the shading functions on plain arrays (nxhip_bsdf_*_batch, nxhip_tex2d_batch)

template <int TYPE>
NXD void bsdf_hook_one(const MatParams& mp, bool sample, const nx_bsdf_query& q, nx_bsdf_result& r)
{
    f3 wo = ld3(q.wo), thr = mk3(0.0f);
    float pdf = 0.0f;
    uint32_t rng = q.rng;
    const bool ok = sample ? Bsdf<TYPE>::sample(mp, ld3(q.wi), rng, wo, thr, pdf) : Bsdf<TYPE>::eval(mp, ld3(q.wi), wo, thr, pdf);
    r.wo[0] = wo.x; r.wo[1] = wo.y; r.wo[2] = wo.z;
    r.pdf = pdf;
    r.throughput[0] = thr.x; r.throughput[1] = thr.y; r.throughput[2] = thr.z;
    r.ok = ok ? 1u : 0u;
    r.rngOut = rng;
    r.pad_[0] = r.pad_[1] = r.pad_[2] = 0u;
}

__global__ void __launch_bounds__(kWideBlock) bsdf_hook_kernel(const nx_material* __restrict__ material, const nx_bsdf_query* __restrict__ q, const uint32_t count,
                                                                const int sample, nx_bsdf_result* __restrict__ out)
{
    const nx_material m = *material;
    const MatParams mp = load_params(m);
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        nx_bsdf_result r;
        switch (m.type) {
        case NX_MAT_DIFFUSE: bsdf_hook_one<NX_MAT_DIFFUSE>(mp, sample != 0, q[k], r); break;
        case NX_MAT_DIELECTRIC: bsdf_hook_one<NX_MAT_DIELECTRIC>(mp, sample != 0, q[k], r); break;
        case NX_MAT_PLASTIC: bsdf_hook_one<NX_MAT_PLASTIC>(mp, sample != 0, q[k], r); break;
        default: bsdf_hook_one<NX_MAT_CONDUCTOR>(mp, sample != 0, q[k], r); break;
        }
        out[k] = r;
    }
}

// include/nexus_fmath.h on arrays (nxhip_fmath_batch, a test hook like the two above)
__global__ void __launch_bounds__(kWideBlock) fmath_hook_kernel(const int op, const double* __restrict__ a, const double* __restrict__ b, const uint32_t count,
                                                                 double* __restrict__ out)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) out[k] = nxf_apply(op, a[k], b ? b[k] : 0.0);
}

// (the texture descriptor comes by value: selecting one of S->diffuseMaps[i] / S->emissiveMaps[i] / S->hdrMap with a
//  uniform three-way branch here was miscompiled by hipcc 7.2 — the third arm left the descriptor pointer unset)
__global__ void __launch_bounds__(kWideBlock) tex2d_hook_kernel(const TextureDev t, const float* __restrict__ srgbLut, const float* __restrict__ uv,
                                                                 const uint32_t count, float4* __restrict__ out)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x)
        out[k] = tex2d(t, srgbLut, uv[2 * k], uv[2 * k + 1]);
}

// ------------------------------------------------------------------------------------------------------

// `items` per thread: kLogicItems (2) unless the caller asks for 1 — what a scene with an environment map gets, whose misses run the
// map lookups (binary64 arc functions) in this kernel: with a second item's state live beside them the 64-register budget spills
// into that path (configs[3]: logic kernel +8 % with two items, -6 % on configs[1])
const void* logic_kernel_ptr(bool ordered, int items)
{
    if (items == 1 || kLogicItems == 1) return ordered ? (const void*)logic_kernel<true, 1> : (const void*)logic_kernel<false, 1>;
    return ordered ? (const void*)logic_kernel<true, kLogicItems> : (const void*)logic_kernel<false, kLogicItems>;
}

const void* shade_kernel_ptr(int type, bool ordered)
{
    switch (type) {
    case NX_MAT_DIFFUSE: return ordered ? (const void*)shade_kernel<NX_MAT_DIFFUSE, true> : (const void*)shade_kernel<NX_MAT_DIFFUSE, false>;
    case NX_MAT_DIELECTRIC: return ordered ? (const void*)shade_kernel<NX_MAT_DIELECTRIC, true> : (const void*)shade_kernel<NX_MAT_DIELECTRIC, false>;
    case NX_MAT_PLASTIC: return ordered ? (const void*)shade_kernel<NX_MAT_PLASTIC, true> : (const void*)shade_kernel<NX_MAT_PLASTIC, false>;
    default: return ordered ? (const void*)shade_kernel<NX_MAT_CONDUCTOR, true> : (const void*)shade_kernel<NX_MAT_CONDUCTOR, false>;
    }
}
const void* shade_scan_kernel_ptr() { return (const void*)shade_scan_kernel; }
const void* miss_scan_kernel_ptr() { return (const void*)miss_scan_kernel; }
const void* count_scan_kernel_ptr() { return (const void*)count_scan_kernel; }
const void* tail_kernel_ptr() { return (const void*)tail_kernel; }
const void* begin_frame_kernel_ptr() { return (const void*)begin_frame_kernel; }
const void* hook_sizes_kernel_ptr() { return (const void*)hook_sizes_kernel; }
const void* generate_kernel_ptr() { return (const void*)generate_kernel; }
const void* accumulate_kernel_ptr() { return (const void*)accumulate_kernel; }
const void* compose_kernel_ptr() { return (const void*)compose_kernel; }
const void* bsdf_hook_kernel_ptr() { return (const void*)bsdf_hook_kernel; }
const void* tex2d_hook_kernel_ptr() { return (const void*)tex2d_hook_kernel; }
const void* fmath_hook_kernel_ptr() { return (const void*)fmath_hook_kernel; }

// the device-side layouts this translation unit was compiled with (nx_device.h layout_stamp; compared by nxhip_create)
uint64_t layout_stamp_wavefront() { return layout_stamp(); }

}  // namespace nxd
