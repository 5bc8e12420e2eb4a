// nxhip_multigpu.hip — the multi-GPU step of the hot path behind the C-ABI: interleaved row-tile split of the frame and ONE
// RCCL gather of the ranks' accumulated tiles per pass (xGMI), composed into the full image on the root.
//
// The reference is single-GPU (no NCCL / MPI anywhere under /root/reference/Nexus/src); SURVEY.md section 8e defines this
// added data-parallel layer: scene replicated, pixels split, no exchange during path tracing, one collective at
// accumulate time.  RCCL is loaded on first use (dlopen): libnexus_amd.so has no link-time dependency on it, single-GPU
// users never touch it, and a process that already carries an RCCL (PyTorch ships one) is not disturbed.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "nx_context.h"

namespace nxd {
const void* compose_kernel_ptr();

namespace {

// the slice of rccl.h this file uses (declared here so that the header is not needed to build)
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
constexpr int kNcclFloat32 = 7;

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Gather)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl& rccl()
{
    static Rccl r;
    if (r.handle || !r.error.empty()) return r;
    // NX_RCCL_LIB, when set, is the only candidate (a caller that names a library does not want another one picked silently)
    const char* forced = std::getenv("NX_RCCL_LIB");
    const char* defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why;
    auto attempt = [&](const char* n) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) {
            const char* e = dlerror();  // ONE call: dlerror() clears the message it returns
            if (why.empty()) why = e ? e : "not found";
        }
    };
    if (forced && *forced) attempt(forced);
    else
        for (const char* n : defaults) {
            attempt(n);
            if (r.handle) break;
        }
    if (!r.handle) {
        r.error = std::string("cannot load RCCL (") + (forced && *forced ? forced : "librccl.so.1") + "): " + (why.empty() ? "not found" : why) + "; set NX_RCCL_LIB";
        return r;
    }
    auto sym = [&](const char* s) { return dlsym(r.handle, s); };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.Gather = reinterpret_cast<decltype(r.Gather)>(sym("ncclGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Gather) {
        r.error = "the RCCL library found lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclGather";
        dlclose(r.handle);
        r.handle = nullptr;
    }
    return r;
}

int fail(int code, const std::string& msg)
{
    set_error(msg);
    return code;
}

int nccl_ok(int rc, const char* what)
{
    if (rc == 0) return NXHIP_OK;
    Rccl& r = rccl();
    return fail(NXHIP_ERR_HIP, std::string(what) + ": RCCL error " + std::to_string(rc) + (r.GetErrorString ? std::string(" (") + r.GetErrorString(rc) + ")" : ""));
}

}  // namespace
// the device-side layouts this translation unit was compiled with (nx_device.h layout_stamp; compared by nxhip_create)
uint64_t layout_stamp_multigpu() { return layout_stamp(); }

}  // namespace nxd

using namespace nxd;

extern "C" {

// Global pixel index of every local pixel of `rank` under the interleaved row-tile split, in the order the context keeps
// its paths: 8x8 pixel tiles over the rank-local rows (a wave's 64 primary rays are a compact block of the image), or plain
// rows.  Same result as nexus_amd/multigpu.py tile_pixel_map + tiled_order (tests/test_multigpu_native.py).
int nxhip_tile_pixel_map(uint32_t width, uint32_t height, int worldSize, int rank, uint32_t tileRows, int tiledOrder, uint32_t* out, uint32_t* localCount)
try {
    if (width == 0 || height == 0 || worldSize < 1 || rank < 0 || rank >= worldSize || tileRows == 0) return fail(NXHIP_ERR_INVALID, "nxhip_tile_pixel_map: bad arguments");
    if (height % (tileRows * (uint32_t)worldSize) != 0) return fail(NXHIP_ERR_INVALID, "nxhip_tile_pixel_map: height must be a multiple of tileRows * worldSize (equal tiles per rank)");
    std::vector<uint32_t> rows;
    for (uint32_t y = 0; y < height; y++)
        if ((y / tileRows) % (uint32_t)worldSize == (uint32_t)rank) rows.push_back(y);
    const uint32_t n = (uint32_t)rows.size() * width;
    if (localCount) *localCount = n;
    if (!out) return NXHIP_OK;
    if (!tiledOrder) {
        uint32_t k = 0;
        for (uint32_t y : rows)
            for (uint32_t x = 0; x < width; x++) out[k++] = y * width + x;
        return NXHIP_OK;
    }
    constexpr uint32_t T = 8;
    const uint32_t tilesX = (width + T - 1) / T;
    uint32_t k = 0;
    for (uint32_t r0 = 0; r0 < rows.size(); r0 += T)        // bands of 8 rank-local rows
        for (uint32_t tx = 0; tx < tilesX; tx++)            // tiles of a band, left to right
            for (uint32_t r = r0; r < std::min<uint32_t>(r0 + T, (uint32_t)rows.size()); r++)
                for (uint32_t x = tx * T; x < std::min(width, (tx + 1) * T); x++) out[k++] = rows[r] * width + x;
    return NXHIP_OK;
} catch (const std::exception& e) {
    return fail(NXHIP_ERR_INVALID, std::string("nxhip_tile_pixel_map: ") + e.what());
}

int nxhip_mgpu_unique_id(void* id128)
{
    if (!id128) return fail(NXHIP_ERR_INVALID, "nxhip_mgpu_unique_id: null destination");
    Rccl& r = rccl();
    if (!r.handle) return fail(NXHIP_ERR_INVALID, r.error);
    ncclUniqueId id;
    const int rc = nccl_ok(r.GetUniqueId(&id), "ncclGetUniqueId");
    if (rc != NXHIP_OK) return rc;
    std::memcpy(id128, id.internal, 128);
    return NXHIP_OK;
}

static int mgpu_setup(nxhip_ctx* c, int worldSize, int rank, uint32_t tileRows, void* comm, bool ownsComm)
try {
    if (worldSize < 1 || rank < 0 || rank >= worldSize) return fail(NXHIP_ERR_INVALID, "nxhip_mgpu: rank / world size out of range");
    std::vector<uint32_t> map;
    uint32_t n = 0;
    int rc = nxhip_tile_pixel_map(c->width, c->height, worldSize, rank, tileRows, 1, nullptr, &n);
    if (rc != NXHIP_OK) return rc;
    map.resize(n);
    rc = nxhip_tile_pixel_map(c->width, c->height, worldSize, rank, tileRows, 1, map.data(), &n);
    if (rc != NXHIP_OK) return rc;
    rc = nxhip_set_pixel_map(c, map.data(), n);
    if (rc != NXHIP_OK) return rc;
    // (the context is marked as split only when everything below has succeeded: a failed setup leaves no half-initialised state)
    DevBuf gathered, maps, fullAccum, fullRgba8;
    if (rank == 0) {
        // root: the gather lands in [world][n] float4; every rank's map, so that the tiles can be scattered into the full image
        const size_t full = (size_t)c->width * c->height;
        if (!gathered.alloc((size_t)worldSize * n * sizeof(float4)) || !maps.alloc((size_t)worldSize * n * 4) ||
            !fullAccum.alloc(full * sizeof(float4)) || !fullRgba8.alloc(full * 4)) return NXHIP_ERR_HIP;
        NX_HIP(hipMemset(fullAccum.p, 0, full * sizeof(float4)));
        NX_HIP(hipMemset(fullRgba8.p, 0, full * 4));
        std::vector<uint32_t> all((size_t)worldSize * n);
        for (int r = 0; r < worldSize; r++) {
            uint32_t m = 0;
            rc = nxhip_tile_pixel_map(c->width, c->height, worldSize, r, tileRows, 1, all.data() + (size_t)r * n, &m);
            if (rc != NXHIP_OK) return rc;
            if (m != n) return fail(NXHIP_ERR_INVALID, "nxhip_mgpu: ranks have unequal tile sizes");
        }
        NX_HIP(hipMemcpy(maps.p, all.data(), all.size() * 4, hipMemcpyHostToDevice));
    }
    c->mgpuGathered = std::move(gathered);
    c->mgpuMaps = std::move(maps);
    c->mgpuFullAccum = std::move(fullAccum);
    c->mgpuFullRgba8 = std::move(fullRgba8);
    c->mgpuWorld = worldSize;
    c->mgpuRank = rank;
    c->mgpuComm = comm;
    c->mgpuOwnsComm = ownsComm;
    c->mgpuTileRows = tileRows;
    c->mgpuPixelSet = c->pixelSetGeneration;  // the pixel set (viewport + map) the gather buffers and maps were built for
    return NXHIP_OK;
} catch (const std::exception& e) {
    return fail(NXHIP_ERR_INVALID, std::string("nxhip_mgpu: ") + e.what());
}

int nxhip_mgpu_init(nxhip_ctx* c, int worldSize, int rank, const void* id128, uint32_t tileRows)
{
    if (!c) return fail(NXHIP_ERR_INVALID, "null context");
    if (!id128) return fail(NXHIP_ERR_INVALID, "nxhip_mgpu_init: null unique id");
    if (c->mgpuComm) return fail(NXHIP_ERR_INVALID, "nxhip_mgpu_init: already initialised (nxhip_mgpu_shutdown first)");
    Rccl& r = rccl();
    if (!r.handle) return fail(NXHIP_ERR_INVALID, r.error);
    NX_HIP(hipSetDevice(c->device));
    ncclUniqueId id;
    std::memcpy(id.internal, id128, 128);
    ncclComm_t comm = nullptr;
    int rc = nccl_ok(r.CommInitRank(&comm, worldSize, id, rank), "ncclCommInitRank");
    if (rc != NXHIP_OK) return rc;
    rc = mgpu_setup(c, worldSize, rank, tileRows, comm, true);
    if (rc != NXHIP_OK) {
        (void)r.CommDestroy(comm);
        c->mgpuComm = nullptr;
    }
    return rc;
}

int nxhip_mgpu_attach(nxhip_ctx* c, void* ncclComm, int worldSize, int rank, uint32_t tileRows)
{
    if (!c) return fail(NXHIP_ERR_INVALID, "null context");
    if (!ncclComm) return fail(NXHIP_ERR_INVALID, "nxhip_mgpu_attach: null communicator");
    if (c->mgpuComm) return fail(NXHIP_ERR_INVALID, "nxhip_mgpu_attach: already initialised (nxhip_mgpu_shutdown first)");
    Rccl& r = rccl();
    if (!r.handle) return fail(NXHIP_ERR_INVALID, r.error);
    NX_HIP(hipSetDevice(c->device));
    return mgpu_setup(c, worldSize, rank, tileRows, ncclComm, false);
}

int nxhip_mgpu_gather(nxhip_ctx* c)
{
    if (!c) return fail(NXHIP_ERR_INVALID, "null context");
    if (!c->mgpuComm) return fail(NXHIP_ERR_INVALID, "nxhip_mgpu_gather: call nxhip_mgpu_init first");
    // nxhip_resize / nxhip_set_pixel_map after the split was set up change localCount and drop the map: the gather buffers,
    // the compose maps and (across ranks) the collective's element count would no longer fit
    if (c->mgpuPixelSet != c->pixelSetGeneration)
        return fail(NXHIP_ERR_INVALID, "nxhip_mgpu_gather: the viewport or pixel map changed after the tile split was set up (nxhip_mgpu_shutdown, then nxhip_mgpu_init again)");
    Rccl& r = rccl();
    NX_HIP(hipSetDevice(c->device));
    const uint32_t n = c->localCount;
    // the one collective of the path: every rank's accumulated tile (float4 per local pixel) to rank 0, on the stream
    // that carries the kernels — render, accumulate, gather and compose need no host synchronisation in between
    const int rc = nccl_ok(r.Gather(c->accumulation.p, c->mgpuRank == 0 ? c->mgpuGathered.p : nullptr, (size_t)n * 4, kNcclFloat32, 0,
                                    static_cast<ncclComm_t>(c->mgpuComm), c->stream), "ncclGather");
    if (rc != NXHIP_OK) return rc;
    if (c->mgpuRank == 0) {
        for (int k = 0; k < c->mgpuWorld; k++) {
            const float4* src = c->mgpuGathered.as<float4>() + (size_t)k * n;
            const uint32_t* map = c->mgpuMaps.as<uint32_t>() + (size_t)k * n;
            float4* dstA = c->mgpuFullAccum.as<float4>();
            uint32_t* dstP = c->mgpuFullRgba8.as<uint32_t>();
            uint32_t count = n;
            void* args[5] = {(void*)&src, (void*)&count, (void*)&map, (void*)&dstA, (void*)&dstP};
            NX_HIP(hipLaunchKernel(compose_kernel_ptr(), dim3(c->wideBlocks), dim3(256), args, 0, c->stream));
        }
    }
    return NXHIP_OK;
}

int nxhip_mgpu_read_rgba8(nxhip_ctx* c, uint32_t* dst)
{
    if (!c || !dst) return fail(NXHIP_ERR_INVALID, "nxhip_mgpu_read_rgba8: null argument");
    if (!c->mgpuComm || c->mgpuRank != 0) return fail(NXHIP_ERR_INVALID, "nxhip_mgpu_read_rgba8: only rank 0 of an initialised tile split holds the full image");
    NX_HIP(hipSetDevice(c->device));
    NX_HIP(hipStreamSynchronize(c->stream));
    NX_HIP(hipMemcpy(dst, c->mgpuFullRgba8.p, (size_t)c->width * c->height * 4, hipMemcpyDeviceToHost));
    return NXHIP_OK;
}

int nxhip_mgpu_read_accumulation(nxhip_ctx* c, float* dst)
try {
    if (!c || !dst) return fail(NXHIP_ERR_INVALID, "nxhip_mgpu_read_accumulation: null argument");
    if (!c->mgpuComm || c->mgpuRank != 0) return fail(NXHIP_ERR_INVALID, "nxhip_mgpu_read_accumulation: only rank 0 of an initialised tile split holds the full image");
    NX_HIP(hipSetDevice(c->device));
    NX_HIP(hipStreamSynchronize(c->stream));
    const size_t full = (size_t)c->width * c->height;
    std::vector<float4> tmp(full);
    NX_HIP(hipMemcpy(tmp.data(), c->mgpuFullAccum.p, full * sizeof(float4), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < full; i++) { dst[3 * i] = tmp[i].x; dst[3 * i + 1] = tmp[i].y; dst[3 * i + 2] = tmp[i].z; }
    return NXHIP_OK;
} catch (const std::exception& e) {
    return fail(NXHIP_ERR_INVALID, std::string("nxhip_mgpu_read_accumulation: ") + e.what());
}

int nxhip_mgpu_shutdown(nxhip_ctx* c)
{
    if (!c) return fail(NXHIP_ERR_INVALID, "null context");
    if (!c->mgpuComm) return NXHIP_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    int rc = NXHIP_OK;
    if (c->mgpuOwnsComm) rc = nccl_ok(rccl().CommDestroy(static_cast<ncclComm_t>(c->mgpuComm)), "ncclCommDestroy");
    c->mgpuComm = nullptr;
    c->mgpuWorld = 1;
    c->mgpuRank = 0;
    c->mgpuGathered.release();
    c->mgpuMaps.release();
    c->mgpuFullAccum.release();
    c->mgpuFullRgba8.release();
    return rc;
}

}  // extern "C"
