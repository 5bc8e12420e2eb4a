// nexus/PathTracer.h — host driver of the hot path with the reference's interface
// (/root/reference/Nexus/src/Renderer/PathTracer.h:9-70, PathTracer.cpp:5-317).  Every method maps onto the C-ABI device
// layer (include/nexus_hip.h) instead of CUDA symbols, kernels and a CUDA graph.
#pragma once

#include <cstdint>
#include <vector>

#include "../nexus_hip.h"
#include "Scene.h"

namespace nexus {

class PathTracer {
public:
    PathTracer(uint32_t width, uint32_t height, int device = 0);
    ~PathTracer();
    PathTracer(const PathTracer&) = delete;
    PathTracer& operator=(const PathTracer&) = delete;

    void Reset();
    // the queue / path-state buffers go back to the device allocator (the reference frees them in Reset and on resize,
    // PathTracer.cpp:31, :300); the next Render() brings them back
    void FreeDeviceBuffers();
    void ResetFrameNumber();
    void Render(const Scene& scene);  // one frame: generate, trace, pathLength x (logic, shade, trace, shadow), accumulate
    void OnResize(uint32_t width, uint32_t height);
    void UpdateDeviceScene(const Scene& scene);
    // Extension, off by default: meshes added to `scene` from now on get their BVH8 from the device builder (nxhip_build_blas:
    // the reference's binned-SAH rule and SAH-DP collapse run in HBM, a tenth of the host builder's time) instead of
    // AssetManager::CreateBVH's host build; the nodes come back once, so BVH8::nodes / triangleIdx hold the tree as they do for
    // a host-built one.  The scene must not outlive this PathTracer while the switch is on.
    void SetDeviceBlasBuild(Scene& scene, bool enable);
    void SetPixelQuery(uint32_t x, uint32_t y);
    int32_t GetSelectedInstance();
    uint32_t GetFrameNumber() const { return m_FrameNumber; }
    // The reference hands out a GL pixel buffer; headless here: the RGBA8 image, read back on demand.
    const std::vector<uint32_t>& GetPixelBuffer();
    // Extensions (see nexus_pod.h)
    void SetModes(int rngMode, int compactMode, int conductorMode);
    // Render() renders `frames` consecutive frames per call (bit-identical to as many single calls), lets `passes`
    // consecutive calls overlap on the GPU, and finishes the late bounces of small passes in one launch: the three
    // small-pass measures of the device layer (include/nexus_hip.h), none of which changes the image.
    void SetFramesPerPass(uint32_t frames);
    void SetPassesInFlight(uint32_t passes);
    void SetTailBounce(uint32_t bounce);
    // Primary rays start from the traversal state the first node steps of their run of 64 paths provably share instead of the TLAS
    // root (nxhip_set_entry_points: hit records unchanged, pinhole cameras only).  Off by default.
    void SetEntryPoints(bool on);
    // The order of the frame's paths: NXHIP_ORDER_ROWS (the reference's, default) or NXHIP_ORDER_TILES (8 x 8 pixel tiles: the rays a
    // wave fetches together are a compact block of the image; kept across OnResize).  nxhip_set_pixel_order.
    void SetPixelOrder(int order);
    // Multi-GPU extension (no counterpart in the reference): one PathTracer per GPU, each renders and
    // accumulates the interleaved row tiles of its rank; Render() then ends with ONE RCCL gather of the accumulated tiles to
    // rank 0, whose GetPixelBuffer() returns the full frame.  `id128`: the 128 bytes rank 0 obtained from
    // CreateTileSplitId() and handed to every rank.  Call after construction / OnResize, before the first Render.
    static void CreateTileSplitId(void* id128);
    void EnableTileSplit(int worldSize, int rank, const void* id128, uint32_t tileRows = 5);
    void DisableTileSplit();
    bool IsTileSplitRoot() const { return m_TileSplit && m_Rank == 0; }
    nxhip_ctx* GetDeviceContext() const { return m_Ctx; }
    uint32_t GetWidth() const { return m_ViewportWidth; }
    uint32_t GetHeight() const { return m_ViewportHeight; }

private:
    void UploadPendingBlas(AssetManager& assets);
    nxhip_ctx* m_Ctx = nullptr;
    uint32_t m_FrameNumber = 0;
    uint32_t m_FramesPerPass = 1;
    uint32_t m_ViewportWidth = 0, m_ViewportHeight = 0;
    std::vector<uint32_t> m_Pixels;
    bool m_PixelQueryPending = false;
    bool m_TileSplit = false;
    int m_Rank = 0;
};

}  // namespace nexus
