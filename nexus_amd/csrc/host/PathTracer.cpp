// PathTracer.cpp — see include/nexus/PathTracer.h.  Where the reference binds CUDA symbols, allocates queues and replays a
// CUDA graph (/root/reference/Nexus/src/Renderer/PathTracer.cpp:5-317), this class forwards to the C-ABI device layer.
#include "nexus/PathTracer.h"

#include <cstring>
#include <stdexcept>
#include <string>

namespace nexus {

namespace {
void Check(int rc, const char* what)
{
    if (rc != NXHIP_OK) throw std::runtime_error(std::string(what) + ": " + nxhip_last_error());
}
}  // namespace

PathTracer::PathTracer(uint32_t width, uint32_t height, int device) : m_ViewportWidth(width), m_ViewportHeight(height)
{
    Check(nxhip_create(device, width, height, nullptr, &m_Ctx), "nxhip_create");
}

PathTracer::~PathTracer() { nxhip_destroy(m_Ctx); }

void PathTracer::Reset()
{
    // A tile split is set up for one viewport: its tiles, gather buffers and (across ranks) the collective's element count no
    // longer fit another.  It is shut down here; EnableTileSplit sets it up again for the new size, on every rank.
    if (m_TileSplit) {
        Check(nxhip_mgpu_shutdown(m_Ctx), "nxhip_mgpu_shutdown");
        m_TileSplit = false;
        m_Rank = 0;
    }
    Check(nxhip_resize(m_Ctx, m_ViewportWidth, m_ViewportHeight), "nxhip_resize");
    m_FrameNumber = 0;
}

void PathTracer::FreeDeviceBuffers() { Check(nxhip_release_queues(m_Ctx), "nxhip_release_queues"); }

void PathTracer::ResetFrameNumber()
{
    m_FrameNumber = 0;
    Check(nxhip_reset_frame_number(m_Ctx), "nxhip_reset_frame_number");
}

void PathTracer::OnResize(uint32_t width, uint32_t height)
{
    if ((m_ViewportWidth != width || m_ViewportHeight != height) && width != 0 && height != 0) {
        m_ViewportWidth = width;
        m_ViewportHeight = height;
        Reset();
    }
}

void PathTracer::SetModes(int rngMode, int compactMode, int conductorMode) { Check(nxhip_set_modes(m_Ctx, rngMode, compactMode, conductorMode), "nxhip_set_modes"); }

// BLAS k of the device == BVH k of the asset manager (instances carry the *mesh* id, whose bvhId the reference assumes equal:
// Scene.cpp:69).  A scene that starts over (AssetManager::Reset) starts the device's list over too.
void PathTracer::UploadPendingBlas(AssetManager& assets)
{
    std::vector<BVH8>& bvhs = assets.GetBVHs();
    if (assets.uploadedBvhs == 0 && !bvhs.empty()) Check(nxhip_clear_blas(m_Ctx), "nxhip_clear_blas");
    for (size_t i = assets.uploadedBvhs; i < bvhs.size(); i++) {
        std::vector<nx_triangle> tris(bvhs[i].triangles.size());
        for (size_t t = 0; t < tris.size(); t++) tris[t] = Triangle::ToDevice(bvhs[i].triangles[t]);
        int32_t id = -1;
        Check(nxhip_upload_blas(m_Ctx, bvhs[i].nodes.data(), static_cast<uint32_t>(bvhs[i].nodes.size()), tris.data(), static_cast<uint32_t>(tris.size()),
                                bvhs[i].triangleIdx.data(), &id), "nxhip_upload_blas");
        if (id != static_cast<int32_t>(i)) throw std::runtime_error("PathTracer: the device's BLAS ids and the asset manager's BVH ids have diverged");
        bvhs[i].deviceBlasId = id;
        assets.uploadedBvhs = i + 1;
    }
}

void PathTracer::SetDeviceBlasBuild(Scene& scene, bool enable)
{
    AssetManager& assets = scene.GetAssetManager();
    if (!enable) {
        assets.SetBlasBuilder(nullptr);
        assets.SetBlasBatchBuilder(nullptr);
        return;
    }
    // all meshes of a file in one device build (nxhip_build_blas_batch), the trees back in one transfer
    assets.SetBlasBatchBuilder([this, &assets](const std::vector<std::vector<Triangle>>& meshes, size_t firstIndex) {
        UploadPendingBlas(assets);
        if (assets.uploadedBvhs == 0) Check(nxhip_clear_blas(m_Ctx), "nxhip_clear_blas");
        std::vector<std::vector<nx_triangle>> dev(meshes.size());
        std::vector<const nx_triangle*> ptrs(meshes.size());
        std::vector<uint32_t> counts(meshes.size());
        size_t triTotal = 0;
        for (size_t m = 0; m < meshes.size(); m++) {
            dev[m].resize(meshes[m].size());
            for (size_t t = 0; t < dev[m].size(); t++) dev[m][t] = Triangle::ToDevice(meshes[m][t]);
            ptrs[m] = dev[m].data();
            counts[m] = static_cast<uint32_t>(dev[m].size());
            triTotal += dev[m].size();
        }
        std::vector<int32_t> ids(meshes.size(), -1);
        Check(nxhip_build_blas_batch(m_Ctx, ptrs.data(), counts.data(), static_cast<uint32_t>(meshes.size()), ids.data()), "nxhip_build_blas_batch");
        for (size_t m = 0; m < meshes.size(); m++)
            if (ids[m] != static_cast<int32_t>(firstIndex + m)) throw std::runtime_error("PathTracer: the device's BLAS ids and the asset manager's BVH ids have diverged");
        std::vector<uint32_t> nodeCounts(meshes.size(), 0u);
        Check(nxhip_read_blas_batch(m_Ctx, ids[0], static_cast<uint32_t>(meshes.size()), nullptr, 0, nodeCounts.data(), nullptr, 0), "nxhip_read_blas_batch");
        size_t nodeTotal = 0;
        for (uint32_t n : nodeCounts) nodeTotal += n;
        std::vector<BVH8Node> nodes(nodeTotal);
        std::vector<uint32_t> idx(triTotal);
        Check(nxhip_read_blas_batch(m_Ctx, ids[0], static_cast<uint32_t>(meshes.size()), nodes.data(), static_cast<uint32_t>(nodeTotal), nodeCounts.data(), idx.data(),
                                    static_cast<uint32_t>(triTotal)), "nxhip_read_blas_batch");
        std::vector<BVH8> out;
        out.reserve(meshes.size());
        size_t nodeAt = 0, triAt = 0;
        for (size_t m = 0; m < meshes.size(); m++) {
            BVH8 bvh(meshes[m]);
            bvh.nodes.assign(nodes.begin() + static_cast<std::ptrdiff_t>(nodeAt), nodes.begin() + static_cast<std::ptrdiff_t>(nodeAt + nodeCounts[m]));
            bvh.triangleIdx.assign(idx.begin() + static_cast<std::ptrdiff_t>(triAt), idx.begin() + static_cast<std::ptrdiff_t>(triAt + counts[m]));
            bvh.deviceBlasId = ids[m];
            nodeAt += nodeCounts[m];
            triAt += counts[m];
            out.push_back(std::move(bvh));
        }
        assets.uploadedBvhs = firstIndex + meshes.size();
        return out;
    });
    assets.SetBlasBuilder([this, &assets](const std::vector<Triangle>& triangles, size_t index) {
        // everything created before is on the device first, so that this tree's id there is its id here
        UploadPendingBlas(assets);
        if (assets.uploadedBvhs == 0) Check(nxhip_clear_blas(m_Ctx), "nxhip_clear_blas");
        BVH8 bvh(triangles);
        std::vector<nx_triangle> tris(triangles.size());
        for (size_t t = 0; t < tris.size(); t++) tris[t] = Triangle::ToDevice(triangles[t]);
        int32_t id = -1;
        Check(nxhip_build_blas(m_Ctx, tris.data(), static_cast<uint32_t>(tris.size()), &id), "nxhip_build_blas");
        if (id != static_cast<int32_t>(index)) throw std::runtime_error("PathTracer: the device's BLAS ids and the asset manager's BVH ids have diverged");
        uint32_t nodeCount = 0;
        Check(nxhip_read_blas(m_Ctx, id, nullptr, 0, nullptr, 0, &nodeCount), "nxhip_read_blas");
        bvh.nodes.resize(nodeCount);
        Check(nxhip_read_blas(m_Ctx, id, bvh.nodes.data(), nodeCount, bvh.triangleIdx.data(), static_cast<uint32_t>(bvh.triangleIdx.size()), &nodeCount), "nxhip_read_blas");
        bvh.deviceBlasId = id;
        assets.uploadedBvhs = index + 1;
        return bvh;
    });
}

void PathTracer::UpdateDeviceScene(const Scene& scene)
{
    const AssetManager& assets = scene.GetAssetManager();
    AssetManager& mutableAssets = const_cast<AssetManager&>(assets);
    UploadPendingBlas(mutableAssets);
    if (assets.texturesDirty) {
        Check(nxhip_clear_textures(m_Ctx), "nxhip_clear_textures");
        for (const Texture& t : assets.GetDiffuseMaps()) Check(nxhip_upload_texture(m_Ctx, 0, t.pixels.data(), t.width, t.height, nullptr), "nxhip_upload_texture");
        for (const Texture& t : assets.GetEmissiveMaps()) Check(nxhip_upload_texture(m_Ctx, 1, t.pixels.data(), t.width, t.height, nullptr), "nxhip_upload_texture");
        if (!scene.GetHDRMap().pixels.empty()) Check(nxhip_upload_texture(m_Ctx, 2, scene.GetHDRMap().pixels.data(), scene.GetHDRMap().width, scene.GetHDRMap().height, nullptr), "nxhip_upload_texture");
        mutableAssets.texturesDirty = false;
        scene.hdrDirty = false;
    } else if (scene.hdrDirty) {
        Check(nxhip_upload_texture(m_Ctx, 2, scene.GetHDRMap().pixels.data(), scene.GetHDRMap().width, scene.GetHDRMap().height, nullptr), "nxhip_upload_texture");
        scene.hdrDirty = false;
    }
    if (assets.materialsDirty && !assets.GetMaterials().empty()) {
        Check(nxhip_set_materials(m_Ctx, assets.GetMaterials().data(), static_cast<uint32_t>(assets.GetMaterials().size())), "nxhip_set_materials");
        mutableAssets.materialsDirty = false;
    }
    if (scene.tlasDirty && scene.UsesDeviceTlasBuild() && !scene.GetBVHInstances().empty()) {
        // the TLAS is built where it is used: instances (with the world bounds SetTransform gave them) go up, the tree never
        // exists on the host
        const std::vector<BVHInstance>& placed = scene.GetBVHInstances();
        std::vector<nx_bvh_instance> inst(placed.size());
        for (size_t i = 0; i < inst.size(); i++) inst[i] = BVHInstance::ToDevice(placed[i]);
        Check(nxhip_rebuild_tlas(m_Ctx, inst.data(), static_cast<uint32_t>(inst.size())), "nxhip_rebuild_tlas");
        scene.tlasDirty = false;
        scene.movedInstances.clear();
    } else if (scene.tlasDirty && scene.GetTLAS() && !scene.GetTLAS()->bvh8.nodes.empty()) {
        const TLAS& tlas = *scene.GetTLAS();
        std::vector<nx_bvh_instance> inst(tlas.bvhInstances.size());
        for (size_t i = 0; i < inst.size(); i++) inst[i] = BVHInstance::ToDevice(tlas.bvhInstances[i]);
        Check(nxhip_set_tlas(m_Ctx, tlas.bvh8.nodes.data(), static_cast<uint32_t>(tlas.bvh8.nodes.size()), tlas.bvh8.triangleIdx.data(), inst.data(),
                             static_cast<uint32_t>(inst.size())), "nxhip_set_tlas");
        scene.tlasDirty = false;
        scene.movedInstances.clear();
    } else if (!scene.movedInstances.empty()) {
        // dynamic transforms: the device recomputes inverse, bounds and traversal records of the moved instances and refits
        // its TLAS in place (nx_refit.hip); the tree is not uploaded again
        const std::vector<BVHInstance>& inst = scene.GetBVHInstances();
        std::vector<float> matrices(scene.movedInstances.size() * 16);
        for (size_t k = 0; k < scene.movedInstances.size(); k++) {
            const Mat4 m = inst[scene.movedInstances[k]].GetTransform();
            std::memcpy(&matrices[16 * k], m.cell, 64);
        }
        Check(nxhip_set_instance_transforms(m_Ctx, scene.movedInstances.data(), matrices.data(), static_cast<uint32_t>(scene.movedInstances.size())),
              "nxhip_set_instance_transforms");
        scene.movedInstances.clear();
    }
    if (scene.lightsDirty) {
        Check(nxhip_set_lights(m_Ctx, scene.GetLights().data(), static_cast<uint32_t>(scene.GetLights().size())), "nxhip_set_lights");
        scene.lightsDirty = false;
    }
    const nx_camera cam = Camera::ToDevice(*scene.GetCamera());
    Check(nxhip_set_camera(m_Ctx, &cam), "nxhip_set_camera");
    Check(nxhip_set_render_settings(m_Ctx, reinterpret_cast<const nx_render_settings*>(&scene.GetRenderSettings())), "nxhip_set_render_settings");
}

void PathTracer::SetFramesPerPass(uint32_t frames)
{
    Check(nxhip_set_frames_per_pass(m_Ctx, frames), "nxhip_set_frames_per_pass");
    m_FramesPerPass = frames;
}
void PathTracer::SetPassesInFlight(uint32_t passes) { Check(nxhip_set_passes_in_flight(m_Ctx, passes), "nxhip_set_passes_in_flight"); }
void PathTracer::SetTailBounce(uint32_t bounce) { Check(nxhip_set_tail_bounce(m_Ctx, bounce), "nxhip_set_tail_bounce"); }
void PathTracer::SetEntryPoints(bool on) { Check(nxhip_set_entry_points(m_Ctx, on ? 1 : 0), "nxhip_set_entry_points"); }
void PathTracer::SetPixelOrder(int order)
{
    Check(nxhip_set_pixel_order(m_Ctx, order), "nxhip_set_pixel_order");
    m_FrameNumber = 0;  // (a new pixel set starts the accumulation over, as OnResize does)
}

void PathTracer::Render(const Scene&)
{
    m_FrameNumber += m_FramesPerPass;
    Check(nxhip_render_frame(m_Ctx), "nxhip_render_frame");
    Check(nxhip_accumulate(m_Ctx), "nxhip_accumulate");
    if (m_TileSplit) Check(nxhip_mgpu_gather(m_Ctx), "nxhip_mgpu_gather");
}

void PathTracer::CreateTileSplitId(void* id128) { Check(nxhip_mgpu_unique_id(id128), "nxhip_mgpu_unique_id"); }

void PathTracer::EnableTileSplit(int worldSize, int rank, const void* id128, uint32_t tileRows)
{
    Check(nxhip_mgpu_init(m_Ctx, worldSize, rank, id128, tileRows), "nxhip_mgpu_init");
    m_TileSplit = true;
    m_Rank = rank;
    m_FrameNumber = 0;
}

void PathTracer::DisableTileSplit()
{
    if (!m_TileSplit) return;
    Check(nxhip_mgpu_shutdown(m_Ctx), "nxhip_mgpu_shutdown");
    Check(nxhip_set_pixel_map(m_Ctx, nullptr, 0), "nxhip_set_pixel_map");
    m_TileSplit = false;
    m_Rank = 0;
    m_FrameNumber = 0;
}

void PathTracer::SetPixelQuery(uint32_t x, uint32_t y)
{
    Check(nxhip_set_pixel_query(m_Ctx, x, y), "nxhip_set_pixel_query");
    m_PixelQueryPending = true;
}

int32_t PathTracer::GetSelectedInstance()
{
    int32_t idx = -1;
    Check(nxhip_get_selected_instance(m_Ctx, &idx), "nxhip_get_selected_instance");
    m_PixelQueryPending = false;
    return idx;
}

const std::vector<uint32_t>& PathTracer::GetPixelBuffer()
{
    m_Pixels.resize(static_cast<size_t>(m_ViewportWidth) * m_ViewportHeight);
    if (m_TileSplit) Check(nxhip_mgpu_read_rgba8(m_Ctx, m_Pixels.data()), "nxhip_mgpu_read_rgba8");  // rank 0: the assembled frame
    else Check(nxhip_read_rgba8(m_Ctx, m_Pixels.data()), "nxhip_read_rgba8");
    return m_Pixels;
}

}  // namespace nexus
