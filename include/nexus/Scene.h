// nexus/Scene.h — the scene graph of the kept API surface: assets, placed mesh instances, derived lights, TLAS.
// Public methods of /root/reference/Nexus/src/Scene/Scene.h:17-76 (Scene.cpp:10-176).  Differences in kind, not in API:
// there are no device members here — PathTracer::UpdateDeviceScene pushes what changed through the C-ABI and uses the
// `*Dirty` flags below to know what that is — and the HDR map is handed over as pixels, not as a file to decode.
#pragma once

#include <memory>
#include <set>
#include <string>
#include <vector>

#include "Assets.h"
#include "Camera.h"
#include "RenderSettings.h"
#include "TLAS.h"

namespace nexus {

class Scene {
public:
    Scene(uint32_t width, uint32_t height);
    void Reset();

    // ---- building the scene ------------------------------------------------------------------------------------
    // Load a .glb / .obj through OBJLoader::LoadOBJ: one BVH8 per mesh, one instance per (node, primitive) — Scene.cpp:83-91
    void CreateMeshInstanceFromFile(const std::string& path, const std::string& fileName);
    // Place mesh `meshId` (AssetManager::AddMesh) with the mesh's own transform and material; a light is derived if its
    // material emits.  The returned reference is valid until the next instance is created.
    MeshInstance& CreateMeshInstance(uint32_t meshId);
    void AddMaterial(Material& material) { m_AssetManager.AddMaterial(material); }
    void AddHDRMap(const Texture& texture);
    void AddHDRMap(const std::string& filePath, const std::string& fileName);  // Scene.cpp:93-97: IMGLoader::LoadIMG (.hdr or .png)
    size_t AddLight(const Light& light);
    void RemoveLight(size_t index);

    // ---- editing: change a MeshInstance, then tell the scene which one --------------------------------------------
    std::vector<MeshInstance>& GetMeshInstances() { return m_MeshInstances; }
    void InvalidateMeshInstance(uint32_t instanceId) { m_InvalidMeshInstances.insert(instanceId); }
    void Invalidate() { m_Invalid = true; }
    // Extension, off by default: when only existing instances changed, refit the TLAS (O(n)) in Update() instead of the
    // reference's full agglomerative rebuild (Scene.cpp:29-55).
    void SetTlasRefit(bool enable) { m_TlasRefit = enable; }
    // Extension, off by default: leave the TLAS to the device.  Update() then builds no tree on the host when instances were
    // added or removed — PathTracer::UpdateDeviceScene has the device build it from the instances' world boxes
    // (nxhip_rebuild_tlas: the binned-SAH builder over the instance boxes, milliseconds where the reference's agglomerative
    // clustering takes seconds, and a tree rays enter fewer instances of) — and
    // moved instances are refitted on the device as with SetTlasRefit.  GetTLAS() holds no tree in this mode.
    void SetDeviceTlasBuild(bool enable) { m_DeviceTlas = enable; tlasDirty = true; }
    bool UsesDeviceTlasBuild() const { return m_DeviceTlas; }

    // ---- per frame -----------------------------------------------------------------------------------------------
    bool IsInvalid() const { return m_Invalid || !m_InvalidMeshInstances.empty() || m_Camera->IsInvalid() || m_AssetManager.IsInvalid(); }
    void Update();     // apply pending instance edits, refresh lights, bring the TLAS up to date
    void BuildTLAS();  // unconditional rebuild + BVH8 conversion

    // ---- read access ---------------------------------------------------------------------------------------------
    bool IsEmpty() const { return m_MeshInstances.empty(); }
    std::shared_ptr<Camera> GetCamera() const { return m_Camera; }
    std::shared_ptr<TLAS> GetTLAS() const { return m_Tlas; }
    AssetManager& GetAssetManager() { return m_AssetManager; }
    const AssetManager& GetAssetManager() const { return m_AssetManager; }
    std::vector<Material>& GetMaterials() { return m_AssetManager.GetMaterials(); }
    RenderSettings& GetRenderSettings() { return m_RenderSettings; }
    const RenderSettings& GetRenderSettings() const { return m_RenderSettings; }
    const std::vector<BVHInstance>& GetBVHInstances() const { return m_BVHInstances; }
    const std::vector<Light>& GetLights() const { return m_Lights; }
    const Texture& GetHDRMap() const { return m_HdrMap; }

    // what the device has not seen yet (set here, cleared by PathTracer::UpdateDeviceScene)
    mutable bool tlasDirty = true, lightsDirty = true, hdrDirty = false;
    // With SetTlasRefit(true): BVH-instance ids whose transform (and nothing else) changed since the device last saw the TLAS.
    // PathTracer::UpdateDeviceScene hands them to nxhip_set_instance_transforms — inverse, bounds, traversal records and the
    // TLAS refit happen in HBM — instead of uploading the tree again.
    mutable std::vector<uint32_t> movedInstances;

private:
    // derive / drop the MESH_LIGHT of instance `index` from its material (Scene.cpp:142-176)
    void UpdateInstanceLighting(size_t index);

    AssetManager m_AssetManager;
    RenderSettings m_RenderSettings;
    std::shared_ptr<Camera> m_Camera;
    Texture m_HdrMap;

    std::vector<MeshInstance> m_MeshInstances;  // what the user edits
    std::vector<BVHInstance> m_BVHInstances;    // what the TLAS and the device see, one per mesh instance
    std::vector<Light> m_Lights;
    std::shared_ptr<TLAS> m_Tlas;

    std::set<uint32_t> m_InvalidMeshInstances;
    bool m_Invalid = true;
    bool m_TlasRefit = false;
    bool m_DeviceTlas = false;
    size_t m_TlasBuiltFor = 0;  // instance count of the last full TLAS build
};

}  // namespace nexus
