// nexus/Scene.h — mirrors /root/reference/Nexus/src/Scene/Scene.h:17-76, Scene.cpp:10-176: mesh instances, lights
// (an instance is a light iff its material is emissive), dirty tracking, TLAS rebuild on change.
// The device copies are pushed by PathTracer::UpdateDeviceScene through the C-ABI instead of by DeviceVector members.
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

    std::shared_ptr<Camera> GetCamera() const { return m_Camera; }
    void AddMaterial(Material& material) { m_AssetManager.AddMaterial(material); }
    std::vector<Material>& GetMaterials() { return m_AssetManager.GetMaterials(); }
    AssetManager& GetAssetManager() { return m_AssetManager; }
    const AssetManager& GetAssetManager() const { return m_AssetManager; }
    std::shared_ptr<TLAS> GetTLAS() const { return m_Tlas; }
    const RenderSettings& GetRenderSettings() const { return m_RenderSettings; }
    RenderSettings& GetRenderSettings() { return m_RenderSettings; }

    bool IsEmpty() const { return m_MeshInstances.empty(); }
    void Invalidate() { m_Invalid = true; }
    bool IsInvalid() const { return m_Invalid || !m_InvalidMeshInstances.empty() || m_Camera->IsInvalid() || m_AssetManager.IsInvalid(); }

    void Update();     // apply instance transforms / materials, rebuild + convert the TLAS
    void BuildTLAS();
    // Extension: when only transforms / materials of existing instances change, refit the TLAS (O(n)) instead of
    // rebuilding it (the reference always rebuilds, Scene.cpp:29-55).  Off by default.
    void SetTlasRefit(bool enable) { m_TlasRefit = enable; }
    MeshInstance& CreateMeshInstance(uint32_t meshId);
    // Scene.cpp:83-91: load a .glb / .obj (OBJLoader::LoadOBJ), one BVH per mesh, one instance per (node, primitive)
    void CreateMeshInstanceFromFile(const std::string& path, const std::string& fileName);
    std::vector<MeshInstance>& GetMeshInstances() { return m_MeshInstances; }
    const std::vector<BVHInstance>& GetBVHInstances() const { return m_BVHInstances; }
    void AddHDRMap(const Texture& texture);
    const Texture& GetHDRMap() const { return m_HdrMap; }
    void InvalidateMeshInstance(uint32_t instanceId) { m_InvalidMeshInstances.insert(instanceId); }
    const std::vector<Light>& GetLights() const { return m_Lights; }
    size_t AddLight(const Light& light);
    void RemoveLight(size_t index);

    // consumed by PathTracer::UpdateDeviceScene
    mutable bool tlasDirty = true, lightsDirty = true, hdrDirty = false;

private:
    void UpdateInstanceLighting(size_t index);

    std::shared_ptr<Camera> m_Camera;
    std::vector<BVHInstance> m_BVHInstances;
    std::vector<MeshInstance> m_MeshInstances;
    std::vector<Light> m_Lights;
    std::set<uint32_t> m_InvalidMeshInstances;
    std::shared_ptr<TLAS> m_Tlas;
    Texture m_HdrMap;
    AssetManager m_AssetManager;
    RenderSettings m_RenderSettings;
    bool m_Invalid = true;
    bool m_TlasRefit = false;
    size_t m_TlasBuiltFor = 0;  // instance count of the last full TLAS build
};

}  // namespace nexus
