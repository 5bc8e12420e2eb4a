// Scene.cpp — see include/nexus/Scene.h.  Behaviour of /root/reference/Nexus/src/Scene/Scene.cpp:10-176: default camera
// (0,4,14) looking down -z, 60 degree horizontal FOV, focus 5; an instance becomes a MESH_LIGHT when its material has an
// emissive map or intensity * max(emissive) > 0; the TLAS is rebuilt whenever an instance changed.
#include "nexus/Scene.h"

#include "nexus/OBJLoader.h"

namespace nexus {

Scene::Scene(uint32_t width, uint32_t height)
    : m_Camera(std::make_shared<Camera>(make_float3(0.0f, 4.0f, 14.0f), make_float3(0.0f, 0.0f, -1.0f), 60.0f, width, height, 5.0f, 0.0f))
{
}

void Scene::Reset()
{
    m_Invalid = true;
    m_BVHInstances.clear();
    m_InvalidMeshInstances.clear();
    m_MeshInstances.clear();
    m_Lights.clear();
    m_AssetManager.Reset();
    m_Camera->Invalidate();
    m_TlasBuiltFor = 0;
    tlasDirty = lightsDirty = true;
}

void Scene::Update()
{
    m_Camera->SetInvalid(false);
    m_AssetManager.SendDataToDevice();
    if (!m_InvalidMeshInstances.empty()) {
        for (uint32_t i : m_InvalidMeshInstances) {
            MeshInstance& meshInstance = m_MeshInstances[i];
            BVHInstance& inst = m_BVHInstances[meshInstance.bvhInstanceIdx];
            inst.SetBvh(&m_AssetManager.GetBVHs()[m_AssetManager.GetMeshes()[inst.GetBvhIdx()].bvhId]);
            inst.SetTransform(meshInstance.position, meshInstance.rotation, meshInstance.scale);
            if (meshInstance.materialId != -1) {
                inst.AssignMaterial(meshInstance.materialId);
                UpdateInstanceLighting(i);
            }
        }
        if (!m_Tlas) m_Tlas = std::make_shared<TLAS>(m_BVHInstances);
        m_Tlas->SetBVHInstances(m_BVHInstances);
        if (!(m_TlasRefit && m_TlasBuiltFor == m_BVHInstances.size() && m_Tlas->Refit())) {
            m_Tlas->Build();
            m_Tlas->Convert();
            m_TlasBuiltFor = m_BVHInstances.size();
        }
        tlasDirty = true;
        m_InvalidMeshInstances.clear();
    }
    m_Invalid = false;
}

void Scene::BuildTLAS()
{
    m_Tlas = std::make_shared<TLAS>(m_BVHInstances);
    m_Tlas->Build();
    m_Tlas->Convert();
    m_TlasBuiltFor = m_BVHInstances.size();
    tlasDirty = true;
}

MeshInstance& Scene::CreateMeshInstance(uint32_t meshId)
{
    const Mesh& mesh = m_AssetManager.GetMeshes()[meshId];
    const BVH8& bvh8 = m_AssetManager.GetBVHs()[mesh.bvhId];
    m_BVHInstances.push_back(BVHInstance(meshId, &bvh8));  // bvhIdx := meshId, as the reference (Scene.cpp:69)
    m_MeshInstances.push_back(MeshInstance(mesh, static_cast<int>(m_BVHInstances.size()) - 1, mesh.materialId));
    const size_t instanceId = m_MeshInstances.size() - 1;
    UpdateInstanceLighting(instanceId);
    InvalidateMeshInstance(static_cast<uint32_t>(instanceId));
    return m_MeshInstances[instanceId];
}

void Scene::CreateMeshInstanceFromFile(const std::string& path, const std::string& fileName)
{
    OBJLoader::LoadOBJ(path, fileName, this, &m_AssetManager);
    Invalidate();  // the TLAS is rebuilt by the next Update()
}

void Scene::AddHDRMap(const Texture& texture)
{
    m_HdrMap = texture;
    hdrDirty = true;
}

size_t Scene::AddLight(const Light& light)
{
    m_Lights.push_back(light);
    lightsDirty = true;
    return m_Lights.size() - 1;
}

void Scene::RemoveLight(size_t index)
{
    m_Lights.erase(m_Lights.begin() + static_cast<std::ptrdiff_t>(index));
    lightsDirty = true;
}

void Scene::UpdateInstanceLighting(size_t index)
{
    const MeshInstance& meshInstance = m_MeshInstances[index];
    if (meshInstance.materialId == -1) return;
    const Material& material = m_AssetManager.GetMaterials()[meshInstance.materialId];
    const float3 emissive = make_float3(material.emissive);
    for (size_t i = 0; i < m_Lights.size(); i++) {
        const Light& light = m_Lights[i];
        if (light.type == NX_LIGHT_MESH && light.mesh.meshId == index) {
            if (fmaxf(emissive * material.intensity) == 0.0f) RemoveLight(i);
            return;
        }
    }
    if (material.emissiveMapId != -1 || material.intensity * fmaxf(emissive) > 0.0f) {
        Light meshLight;
        meshLight.type = NX_LIGHT_MESH;
        meshLight.mesh.meshId = static_cast<uint32_t>(index);
        AddLight(meshLight);
    }
}

}  // namespace nexus
