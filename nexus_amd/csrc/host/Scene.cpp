// Scene.cpp — see include/nexus/Scene.h.  Behaviour of /root/reference/Nexus/src/Scene/Scene.cpp:10-176: default camera
// (0,4,14) looking down -z, 60 degree horizontal FOV, focus 5; an instance becomes a MESH_LIGHT when its material has an
// emissive map or intensity * max(emissive) > 0; the TLAS is rebuilt whenever an instance changed.
#include "nexus/Scene.h"

#include "nexus/IMGLoader.h"
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

// Scene::Update — Scene.cpp:29-55: push pending instance edits into the BVH instances, refresh lights, then bring the TLAS
// up to date (a full agglomerative rebuild as in the reference, or the O(n) refit extension when only existing instances
// changed and it is enabled).
void Scene::Update()
{
    m_Camera->SetInvalid(false);
    m_AssetManager.SendDataToDevice();
    if (m_InvalidMeshInstances.empty()) {
        m_Invalid = false;
        return;
    }

    const std::vector<Mesh>& meshes = m_AssetManager.GetMeshes();
    std::vector<BVH8>& blases = m_AssetManager.GetBVHs();
    std::vector<uint32_t> moved;
    bool onlyTransforms = true;  // nothing but the placement of existing instances changed
    for (const uint32_t id : m_InvalidMeshInstances) {
        const MeshInstance& edited = m_MeshInstances[id];
        BVHInstance& placed = m_BVHInstances[static_cast<size_t>(edited.bvhInstanceIdx)];
        placed.SetBvh(&blases[static_cast<size_t>(meshes[placed.GetBvhIdx()].bvhId)]);  // the vector may have grown since
        placed.SetTransform(edited.position, edited.rotation, edited.scale);
        moved.push_back(static_cast<uint32_t>(edited.bvhInstanceIdx));
        if (edited.materialId == -1) continue;
        if (placed.GetMaterialId() != edited.materialId) onlyTransforms = false;
        placed.AssignMaterial(edited.materialId);
        UpdateInstanceLighting(id);
    }
    m_InvalidMeshInstances.clear();

    const bool sameInstances = m_TlasBuiltFor == m_BVHInstances.size();
    if (m_DeviceTlas) {
        // the tree lives on the device only: same instances moved -> refit there; anything else -> rebuilt there
        if (sameInstances && onlyTransforms && !tlasDirty) movedInstances.insert(movedInstances.end(), moved.begin(), moved.end());
        else {
            tlasDirty = true;
            movedInstances.clear();
        }
        m_TlasBuiltFor = m_BVHInstances.size();
        m_Invalid = false;
        return;
    }
    if (!m_Tlas) m_Tlas = std::make_shared<TLAS>(m_BVHInstances);
    m_Tlas->SetBVHInstances(m_BVHInstances);
    if (m_TlasRefit && sameInstances && m_Tlas->Refit()) {
        // the host copy of the tree is refitted too (O(n)), so that it stays what the device holds; the device is told
        // only which instances moved, unless it has yet to see this tree at all
        if (onlyTransforms && !tlasDirty) movedInstances.insert(movedInstances.end(), moved.begin(), moved.end());
        else tlasDirty = true;
    } else {
        m_Tlas->Build();
        m_Tlas->Convert();
        m_TlasBuiltFor = m_BVHInstances.size();
        tlasDirty = true;
    }
    if (tlasDirty) movedInstances.clear();
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

void Scene::AddHDRMap(const std::string& filePath, const std::string& fileName) { AddHDRMap(IMGLoader::LoadIMG(filePath + fileName)); }

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

namespace {

// Scene.cpp:142-176 in two predicates.  An instance *becomes* a light when its material has an emissive map or
// intensity * max(emissive) > 0; an existing light is *dropped* only when max(emissive * intensity) is exactly 0 (so a
// textured emitter with a black emissive factor keeps its light — the reference's asymmetry, kept).
bool starts_emitting(const Material& m) { return m.emissiveMapId != -1 || m.intensity * fmaxf(make_float3(m.emissive)) > 0.0f; }
bool stops_emitting(const Material& m) { return fmaxf(make_float3(m.emissive) * m.intensity) == 0.0f; }

}  // namespace

void Scene::UpdateInstanceLighting(size_t index)
{
    const int materialId = m_MeshInstances[index].materialId;
    if (materialId == -1) return;
    const Material& material = m_AssetManager.GetMaterials()[static_cast<size_t>(materialId)];

    size_t existing = m_Lights.size();
    for (size_t i = 0; i < m_Lights.size() && existing == m_Lights.size(); i++)
        if (m_Lights[i].type == NX_LIGHT_MESH && m_Lights[i].mesh.meshId == index) existing = i;

    if (existing != m_Lights.size()) {
        if (stops_emitting(material)) RemoveLight(existing);
    } else if (starts_emitting(material)) {
        Light meshLight;
        meshLight.type = NX_LIGHT_MESH;
        meshLight.mesh.meshId = static_cast<uint32_t>(index);
        AddLight(meshLight);
    }
}

}  // namespace nexus
