// Assets.cpp — AssetManager (see include/nexus/Assets.h); behaviour of /root/reference/Nexus/src/Assets/AssetManager.cpp:12-116.
#include "nexus/Assets.h"

#include "nexus/BVH8Builder.h"

namespace nexus {

void AssetManager::Reset()
{
    m_Materials.clear();
    m_InvalidMaterials.clear();
    m_DiffuseMaps.clear();
    m_EmissiveMaps.clear();
    m_Meshes.clear();
    m_Bvhs.clear();
    uploadedBvhs = 0;
    materialsDirty = texturesDirty = true;
}

int32_t AssetManager::CreateBVH(const std::vector<Triangle>& triangles)
{
    if (m_BlasBuilder) {
        m_Bvhs.push_back(m_BlasBuilder(triangles, m_Bvhs.size()));
        return static_cast<int32_t>(m_Bvhs.size()) - 1;
    }
    BVH8Builder builder(triangles);
    builder.Init();
    m_Bvhs.push_back(builder.Build());
    return static_cast<int32_t>(m_Bvhs.size()) - 1;
}

int32_t AssetManager::CreateBVHs(const std::vector<std::vector<Triangle>>& meshes)
{
    const int32_t first = static_cast<int32_t>(m_Bvhs.size());
    if (m_BlasBatchBuilder && meshes.size() > 1) {
        std::vector<BVH8> built = m_BlasBatchBuilder(meshes, m_Bvhs.size());
        for (BVH8& b : built) m_Bvhs.push_back(std::move(b));
    } else {
        for (const std::vector<Triangle>& m : meshes) CreateBVH(m);
    }
    return first;
}

int32_t AssetManager::AddMesh(Mesh&& mesh)
{
    m_Meshes.push_back(std::move(mesh));
    return static_cast<int32_t>(m_Meshes.size()) - 1;
}

void AssetManager::AddMaterial()
{
    Material material;
    material.diffuse.albedo[0] = material.diffuse.albedo[1] = material.diffuse.albedo[2] = 0.2f;
    AddMaterial(material);
}

int AssetManager::AddMaterial(const Material& material)
{
    m_Materials.push_back(material);
    materialsDirty = true;
    return static_cast<int>(m_Materials.size()) - 1;
}

int AssetManager::AddTexture(const Texture& texture)
{
    if (texture.pixels.empty()) return -1;
    texturesDirty = true;
    if (texture.type == Texture::Type::DIFFUSE) {
        m_DiffuseMaps.push_back(texture);
        return static_cast<int>(m_DiffuseMaps.size()) - 1;
    }
    if (texture.type == Texture::Type::EMISSIVE) {
        m_EmissiveMaps.push_back(texture);
        return static_cast<int>(m_EmissiveMaps.size()) - 1;
    }
    return -1;
}

void AssetManager::ApplyTextureToMaterial(int materialId, int diffuseMapId)
{
    m_Materials[materialId].diffuseMapId = diffuseMapId;
    InvalidateMaterial(materialId);
}

bool AssetManager::SendDataToDevice()
{
    const bool invalid = !m_InvalidMaterials.empty();
    if (invalid) materialsDirty = true;
    m_InvalidMaterials.clear();
    return invalid;
}

}  // namespace nexus
