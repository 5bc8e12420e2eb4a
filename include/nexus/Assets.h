// nexus/Assets.h — Material, Texture, Mesh, MeshInstance, Light, AssetManager of the kept API surface.
// Mirrors /root/reference/Nexus/src/Assets/{Material.h:5-72, Texture.h:5-27, Mesh.h:9-32, AssetManager.h:13-63},
// Scene/{MeshInstance.h:6-42, Light.h:4-33}.  Host Material / Light share the device POD layout (the reference copies
// them raw: DeviceVector<Material, D_Material> with no ToDevice, AssetManager.h:55).
#pragma once

#include <functional>
#include <set>
#include <string>
#include <vector>

#include "BVH8.h"
#include "Math.h"

namespace nexus {

struct Material : nx_material {
    enum struct Type : int8_t { DIFFUSE = NX_MAT_DIFFUSE, DIELECTRIC = NX_MAT_DIELECTRIC, PLASTIC = NX_MAT_PLASTIC, CONDUCTOR = NX_MAT_CONDUCTOR };
    Material()
    {
        std::memset(static_cast<nx_material*>(this), 0, sizeof(nx_material));
        opacity = 1.0f;
        diffuseMapId = -1;
        emissiveMapId = -1;
        type = NX_MAT_DIFFUSE;
    }
};
static_assert(sizeof(Material) == sizeof(nx_material), "Material must alias nx_material");

struct Light : nx_light {
    enum struct Type : int8_t { POINT_LIGHT = NX_LIGHT_POINT, AREA_LIGHT = NX_LIGHT_AREA, MESH_LIGHT = NX_LIGHT_MESH };
    Light() { std::memset(static_cast<nx_light*>(this), 0, sizeof(nx_light)); }
};
static_assert(sizeof(Light) == sizeof(nx_light), "Light must alias nx_light");

struct Texture {
    enum struct Type { DIFFUSE, ROUGHNESS, METALLIC, EMISSIVE };
    Texture() = default;
    Texture(uint32_t w, uint32_t h, uint32_t c, const unsigned char* d) : width(w), height(h), channels(c), pixels(d, d + static_cast<size_t>(w) * h * 4) {}
    uint32_t width = 0, height = 0, channels = 0;
    std::vector<unsigned char> pixels;  // RGBA8, row 0 first
    Type type = Type::DIFFUSE;
};

struct Mesh {
    Mesh() = default;
    Mesh(const std::string& n, int32_t bId = -1, int32_t mId = -1, float3 p = make_float3(0.0f), float3 r = make_float3(0.0f), float3 s = make_float3(1.0f))
        : bvhId(bId), position(p), rotation(r), scale(s), materialId(mId), name(n)
    {
    }
    int32_t bvhId = -1;
    float3 position = make_float3(0.0f), rotation = make_float3(0.0f), scale = make_float3(1.0f);
    int32_t materialId = -1;
    std::string name;
};

struct MeshInstance {
    MeshInstance() = default;
    MeshInstance(const Mesh& mesh, int bvhInstIdx, int mId = -1)
        : name(mesh.name), bvhInstanceIdx(bvhInstIdx), materialId(mId), rotation(mesh.rotation), scale(mesh.scale), position(mesh.position)
    {
    }
    void SetPosition(float3 p) { position = p; }
    void SetScale(float s) { scale = make_float3(s); }
    void SetScale(float3 s) { scale = s; }
    void SetTransform(float3 p, float3 r, float3 s) { position = p; rotation = r; scale = s; }
    void AssignMaterial(int mId) { materialId = mId; }

    std::string name;
    int bvhInstanceIdx = 0;
    int materialId = -1;
    float3 rotation = make_float3(0.0f), scale = make_float3(1.0f), position = make_float3(0.0f);
};

class AssetManager {
public:
    void Reset();
    int32_t CreateBVH(const std::vector<Triangle>& triangles);  // BVH8Builder(tris).Init().Build(), or the installed builder
    // Extension: who builds a mesh's BVH8.  Default (empty): the host builder, as the reference.  PathTracer::SetDeviceBlasBuild
    // installs one that builds on the GPU (nxhip_build_blas) and returns the tree with deviceBlasId set; `index` is the id the
    // new BVH will have in GetBVHs().
    using BlasBuilder = std::function<BVH8(const std::vector<Triangle>& triangles, size_t index)>;
    void SetBlasBuilder(BlasBuilder builder) { m_BlasBuilder = std::move(builder); }
    // The BVHs of all meshes of a file at once (OBJLoader::LoadOBJ; the reference calls CreateBVH per aiMesh,
    // Assets/OBJLoader.cpp:213-239): returns the id of the first, the others follow.  With a batch builder installed
    // (PathTracer::SetDeviceBlasBuild: nxhip_build_blas_batch, one device build for the whole file) they are built together,
    // otherwise one by one through CreateBVH.  `firstIndex` is the id the first new BVH will have.
    using BlasBatchBuilder = std::function<std::vector<BVH8>(const std::vector<std::vector<Triangle>>& meshes, size_t firstIndex)>;
    void SetBlasBatchBuilder(BlasBatchBuilder builder) { m_BlasBatchBuilder = std::move(builder); }
    int32_t CreateBVHs(const std::vector<std::vector<Triangle>>& meshes);
    int32_t AddMesh(Mesh&& mesh);
    void AddMaterial();
    int AddMaterial(const Material& material);
    std::vector<Material>& GetMaterials() { return m_Materials; }
    const std::vector<Material>& GetMaterials() const { return m_Materials; }
    void InvalidateMaterial(uint32_t index) { m_InvalidMaterials.insert(index); }
    std::vector<BVH8>& GetBVHs() { return m_Bvhs; }
    const std::vector<BVH8>& GetBVHs() const { return m_Bvhs; }
    std::vector<Mesh>& GetMeshes() { return m_Meshes; }
    int AddTexture(const Texture& texture);  // -1 if it has no pixels; id within its kind (diffuse / emissive)
    void ApplyTextureToMaterial(int materialId, int diffuseMapId);
    const std::vector<Texture>& GetDiffuseMaps() const { return m_DiffuseMaps; }
    const std::vector<Texture>& GetEmissiveMaps() const { return m_EmissiveMaps; }
    bool SendDataToDevice();  // clears the invalid-material set; returns whether anything changed
    bool IsInvalid() const { return !m_InvalidMaterials.empty(); }

    // what the device still has to receive (consumed by PathTracer::UpdateDeviceScene)
    bool materialsDirty = true, texturesDirty = true;
    size_t uploadedBvhs = 0;

private:
    std::vector<Material> m_Materials;
    std::set<uint32_t> m_InvalidMaterials;
    std::vector<Texture> m_DiffuseMaps, m_EmissiveMaps;
    std::vector<BVH8> m_Bvhs;
    std::vector<Mesh> m_Meshes;
    BlasBuilder m_BlasBuilder;
    BlasBatchBuilder m_BlasBatchBuilder;
};

}  // namespace nexus
