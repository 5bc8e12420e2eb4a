// nexus/OBJLoader.h — scene ingestion of the kept API surface without Assimp.
// Mirrors /root/reference/Nexus/src/Assets/OBJLoader.h:13-20 (OBJLoader::LoadOBJ(path, filename, scene, assetManager)) and
// what OBJLoader.cpp:8-239 yields through Assimp for the files the reference ships: one mesh + BVH per glTF primitive,
// one instance per (node, primitive) placed with the node's TRS decomposed to Euler degrees, the material heuristics of
// OBJLoader.cpp:71-163, aiProcess_FlipUVs.  Reads binary glTF 2.0 (.glb) and triangulated / polygonal Wavefront .obj.
#pragma once

#include <string>
#include <vector>

#include "Assets.h"
#include "Triangle.h"

namespace nexus {

class Scene;

struct LoadedInstance {
    int mesh = 0, material = 0;
    float3 position = make_float3(0.0f), rotation = make_float3(0.0f), scale = make_float3(1.0f);  // rotation: Euler XYZ, degrees
    std::string name;
};

struct LoadedScene {
    std::vector<std::vector<Triangle>> meshes;
    std::vector<std::string> meshNames;
    std::vector<Material> materials;
    std::vector<LoadedInstance> instances;
    // Images the materials refer to, decoded to RGBA8 (glTF baseColorTexture -> DIFFUSE, emissiveTexture -> EMISSIVE, as
    // Assimp presents them to OBJLoader.cpp:115-160), and per material the index into `textures` (-1: none).  The
    // materials' diffuseMapId / emissiveMapId stay -1 here: ids are handed out by the AssetManager in LoadOBJ.
    std::vector<Texture> textures;
    std::vector<int> materialDiffuseTexture, materialEmissiveTexture;
    std::vector<std::string> warnings;  // images that could not be decoded (the reference prints and carries on, IMGLoader.cpp:24-25)
};

class OBJLoader {
public:
    // Parse a file into meshes / materials / instances; throws std::runtime_error with a message on malformed input.
    static LoadedScene Parse(const std::string& file);
    // Scene::CreateMeshInstanceFromFile's worker (Scene.cpp:83-91): adds the file's materials, builds one BVH8 per mesh,
    // registers the meshes and creates one mesh instance per loaded instance.
    static void LoadOBJ(const std::string& path, const std::string& filename, Scene* scene, AssetManager* assetManager);
};

}  // namespace nexus
