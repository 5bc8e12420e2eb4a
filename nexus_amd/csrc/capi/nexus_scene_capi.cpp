// nexus_scene_capi.cpp — flat C wrappers over nexus::Scene / nexus::PathTracer (see include/nexus_host.h).
#include <cstring>
#include <exception>
#include <memory>
#include <stdexcept>
#include <string>

#include "nexus/IMGLoader.h"
#include "nexus/OBJLoader.h"
#include "nexus/Renderer.h"
#include "nexus/PathTracer.h"
#include "nexus/Scene.h"
#include "nexus_host.h"

using namespace nexus;

struct nxs_scene {
    Scene scene;
    nxs_scene(uint32_t w, uint32_t h) : scene(w, h) {}
};
struct nxh_loaded_scene {
    LoadedScene ls;
};
struct nxs_pathtracer {
    PathTracer pt;
    nxs_pathtracer(uint32_t w, uint32_t h, int dev) : pt(w, h, dev) {}
};
struct nxs_renderer {
    Renderer r;
    nxs_renderer(uint32_t w, uint32_t h, Scene* s, int dev) : r(w, h, s, dev) {}
};

namespace {
thread_local std::string g_err;
template <typename F> int guarded(F&& f)
{
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return 1;
    }
}
}  // namespace

extern "C" {

const char* nxs_last_error(void) { return g_err.c_str(); }

int nxs_scene_create(uint32_t width, uint32_t height, nxs_scene** out)
{
    return guarded([&] { *out = new nxs_scene(width, height); });
}
void nxs_scene_destroy(nxs_scene* s) { delete s; }

int nxs_scene_add_material(nxs_scene* s, const nx_material* m, int32_t* materialId)
{
    return guarded([&] {
        Material mat;
        std::memcpy(static_cast<nx_material*>(&mat), m, sizeof(nx_material));
        const int id = s->scene.GetAssetManager().AddMaterial(mat);
        if (materialId) *materialId = id;
    });
}

int nxs_scene_add_texture(nxs_scene* s, int kind, const uint8_t* rgba8, uint32_t w, uint32_t h, int32_t* texId)
{
    return guarded([&] {
        Texture t(w, h, 4, rgba8);
        t.type = kind == 1 ? Texture::Type::EMISSIVE : Texture::Type::DIFFUSE;
        const int id = s->scene.GetAssetManager().AddTexture(t);
        if (texId) *texId = id;
    });
}

int nxs_scene_set_hdr_map(nxs_scene* s, const uint8_t* rgba8, uint32_t w, uint32_t h)
{
    return guarded([&] { s->scene.AddHDRMap(Texture(w, h, 4, rgba8)); });
}

int nxs_scene_add_mesh(nxs_scene* s, const nx_triangle* tris, uint32_t triCount, int32_t materialId, int32_t* meshId)
{
    return guarded([&] {
        std::vector<Triangle> v;
        v.reserve(triCount);
        for (uint32_t i = 0; i < triCount; i++) v.emplace_back(tris[i]);
        AssetManager& am = s->scene.GetAssetManager();
        const int32_t bvhId = am.CreateBVH(v);
        const int32_t id = am.AddMesh(Mesh("mesh" + std::to_string(bvhId), bvhId, materialId));
        if (meshId) *meshId = id;
    });
}

int nxs_scene_create_instance(nxs_scene* s, uint32_t meshId, int32_t materialId, const float pos[3], const float rotDeg[3], const float scale[3],
                              int32_t* instanceId)
{
    return guarded([&] {
        MeshInstance& mi = s->scene.CreateMeshInstance(meshId);
        mi.AssignMaterial(materialId);
        mi.SetTransform(make_float3(pos), make_float3(rotDeg), make_float3(scale));
        if (instanceId) *instanceId = static_cast<int32_t>(s->scene.GetMeshInstances().size()) - 1;
    });
}

int nxs_scene_assign_material(nxs_scene* s, uint32_t instanceId, int32_t materialId)
{
    return guarded([&] {
        if (instanceId >= s->scene.GetMeshInstances().size()) throw std::runtime_error("nxs_scene_assign_material: no such instance");
        s->scene.GetMeshInstances()[instanceId].AssignMaterial(materialId);
        s->scene.InvalidateMeshInstance(instanceId);
    });
}

int nxh_load_scene_file(const char* file, nxh_loaded_scene** out)
{
    return guarded([&] {
        if (!file || !out) throw std::runtime_error("nxh_load_scene_file: null argument");
        std::unique_ptr<nxh_loaded_scene> p(new nxh_loaded_scene);
        p->ls = OBJLoader::Parse(file);
        *out = p.release();
    });
}
void nxh_loaded_scene_free(nxh_loaded_scene* s) { delete s; }
uint32_t nxh_loaded_mesh_count(const nxh_loaded_scene* s) { return static_cast<uint32_t>(s->ls.meshes.size()); }
uint32_t nxh_loaded_mesh_triangle_count(const nxh_loaded_scene* s, uint32_t mesh)
{
    return mesh < s->ls.meshes.size() ? static_cast<uint32_t>(s->ls.meshes[mesh].size()) : 0u;
}
int nxh_loaded_mesh_triangles(const nxh_loaded_scene* s, uint32_t mesh, nx_triangle* dst)
{
    return guarded([&] {
        if (mesh >= s->ls.meshes.size() || !dst) throw std::runtime_error("nxh_loaded_mesh_triangles: bad arguments");
        for (size_t i = 0; i < s->ls.meshes[mesh].size(); i++) dst[i] = Triangle::ToDevice(s->ls.meshes[mesh][i]);
    });
}
uint32_t nxh_loaded_material_count(const nxh_loaded_scene* s) { return static_cast<uint32_t>(s->ls.materials.size()); }
int nxh_loaded_materials(const nxh_loaded_scene* s, nx_material* dst)
{
    return guarded([&] {
        if (!dst) throw std::runtime_error("nxh_loaded_materials: null destination");
        for (size_t i = 0; i < s->ls.materials.size(); i++) dst[i] = static_cast<const nx_material&>(s->ls.materials[i]);
    });
}
uint32_t nxh_loaded_instance_count(const nxh_loaded_scene* s) { return static_cast<uint32_t>(s->ls.instances.size()); }
int nxh_loaded_instances(const nxh_loaded_scene* s, nx_loaded_instance* dst)
{
    return guarded([&] {
        if (!dst) throw std::runtime_error("nxh_loaded_instances: null destination");
        for (size_t i = 0; i < s->ls.instances.size(); i++) {
            const LoadedInstance& in = s->ls.instances[i];
            dst[i].mesh = in.mesh;
            dst[i].material = in.material;
            store(dst[i].position, in.position);
            store(dst[i].rotation, in.rotation);
            store(dst[i].scale, in.scale);
        }
    });
}

uint32_t nxh_loaded_texture_count(const nxh_loaded_scene* s) { return static_cast<uint32_t>(s->ls.textures.size()); }
int nxh_loaded_texture_info(const nxh_loaded_scene* s, uint32_t index, uint32_t* width, uint32_t* height, int32_t* kind)
{
    return guarded([&] {
        if (index >= s->ls.textures.size()) throw std::runtime_error("nxh_loaded_texture_info: no such texture");
        const Texture& t = s->ls.textures[index];
        if (width) *width = t.width;
        if (height) *height = t.height;
        if (kind) *kind = t.type == Texture::Type::EMISSIVE ? 1 : 0;
    });
}
int nxh_loaded_texture_pixels(const nxh_loaded_scene* s, uint32_t index, uint8_t* dstRgba8)
{
    return guarded([&] {
        if (index >= s->ls.textures.size() || !dstRgba8) throw std::runtime_error("nxh_loaded_texture_pixels: bad argument");
        const Texture& t = s->ls.textures[index];
        std::memcpy(dstRgba8, t.pixels.data(), t.pixels.size());
    });
}
int nxh_loaded_material_textures(const nxh_loaded_scene* s, int32_t* diffuseTexture, int32_t* emissiveTexture)
{
    return guarded([&] {
        if (!diffuseTexture || !emissiveTexture) throw std::runtime_error("nxh_loaded_material_textures: null destination");
        for (size_t i = 0; i < s->ls.materials.size(); i++) {
            diffuseTexture[i] = s->ls.materialDiffuseTexture[i];
            emissiveTexture[i] = s->ls.materialEmissiveTexture[i];
        }
    });
}
uint32_t nxh_loaded_warning_count(const nxh_loaded_scene* s) { return static_cast<uint32_t>(s->ls.warnings.size()); }
const char* nxh_loaded_warning(const nxh_loaded_scene* s, uint32_t index) { return index < s->ls.warnings.size() ? s->ls.warnings[index].c_str() : ""; }

int nxh_decode_png(const uint8_t* data, size_t size, uint32_t* width, uint32_t* height, uint32_t* channels, uint8_t* dstRgba8, size_t dstCapacity)
{
    return guarded([&] {
        if (!data || !width || !height) throw std::runtime_error("nxh_decode_png: null argument");
        const Texture t = IMGLoader::LoadIMG(data, size);
        *width = t.width;
        *height = t.height;
        if (channels) *channels = t.channels;
        if (dstRgba8) {
            if (dstCapacity < t.pixels.size()) throw std::runtime_error("nxh_decode_png: destination too small");
            std::memcpy(dstRgba8, t.pixels.data(), t.pixels.size());
        }
    });
}

int nxs_scene_set_instance_transform(nxs_scene* s, uint32_t instanceId, const float pos[3], const float rotDeg[3], const float scale[3])
{
    return guarded([&] {
        if (instanceId >= s->scene.GetMeshInstances().size()) throw std::runtime_error("nxs_scene_set_instance_transform: no such instance");
        s->scene.GetMeshInstances()[instanceId].SetTransform(make_float3(pos), make_float3(rotDeg), make_float3(scale));
        s->scene.InvalidateMeshInstance(instanceId);  // what the viewer's transform panel does (SceneHierarchyPanel.cpp)
    });
}

int nxs_scene_set_tlas_refit(nxs_scene* s, int enable)
{
    return guarded([&] { s->scene.SetTlasRefit(enable != 0); });
}

int nxs_scene_set_device_tlas(nxs_scene* s, int enable)
{
    return guarded([&] { s->scene.SetDeviceTlasBuild(enable != 0); });
}

int nxs_scene_load_file(nxs_scene* s, const char* path, const char* fileName)
{
    return guarded([&] {
        if (!path || !fileName) throw std::runtime_error("nxs_scene_load_file: null argument");
        s->scene.CreateMeshInstanceFromFile(path, fileName);
    });
}

int nxs_scene_set_camera(nxs_scene* s, const float pos[3], const float forward[3], float hfov, float focusDist, float defocusAngle)
{
    return guarded([&] {
        Camera& c = *s->scene.GetCamera();
        c.LookAt(make_float3(pos), make_float3(forward));
        c.SetHorizontalFOV(hfov);
        c.GetFocusDist() = focusDist;
        c.GetDefocusAngle() = defocusAngle;
    });
}

int nxs_scene_set_render_settings(nxs_scene* s, const nx_render_settings* settings)
{
    return guarded([&] {
        std::memcpy(&s->scene.GetRenderSettings(), settings, sizeof(nx_render_settings));
        s->scene.Invalidate();
    });
}

int nxs_scene_update(nxs_scene* s)
{
    return guarded([&] { s->scene.Update(); });
}

uint32_t nxs_scene_light_count(const nxs_scene* s) { return static_cast<uint32_t>(s->scene.GetLights().size()); }
uint32_t nxs_scene_instance_count(const nxs_scene* s) { return static_cast<uint32_t>(s->scene.GetBVHInstances().size()); }

int nxs_pathtracer_create(uint32_t width, uint32_t height, int device, nxs_pathtracer** out)
{
    return guarded([&] { *out = new nxs_pathtracer(width, height, device); });
}
void nxs_pathtracer_destroy(nxs_pathtracer* p) { delete p; }
int nxs_pathtracer_set_modes(nxs_pathtracer* p, int rngMode, int compactMode, int conductorMode)
{
    return guarded([&] { p->pt.SetModes(rngMode, compactMode, conductorMode); });
}
int nxs_pathtracer_set_frames_per_pass(nxs_pathtracer* p, uint32_t frames) { return guarded([&] { p->pt.SetFramesPerPass(frames); }); }
int nxs_pathtracer_set_passes_in_flight(nxs_pathtracer* p, uint32_t passes) { return guarded([&] { p->pt.SetPassesInFlight(passes); }); }
int nxs_pathtracer_set_pixel_order(nxs_pathtracer* p, int order) { return guarded([&] { p->pt.SetPixelOrder(order); }); }
int nxs_pathtracer_set_entry_points(nxs_pathtracer* p, int on) { return guarded([&] { p->pt.SetEntryPoints(on != 0); }); }
int nxs_pathtracer_set_device_blas_build(nxs_pathtracer* p, nxs_scene* s, int enable)
{
    return guarded([&] { p->pt.SetDeviceBlasBuild(s->scene, enable != 0); });
}
int nxs_pathtracer_update_device_scene(nxs_pathtracer* p, nxs_scene* s)
{
    return guarded([&] { p->pt.UpdateDeviceScene(s->scene); });
}
int nxs_pathtracer_render(nxs_pathtracer* p, nxs_scene* s)
{
    return guarded([&] { p->pt.Render(s->scene); });
}
int nxs_pathtracer_reset_frame_number(nxs_pathtracer* p)
{
    return guarded([&] { p->pt.ResetFrameNumber(); });
}
uint32_t nxs_pathtracer_frame_number(const nxs_pathtracer* p) { return p->pt.GetFrameNumber(); }
int nxs_pathtracer_read_pixels(nxs_pathtracer* p, uint32_t* rgba8)
{
    return guarded([&] {
        const std::vector<uint32_t>& px = p->pt.GetPixelBuffer();
        std::memcpy(rgba8, px.data(), px.size() * 4);
    });
}
struct nxhip_ctx* nxs_pathtracer_device_context(nxs_pathtracer* p) { return p->pt.GetDeviceContext(); }

int nxs_scene_add_hdr_map_file(nxs_scene* s, const char* path, const char* fileName)
{
    return guarded([&] {
        if (!path || !fileName) throw std::runtime_error("nxs_scene_add_hdr_map_file: null argument");
        s->scene.AddHDRMap(path, fileName);
    });
}

int nxs_renderer_create(uint32_t width, uint32_t height, nxs_scene* scene, int device, nxs_renderer** out)
{
    return guarded([&] {
        if (!scene || !out) throw std::runtime_error("nxs_renderer_create: null argument");
        *out = new nxs_renderer(width, height, &scene->scene, device);
    });
}
void nxs_renderer_destroy(nxs_renderer* r) { delete r; }
int nxs_renderer_render(nxs_renderer* r, nxs_scene* scene, float deltaTime)
{
    return guarded([&] { r->r.Render(scene->scene, deltaTime); });
}
int nxs_renderer_reset(nxs_renderer* r) { return guarded([&] { r->r.Reset(); }); }
int nxs_renderer_on_resize(nxs_renderer* r, uint32_t width, uint32_t height) { return guarded([&] { r->r.OnResize(width, height); }); }
int nxs_renderer_save_screenshot(nxs_renderer* r, const char* path)
{
    return guarded([&] {
        if (!path || !r->r.SaveScreenshot(path)) throw std::runtime_error(std::string("cannot write ") + (path ? path : "(null)"));
    });
}
int nxs_renderer_save_exr(nxs_renderer* r, const char* path)
{
    return guarded([&] {
        if (!path || !r->r.SaveAccumulationEXR(path)) throw std::runtime_error(std::string("cannot write ") + (path ? path : "(null)"));
    });
}
uint32_t nxs_renderer_frame_number(const nxs_renderer* r) { return r->r.GetFrameNumber(); }
double nxs_renderer_megasamples_per_second(const nxs_renderer* r) { return r->r.GetMegaSamplesPerSecond(); }
struct nxhip_ctx* nxs_renderer_device_context(nxs_renderer* r) { return r->r.GetPathTracer().GetDeviceContext(); }
int nxs_renderer_set_modes(nxs_renderer* r, int rngMode, int compactMode, int conductorMode)
{
    return guarded([&] { r->r.GetPathTracer().SetModes(rngMode, compactMode, conductorMode); });
}

int nxh_write_png(const char* path, const uint32_t* rgba8, uint32_t width, uint32_t height, int flipVertically)
{
    return guarded([&] {
        if (!path || !WritePNG(path, rgba8, width, height, flipVertically != 0)) throw std::runtime_error("nxh_write_png: cannot write the file");
    });
}
int nxh_write_exr(const char* path, const float* rgb, uint32_t width, uint32_t height, int flipVertically)
{
    return guarded([&] {
        if (!path || !WriteEXR(path, rgb, width, height, flipVertically != 0)) throw std::runtime_error("nxh_write_exr: cannot write the file");
    });
}

}  // extern "C"
