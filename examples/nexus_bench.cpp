// nexus_bench — BASELINE.json configs[1] measured THROUGH THE KEPT C++ API: the drop-in boundary north_star names ("drops in behind
// the existing viewer" = Renderer::Render -> PathTracer::Render, /root/reference/Nexus/src/Renderer/Renderer.cpp:41-77,
// Renderer/PathTracer.cpp:248-288).  bench.py drives the device layer through ctypes; this program uses nothing but the classes a
// maintainer of the reference keeps — nexus::Scene, OBJLoader (through Scene::CreateMeshInstanceFromFile), AssetManager,
// PathTracer::SetDeviceBlasBuild / UpdateDeviceScene / Render — and reports Msamples/s by the viewer's own definition
// (Renderer/Panels/MetricsPanel.cpp:23-37: width x height per Render()ed frame over the accumulated wall time).
//
//   nexus_bench <dir/> <mesh.obj> [--mode reference|headline] [--frames K] [--warmup W] [--reps R] [--width X --height Y] [--path-length L]
//               [--passes-in-flight P]
//               [--rgba8 out.bin]
//
//   --mode reference : the reference's own semantics — slot-keyed random numbers, ONE frame per Render() call, rows, no entry points,
//                      the conductor kernel counts and does not shade (the reference's is commented out)
//   --mode headline  : bench.py's settings — pixel-keyed random numbers, fast compaction, extended conductor, 8 x 8 tiles, entry points,
//                      K frames per Render() call (one pass)
// The scene is workloads.config2's: the mesh file (bench.py writes its displaced torus as a Wavefront .obj) as a conductor on a diffuse
// floor under an emissive quad; BLASes and TLAS built on the device.  One JSON line on stdout.
// Build: make bench_example   (nexus_amd/lib/nexus_bench, beside the library it links)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "nexus/PathTracer.h"
#include "nexus/Scene.h"
#include "nexus_hip.h"

using nexus::float2;
using nexus::float3;
using nexus::make_float2;
using nexus::make_float3;

namespace {

// two triangles (p0, p1, p2), (p0, p2, p3) with the face normal and unit-square texture coordinates (nexus_amd/scenegen.py quad)
std::vector<nexus::Triangle> Quad(float3 p0, float3 p1, float3 p2, float3 p3, float3 n)
{
    return {nexus::Triangle(p0, p1, p2, n, n, n, make_float2(0, 0), make_float2(1, 0), make_float2(1, 1)),
            nexus::Triangle(p0, p2, p3, n, n, n, make_float2(0, 0), make_float2(1, 1), make_float2(0, 1))};
}

double Median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <dir/> <mesh.obj> [--mode reference|headline] [--frames K] [--warmup W] [--reps R] [--width X] [--height Y] [--path-length L] [--rgba8 file]\n", argv[0]);
        return 2;
    }
    if (nxhip_check_library(nxhip_header_abi_stamp()) != NXHIP_OK) {
        std::fprintf(stderr, "%s\n", nxhip_last_error());
        return 3;
    }
    const std::string dir = argv[1], file = argv[2];
    std::string mode = "headline", rgbaOut;
    uint32_t width = 1920, height = 1080;
    int frames = 20, warmup = 5, reps = 5, pathLength = 8, inFlight = 1;
    for (int i = 3; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--mode") mode = v;
        else if (k == "--frames") frames = std::atoi(v.c_str());
        else if (k == "--warmup") warmup = std::atoi(v.c_str());
        else if (k == "--reps") reps = std::atoi(v.c_str());
        else if (k == "--width") width = static_cast<uint32_t>(std::atoi(v.c_str()));
        else if (k == "--height") height = static_cast<uint32_t>(std::atoi(v.c_str()));
        else if (k == "--path-length") pathLength = std::atoi(v.c_str());
        else if (k == "--rgba8") rgbaOut = v;
        else if (k == "--passes-in-flight") inFlight = std::atoi(v.c_str());  // consecutive Render() calls overlap on the GPU (PathTracer::SetPassesInFlight)
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    const bool headline = mode == "headline";
    if (!headline && mode != "reference") { std::fprintf(stderr, "--mode must be reference or headline\n"); return 2; }
    if (frames < 1 || warmup < 0 || reps < 1) { std::fprintf(stderr, "bad --frames / --warmup / --reps\n"); return 2; }
    try {
        const auto t0 = std::chrono::steady_clock::now();
        nexus::Scene scene(width, height);
        nexus::PathTracer pathTracer(width, height, 0);
        pathTracer.SetDeviceBlasBuild(scene, true);
        scene.SetDeviceTlasBuild(true);
        // materials in workloads.config2's order: 0 the mesh's conductor, 1 the floor, 2 the emitter
        nexus::Material conductor, floorMat, lightMat;
        conductor.type = NX_MAT_CONDUCTOR;
        const float ior[3] = {0.2f, 0.9f, 1.1f}, kk[3] = {3.9f, 2.4f, 2.2f};
        for (int c = 0; c < 3; c++) { conductor.conductor.ior[c] = ior[c]; conductor.conductor.k[c] = kk[c]; }
        conductor.conductor.roughness = 0.3f;
        for (int c = 0; c < 3; c++) floorMat.diffuse.albedo[c] = 0.7f;
        for (int c = 0; c < 3; c++) { lightMat.diffuse.albedo[c] = 0.8f; lightMat.emissive[c] = 1.0f; }
        lightMat.intensity = 20.0f;
        nexus::AssetManager& assets = scene.GetAssetManager();
        const size_t firstMaterial = assets.GetMaterials().size();
        scene.CreateMeshInstanceFromFile(dir, file);  // OBJLoader::LoadOBJ: the mesh, its (default) material, one instance
        const size_t meshInstances = scene.GetMeshInstances().size();
        if (meshInstances < 1) throw std::runtime_error("the file holds no mesh");
        const int conductorId = assets.AddMaterial(conductor), floorId = assets.AddMaterial(floorMat), lightId = assets.AddMaterial(lightMat);
        (void)firstMaterial;
        for (size_t i = 0; i < meshInstances; i++) {
            scene.GetMeshInstances()[i].AssignMaterial(conductorId);
            scene.InvalidateMeshInstance(static_cast<uint32_t>(i));
        }
        const auto addQuad = [&](const char* name, const std::vector<nexus::Triangle>& tris, int materialId) {
            const int32_t bvhId = assets.CreateBVH(tris);
            const int32_t meshId = assets.AddMesh(nexus::Mesh(name, bvhId, materialId));
            scene.CreateMeshInstance(static_cast<uint32_t>(meshId)).AssignMaterial(materialId);
        };
        addQuad("floor", Quad(make_float3(-6, 0, -6), make_float3(-6, 0, 6), make_float3(6, 0, 6), make_float3(6, 0, -6), make_float3(0, 1, 0)), floorId);
        addQuad("light", Quad(make_float3(-1.2f, 4.0f, -1.2f), make_float3(1.2f, 4.0f, -1.2f), make_float3(1.2f, 4.0f, 1.2f), make_float3(-1.2f, 4.0f, 1.2f), make_float3(0, -1, 0)), lightId);
        // workloads._look((0, 3.3, 4.9) -> (0, 0.35, 0), 52 degrees): the forward vector normalised in binary64, then rounded
        const double fx = 0.0 - 0.0, fy = 0.35 - 3.3, fz = 0.0 - 4.9;
        const double len = std::sqrt(fx * fx + fy * fy + fz * fz);
        scene.GetCamera()->LookAt(make_float3(0.0f, 3.3f, 4.9f), make_float3(static_cast<float>(fx / len), static_cast<float>(fy / len), static_cast<float>(fz / len)));
        scene.GetCamera()->SetHorizontalFOV(52.0f);
        scene.GetRenderSettings().useMIS = true;
        scene.GetRenderSettings().pathLength = static_cast<unsigned char>(pathLength);
        scene.GetRenderSettings().backgroundColor = make_float3(1.0f);
        scene.GetRenderSettings().backgroundIntensity = 0.0f;
        scene.Update();
        if (headline) {
            pathTracer.SetModes(NX_RNG_PIXEL_KEYED, NX_COMPACT_FAST, NX_CONDUCTOR_EXTENDED);
            pathTracer.SetPixelOrder(NXHIP_ORDER_TILES);
            pathTracer.SetEntryPoints(true);
            pathTracer.SetFramesPerPass(static_cast<uint32_t>(frames));
        } else {
            pathTracer.SetModes(NX_RNG_REFERENCE_SLOT, NX_COMPACT_FAST, NX_CONDUCTOR_REFERENCE);
        }
        if (inFlight > 1) pathTracer.SetPassesInFlight(static_cast<uint32_t>(inFlight));
        pathTracer.UpdateDeviceScene(scene);
        const double buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

        // Render() calls per timed region: one pass of K frames (headline), or K single frames (the reference's loop)
        const int callsPerRegion = headline ? 1 : frames;
        const int warmCalls = headline ? (warmup + frames - 1) / frames : warmup;
        nxhip_ctx* ctx = pathTracer.GetDeviceContext();
        for (int i = 0; i < warmCalls; i++) pathTracer.Render(scene);
        if (nxhip_sync(ctx) != NXHIP_OK) throw std::runtime_error(nxhip_last_error());
        std::vector<double> ms;
        for (int r = 0; r < reps; r++) {
            const auto a = std::chrono::steady_clock::now();
            for (int i = 0; i < callsPerRegion; i++) pathTracer.Render(scene);
            if (nxhip_sync(ctx) != NXHIP_OK) throw std::runtime_error(nxhip_last_error());
            ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count());
        }
        const double med = Median(ms);
        const double msamples = static_cast<double>(width) * height * frames / (med * 1.0e-3) / 1.0e6;
        const std::vector<uint32_t>& px = pathTracer.GetPixelBuffer();
        unsigned long long sum = 0;
        for (uint32_t p : px) sum = sum * 1099511628211ull + p;
        if (!rgbaOut.empty()) {
            std::FILE* fp = std::fopen(rgbaOut.c_str(), "wb");
            if (!fp || std::fwrite(px.data(), 4, px.size(), fp) != px.size()) throw std::runtime_error("cannot write " + rgbaOut);
            std::fclose(fp);
        }
        std::printf("{\"metric\": \"Msamples/sec through nexus::PathTracer::Render\", \"value\": %.1f, \"unit\": \"Msamples/s\", \"mode\": \"%s\", \"width\": %u, \"height\": %u, "
                    "\"frames_timed\": %d, \"passes_in_flight\": %d, \"render_calls_per_region\": %d, \"warmup_frames\": %d, \"reps\": %d, \"median_region_ms\": %.3f, \"ms_per_frame\": %.4f, "
                    "\"instances\": %zu, \"lights\": %zu, \"frame_number\": %u, \"scene_build_s\": %.2f, \"rgba8_hash\": \"%016llx\", "
                    "\"definition\": \"width x height x frames / wall seconds of the Render() calls + device sync (MetricsPanel.cpp:23-37)\"}\n",
                    msamples, mode.c_str(), width, height, frames, inFlight, callsPerRegion, warmCalls * (headline ? frames : 1), reps, med, med / frames, scene.GetBVHInstances().size(),
                    scene.GetLights().size(), pathTracer.GetFrameNumber(), buildSeconds, sum);
        pathTracer.SetDeviceBlasBuild(scene, false);  // the scene outlives the path tracer in this scope
    } catch (const std::exception& e) {
        std::fprintf(stderr, "nexus_bench: %s\n", e.what());
        return 1;
    }
    return 0;
}
