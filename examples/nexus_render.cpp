// nexus_render — the reference's render loop (Renderer.cpp:41-77: scene.Update, UpdateDeviceScene, Render) driven headless
// through the kept C++ host API: load a .glb / .obj, path-trace `frames` frames on an MI355X, write the image as a PPM.
//
//   nexus_render <dir/> <file.glb|file.obj> <out.ppm> [width height frames pathLength]
//                [eyeX eyeY eyeZ fwdX fwdY fwdZ hfovDeg]
//
// Build: make example   (links nexus_amd/lib/libnexus_amd.so)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>
#include <vector>

#include "nexus/PathTracer.h"
#include "nexus_hip.h"
#include "nexus/Scene.h"

int main(int argc, char** argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <dir/> <file.glb|file.obj> <out.ppm> [width height frames pathLength] [eye(3) forward(3) hfov]\n", argv[0]);
        return 2;
    }
    // a program built against these headers and run with another build of libnexus_amd.so gets an error string here, not a GPU fault
    // in its first launch (include/nexus_hip.h: the header's stamp is compiled into the caller, the library compares it with its own)
    if (nxhip_check_library(nxhip_header_abi_stamp()) != NXHIP_OK) {
        std::fprintf(stderr, "%s\n", nxhip_last_error());
        return 3;
    }
    const std::string dir = argv[1], file = argv[2], out = argv[3];
    const uint32_t width = argc > 4 ? static_cast<uint32_t>(std::atoi(argv[4])) : 512;
    const uint32_t height = argc > 5 ? static_cast<uint32_t>(std::atoi(argv[5])) : 512;
    const int frames = argc > 6 ? std::atoi(argv[6]) : 16;
    const int pathLength = argc > 7 ? std::atoi(argv[7]) : 4;
    float cam[7] = {0.0f, 1.0f, 3.9f, 0.0f, 0.0f, -1.0f, 40.0f};  // the Cornell box view of BASELINE.json configs[0]
    for (int k = 0; k < 7 && 8 + k < argc; k++) cam[k] = static_cast<float>(std::atof(argv[8 + k]));
    try {
        nexus::Scene scene(width, height);
        nexus::PathTracer pathTracer(width, height, 0);
        // NEXUS_DEVICE_BUILDERS=1: the BVH of every mesh the loader adds is built on the GPU (the reference's SAH rule and collapse
        // in HBM: a tenth of the host builder's time, a thousandth of the reference's) and so is the TLAS; the default is the
        // reference's flow, host builders and an upload
        const bool deviceBuilders = std::getenv("NEXUS_DEVICE_BUILDERS") != nullptr;
        if (deviceBuilders) {
            pathTracer.SetDeviceBlasBuild(scene, true);
            scene.SetDeviceTlasBuild(true);
        }
        scene.CreateMeshInstanceFromFile(dir, file);
        scene.GetCamera()->LookAt(nexus::make_float3(cam[0], cam[1], cam[2]), nexus::make_float3(cam[3], cam[4], cam[5]));
        scene.GetCamera()->SetHorizontalFOV(cam[6]);
        scene.GetRenderSettings().pathLength = static_cast<unsigned char>(pathLength);
        scene.Update();

        // NEXUS_DETERMINISTIC=1: pixel-keyed RNG, so the image does not depend on the order in which racing workgroups take
        // queue slots (the reference's slot-keyed RNG makes every run a different noise pattern)
        if (std::getenv("NEXUS_DETERMINISTIC")) pathTracer.SetModes(NX_RNG_PIXEL_KEYED, NX_COMPACT_FAST, NX_CONDUCTOR_REFERENCE);
        pathTracer.UpdateDeviceScene(scene);
        for (int f = 0; f < frames; f++) pathTracer.Render(scene);
        const std::vector<uint32_t>& px = pathTracer.GetPixelBuffer();

        std::FILE* fp = std::fopen(out.c_str(), "wb");
        if (!fp) {
            std::fprintf(stderr, "cannot write %s\n", out.c_str());
            return 1;
        }
        std::fprintf(fp, "P6\n%u %u\n255\n", width, height);
        std::vector<unsigned char> row(static_cast<size_t>(width) * 3);
        for (uint32_t y = 0; y < height; y++) {  // image row 0 is the bottom of the viewport
            const uint32_t* src = px.data() + static_cast<size_t>(height - 1 - y) * width;
            for (uint32_t x = 0; x < width; x++) {
                row[3 * x + 0] = static_cast<unsigned char>(src[x] & 0xffu);
                row[3 * x + 1] = static_cast<unsigned char>((src[x] >> 8) & 0xffu);
                row[3 * x + 2] = static_cast<unsigned char>((src[x] >> 16) & 0xffu);
            }
            std::fwrite(row.data(), 1, row.size(), fp);
        }
        std::fclose(fp);
        std::printf("%s: %u x %u, %d frames, %zu instances, %zu lights%s\n", out.c_str(), width, height, frames, scene.GetBVHInstances().size(), scene.GetLights().size(),
                    deviceBuilders ? ", BVHs built on the device" : "");
        if (deviceBuilders) pathTracer.SetDeviceBlasBuild(scene, false);  // the scene outlives the path tracer in this scope
    } catch (const std::exception& e) {
        std::fprintf(stderr, "nexus_render: %s\n", e.what());
        return 1;
    }
    return 0;
}
