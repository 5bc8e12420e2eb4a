// nexus/Renderer.h — the reference's frame driver without its window: what Renderer::Render does between the ImGui calls
// (/root/reference/Nexus/src/Renderer/Renderer.cpp:41-77, 162-215; Renderer.h:17-46), kept so that an application written
// against `Renderer` + `Scene` keeps its call sequence:
//     Renderer renderer(width, height, &scene);
//     loop: renderer.Render(scene, deltaTime);      // scene.Update() when invalid, UpdateDeviceScene, PathTracer::Render
//     renderer.SaveScreenshot("frame.png");          // the RGBA8 image, rows flipped as stbi_flip_vertically_on_write does
// GLFW / ImGui / OpenGL (the window, the panels, the texture the pixel buffer is unpacked into) are out of scope; the panels'
// one computed figure, the viewer's "Megarays/sec" (Panels/MetricsPanel.cpp:28-56), is kept as GetMegaSamplesPerSecond().
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "PathTracer.h"
#include "Scene.h"

namespace nexus {

class Renderer {
public:
    Renderer(uint32_t width, uint32_t height, Scene* scene, int device = 0);

    void Reset();                                   // Renderer.cpp:35-39
    void OnResize(uint32_t width, uint32_t height);  // Renderer.cpp:171-181 (the camera is resized by the caller, as RenderUI does)
    void Render(Scene& scene, float deltaTime);      // Renderer.cpp:41-77
    // Renderer.cpp:183-215: PNG of the current RGBA8 image ("\.png" appended when missing).  Returns false if it cannot be written.
    bool SaveScreenshot(const std::string& filepath);
    // Extension: the float accumulation as OpenEXR (scanline, uncompressed, 32-bit float R G B), top row first.
    bool SaveAccumulationEXR(const std::string& filepath);

    // Extension: PathTracer::SetDeviceBlasBuild for the renderer's scene (meshes loaded from now on are built on the GPU)
    void SetDeviceBlasBuild(bool enable) { m_PathTracer.SetDeviceBlasBuild(*m_Scene, enable); }

    PathTracer& GetPathTracer() { return m_PathTracer; }
    uint32_t GetFrameNumber() const { return m_PathTracer.GetFrameNumber(); }
    // MetricsPanel: samples per second over the frames rendered since the last Reset, in millions (width * height * frames / s)
    double GetMegaSamplesPerSecond() const { return m_AccumulatedTime > 0.0 ? 1e-6 * static_cast<double>(m_ViewportWidth) * m_ViewportHeight * m_Frames / m_AccumulatedTime : 0.0; }

private:
    uint32_t m_ViewportWidth, m_ViewportHeight;
    Scene* m_Scene;
    PathTracer m_PathTracer;
    double m_AccumulatedTime = 0.0;
    uint64_t m_Frames = 0;
};

// Image files (no stb): PNG (RGBA8, zlib deflate) and OpenEXR (scanline, uncompressed, float32 B G R channels).
// `flipVertically`: write the last row first (the render buffer's row 0 is the bottom of the viewport).
bool WritePNG(const std::string& path, const uint32_t* rgba8, uint32_t width, uint32_t height, bool flipVertically);
bool WriteEXR(const std::string& path, const float* rgb, uint32_t width, uint32_t height, bool flipVertically);

}  // namespace nexus
