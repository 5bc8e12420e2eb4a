// nexus/Camera.h — thin-lens camera of the kept API surface.
// Same public methods as /root/reference/Nexus/src/Scene/Camera.h:9-50 (so viewer code compiles against it) minus the
// GLFW / glm input handling (OnUpdate, RayThroughPixel, mouse state: viewer side, out of scope).  State is grouped the
// way the device consumes it: a pose, a lens, a viewport.  ToDevice() derives the viewport basis the generate kernel
// reads (Camera.cpp:142-168).
#pragma once

#include "Math.h"

namespace nexus {

class Camera {
public:
    struct Pose {
        float3 position, forward, right;
    };
    struct Lens {
        float horizontalFovDeg;  // full horizontal angle
        float defocusAngleDeg;   // aperture cone angle; 0 = pinhole
        float focusDist;
    };

    // Reference defaults: at (0,0,2) looking down -z, defocus 10 degrees, focus distance 5.
    Camera(float horizontalFOV, uint32_t width, uint32_t height);
    Camera(float3 position, float3 forward, float horizontalFOV, uint32_t width, uint32_t height, float focusDistance, float defocusAngle);

    // ---- pose
    void LookAt(float3 position, float3 forward);  // right := forward x +Y
    float3& GetPosition() { return m_Pose.position; }
    float3& GetForwardDirection() { return m_Pose.forward; }
    float3& GetRightDirection() { return m_Pose.right; }

    // ---- lens (the reference hands out references for its ImGui sliders; callers then Invalidate())
    void SetHorizontalFOV(float horizontalFOV)
    {
        m_Lens.horizontalFovDeg = horizontalFOV;
        m_Dirty = true;
    }
    float& GetHorizontalFOV() { return m_Lens.horizontalFovDeg; }
    float& GetDefocusAngle() { return m_Lens.defocusAngleDeg; }
    float& GetFocusDist() { return m_Lens.focusDist; }

    // ---- viewport
    void OnResize(uint32_t width, uint32_t height);
    uint32_t GetViewportWidth() const { return m_Viewport[0]; }
    uint32_t GetViewportHeight() const { return m_Viewport[1]; }

    // ---- dirty tracking consumed by Scene::IsInvalid / PathTracer::UpdateDeviceScene
    void Invalidate() { m_Dirty = true; }
    void SetInvalid(bool invalid) { m_Dirty = invalid; }
    bool IsInvalid() const { return m_Dirty; }

    static nx_camera ToDevice(const Camera& camera);

private:
    Pose m_Pose;
    Lens m_Lens;
    uint32_t m_Viewport[2];
    bool m_Dirty = true;
};

}  // namespace nexus
