// nexus/Camera.h — thin-lens camera of the kept API surface.
// Mirrors /root/reference/Nexus/src/Scene/Camera.h:9-50 and Camera.cpp:14-35,102-168 minus the GLFW/glm input
// handling (viewer side, out of scope): position / forward / right, horizontal FOV, focus distance, defocus angle,
// ToDevice() computing the viewport basis the generate kernel reads.
#pragma once

#include "Math.h"

namespace nexus {

class Camera {
public:
    Camera(float horizontalFOV, uint32_t width, uint32_t height);
    Camera(float3 position, float3 forward, float horizontalFOV, uint32_t width, uint32_t height, float focusDistance, float defocusAngle);

    void OnResize(uint32_t width, uint32_t height);
    void SetHorizontalFOV(float horizontalFOV) { m_HorizontalFOV = horizontalFOV; m_Invalid = true; }
    float& GetHorizontalFOV() { return m_HorizontalFOV; }
    float& GetDefocusAngle() { return m_DefocusAngle; }
    float& GetFocusDist() { return m_FocusDist; }
    uint32_t GetViewportWidth() const { return m_ViewportWidth; }
    uint32_t GetViewportHeight() const { return m_ViewportHeight; }
    float3& GetPosition() { return m_Position; }
    float3& GetForwardDirection() { return m_ForwardDirection; }
    float3& GetRightDirection() { return m_RightDirection; }
    void LookAt(float3 position, float3 forward);

    bool IsInvalid() const { return m_Invalid; }
    void SetInvalid(bool invalid) { m_Invalid = invalid; }
    void Invalidate() { m_Invalid = true; }

    static nx_camera ToDevice(const Camera& camera);

private:
    float m_HorizontalFOV;
    float m_DefocusAngle;
    float m_FocusDist;
    uint32_t m_ViewportWidth;
    uint32_t m_ViewportHeight;
    float3 m_Position;
    float3 m_ForwardDirection;
    float3 m_RightDirection;
    bool m_Invalid = true;
};

}  // namespace nexus
