// Camera.cpp — see include/nexus/Camera.h.  ToDevice follows /root/reference/Nexus/src/Scene/Camera.cpp:142-168:
// halfWidth = focusDist * tan(hFOV/2), halfHeight = halfWidth / aspect, right = cross(forward, +Y), up = cross(right, forward).
#include "nexus/Camera.h"

namespace nexus {

Camera::Camera(float horizontalFOV, uint32_t width, uint32_t height)
    : m_HorizontalFOV(horizontalFOV), m_DefocusAngle(10.0f), m_FocusDist(5.0f), m_ViewportWidth(width), m_ViewportHeight(height),
      m_Position(make_float3(0.0f, 0.0f, 2.0f)), m_ForwardDirection(make_float3(0.0f, 0.0f, -1.0f)), m_RightDirection(make_float3(1.0f, 0.0f, 0.0f))
{
}

Camera::Camera(float3 position, float3 forward, float horizontalFOV, uint32_t width, uint32_t height, float focusDistance, float defocusAngle)
    : m_HorizontalFOV(horizontalFOV), m_DefocusAngle(defocusAngle), m_FocusDist(focusDistance), m_ViewportWidth(width), m_ViewportHeight(height),
      m_Position(position), m_ForwardDirection(forward), m_RightDirection(cross(forward, make_float3(0.0f, 1.0f, 0.0f)))
{
}

void Camera::OnResize(uint32_t width, uint32_t height)
{
    if (width == m_ViewportWidth && height == m_ViewportHeight) return;
    m_ViewportWidth = width;
    m_ViewportHeight = height;
    Invalidate();
}

void Camera::LookAt(float3 position, float3 forward)
{
    m_Position = position;
    m_ForwardDirection = forward;
    m_RightDirection = cross(forward, make_float3(0.0f, 1.0f, 0.0f));
    Invalidate();
}

nx_camera Camera::ToDevice(const Camera& camera)
{
    nx_camera d;
    std::memset(&d, 0, sizeof d);
    const float3 forward = camera.m_ForwardDirection;
    const float3 up = cross(camera.m_RightDirection, forward);
    const float aspectRatio = camera.m_ViewportWidth / static_cast<float>(camera.m_ViewportHeight);
    const float halfWidth = camera.m_FocusDist * std::tan(static_cast<float>(camera.m_HorizontalFOV / 2.0f * PI / 180.0f));
    const float halfHeight = halfWidth / aspectRatio;
    const float3 viewportX = (2 * halfWidth) * camera.m_RightDirection;
    const float3 viewportY = (2 * halfHeight) * up;
    const float3 lowerLeftCorner = camera.m_Position - viewportX / 2.0f - viewportY / 2.0f + forward * camera.m_FocusDist;
    const float lensRadius = camera.m_FocusDist * std::tan(static_cast<float>(camera.m_DefocusAngle / 2.0f * PI / 180.0f));
    store(d.position, camera.m_Position);
    store(d.right, camera.m_RightDirection);
    store(d.up, up);
    d.lensRadius = lensRadius;
    store(d.lowerLeftCorner, lowerLeftCorner);
    store(d.viewportX, viewportX);
    store(d.viewportY, viewportY);
    d.resolution[0] = camera.m_ViewportWidth;
    d.resolution[1] = camera.m_ViewportHeight;
    return d;
}

}  // namespace nexus
