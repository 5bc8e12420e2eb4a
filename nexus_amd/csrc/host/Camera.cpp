// Camera.cpp — see include/nexus/Camera.h.  ToDevice reproduces the arithmetic of
// /root/reference/Nexus/src/Scene/Camera.cpp:142-168 operation for operation (the generate kernel's rays depend on it bit for
// bit): halfWidth = focusDist * tan(hFOV/2), halfHeight = halfWidth / aspect, up = right x forward, viewport edges
// 2*half*axis, lower-left corner = position - X/2 - Y/2 + forward*focusDist, lens radius = focusDist * tan(defocus/2).
#include "nexus/Camera.h"

namespace nexus {

namespace {
const float3 kWorldUp = make_float3(0.0f, 1.0f, 0.0f);

// tan of half an angle given in degrees, with the reference's float -> double(PI) -> float promotions
float tan_half_deg(float angleDeg) { return std::tan(static_cast<float>(angleDeg / 2.0f * PI / 180.0f)); }
}  // namespace

Camera::Camera(float horizontalFOV, uint32_t width, uint32_t height)
    : m_Pose{make_float3(0.0f, 0.0f, 2.0f), make_float3(0.0f, 0.0f, -1.0f), make_float3(1.0f, 0.0f, 0.0f)}, m_Lens{horizontalFOV, 10.0f, 5.0f},
      m_Viewport{width, height}
{
}

Camera::Camera(float3 position, float3 forward, float horizontalFOV, uint32_t width, uint32_t height, float focusDistance, float defocusAngle)
    : m_Pose{position, forward, cross(forward, kWorldUp)}, m_Lens{horizontalFOV, defocusAngle, focusDistance}, m_Viewport{width, height}
{
}

void Camera::LookAt(float3 position, float3 forward)
{
    m_Pose = Pose{position, forward, cross(forward, kWorldUp)};
    m_Dirty = true;
}

void Camera::OnResize(uint32_t width, uint32_t height)
{
    if (m_Viewport[0] == width && m_Viewport[1] == height) return;
    m_Viewport[0] = width;
    m_Viewport[1] = height;
    m_Dirty = true;
}

nx_camera Camera::ToDevice(const Camera& camera)
{
    const Pose& pose = camera.m_Pose;
    const Lens& lens = camera.m_Lens;
    const float aspectRatio = camera.m_Viewport[0] / static_cast<float>(camera.m_Viewport[1]);
    const float halfWidth = lens.focusDist * tan_half_deg(lens.horizontalFovDeg);
    const float halfHeight = halfWidth / aspectRatio;
    const float3 up = cross(pose.right, pose.forward);
    const float3 viewportX = (2 * halfWidth) * pose.right;
    const float3 viewportY = (2 * halfHeight) * up;

    nx_camera d;
    std::memset(&d, 0, sizeof d);
    store(d.position, pose.position);
    store(d.right, pose.right);
    store(d.up, up);
    store(d.viewportX, viewportX);
    store(d.viewportY, viewportY);
    store(d.lowerLeftCorner, pose.position - viewportX / 2.0f - viewportY / 2.0f + pose.forward * lens.focusDist);
    d.lensRadius = lens.focusDist * tan_half_deg(lens.defocusAngleDeg);
    d.resolution[0] = camera.m_Viewport[0];
    d.resolution[1] = camera.m_Viewport[1];
    return d;
}

}  // namespace nexus
