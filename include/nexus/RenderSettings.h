// nexus/RenderSettings.h — mirrors /root/reference/Nexus/src/Renderer/RenderSettings.h:4-10; the reference reinterpret-casts
// this struct to the device POD (Scene/Scene.cpp:129), so the layout is nx_render_settings'.
#pragma once

#include "Math.h"

namespace nexus {

struct RenderSettings {
    bool useMIS = true;
    unsigned char pathLength = 10;
    float3 backgroundColor = make_float3(1.0f);
    float backgroundIntensity = 0.0f;
};
static_assert(sizeof(RenderSettings) == sizeof(nx_render_settings), "RenderSettings must alias nx_render_settings");

}  // namespace nexus
