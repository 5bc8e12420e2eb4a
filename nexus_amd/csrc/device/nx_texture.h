// nx_texture.h — software tex2D<float4>: normalised coordinates, wrap, bilinear with 8-bit fractional
// weights, sRGB decode of RGB (256-entry LUT built at context creation) before filtering.
// gfx950 has no image/texture instructions (hipcc rejects tex2D for this target), so the reference's
// cudaTextureObject fetches (/root/reference/Nexus/src/Cuda/PathTracer/PathTracer.cu:78,295,349,401 with the
// descriptor of Assets/Texture.cpp:26-33) become plain loads from linear RGBA8 buffers.
#pragma once

#include "nx_device.h"
#include "nx_math.h"

namespace nxd {

NXD int wrapi(int i, int n)
{
    i %= n;
    return i < 0 ? i + n : i;
}

NXD float4 tex2d(const TextureDev& t, const float* __restrict__ srgbLut, float u, float v)
{
    const int W = (int)t.width, H = (int)t.height;
    const float xb = u * (float)W - 0.5f, yb = v * (float)H - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    const float ax = floorf((xb - fx) * 256.0f + 0.5f) * (1.0f / 256.0f);
    const float ay = floorf((yb - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
    const int i0 = wrapi((int)fx, W), i1 = wrapi((int)fx + 1, W);
    const int j0 = wrapi((int)fy, H), j1 = wrapi((int)fy + 1, H);
    const uint32_t p00 = t.texels[(size_t)j0 * W + i0], p10 = t.texels[(size_t)j0 * W + i1];
    const uint32_t p01 = t.texels[(size_t)j1 * W + i0], p11 = t.texels[(size_t)j1 * W + i1];
    float out[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t b00 = (p00 >> (8 * c)) & 0xffu, b10 = (p10 >> (8 * c)) & 0xffu;
        const uint32_t b01 = (p01 >> (8 * c)) & 0xffu, b11 = (p11 >> (8 * c)) & 0xffu;
        float t00, t10, t01, t11;
        if (c < 3) { t00 = srgbLut[b00]; t10 = srgbLut[b10]; t01 = srgbLut[b01]; t11 = srgbLut[b11]; }
        else { t00 = (float)b00 / 255.0f; t10 = (float)b10 / 255.0f; t01 = (float)b01 / 255.0f; t11 = (float)b11 / 255.0f; }
        const float top = t00 + ax * (t10 - t00);
        const float bot = t01 + ax * (t11 - t01);
        out[c] = top + ay * (bot - top);
    }
    return make_float4(out[0], out[1], out[2], out[3]);
}

}  // namespace nxd
