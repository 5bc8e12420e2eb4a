// Renderer.cpp — see include/nexus/Renderer.h.
#include "nexus/Renderer.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>

namespace nexus {

Renderer::Renderer(uint32_t width, uint32_t height, Scene* scene, int device)
    : m_ViewportWidth(width), m_ViewportHeight(height), m_Scene(scene), m_PathTracer(width, height, device)
{
}

void Renderer::Reset()
{
    m_PathTracer.ResetFrameNumber();
    m_AccumulatedTime = 0.0;
    m_Frames = 0;
}

void Renderer::OnResize(uint32_t width, uint32_t height)
{
    if ((m_ViewportWidth != width || m_ViewportHeight != height) && width != 0 && height != 0) {
        m_PathTracer.OnResize(width, height);
        m_AccumulatedTime = 0.0;
        m_Frames = 0;
        m_ViewportWidth = width;
        m_ViewportHeight = height;
    }
}

void Renderer::Render(Scene& scene, float deltaTime)
{
    m_AccumulatedTime += deltaTime;
    if (scene.IsInvalid()) {
        scene.Update();
        m_PathTracer.ResetFrameNumber();
    }
    if (!scene.IsEmpty()) {
        m_PathTracer.UpdateDeviceScene(*m_Scene);
        m_PathTracer.Render(scene);
        m_Frames++;
    } else {
        m_PathTracer.ResetFrameNumber();
    }
}

bool Renderer::SaveScreenshot(const std::string& filepath)
{
    std::string path = filepath;
    const std::string extension = ".png";
    if (path.length() < extension.length() || path.compare(path.size() - extension.size(), extension.size(), extension) != 0) path += extension;
    const std::vector<uint32_t>& pixels = m_PathTracer.GetPixelBuffer();
    return WritePNG(path, pixels.data(), m_ViewportWidth, m_ViewportHeight, true);
}

bool Renderer::SaveAccumulationEXR(const std::string& filepath)
{
    std::vector<float> rgb(static_cast<size_t>(m_ViewportWidth) * m_ViewportHeight * 3);
    if (nxhip_read_accumulation(m_PathTracer.GetDeviceContext(), rgb.data()) != NXHIP_OK) return false;
    return WriteEXR(filepath, rgb.data(), m_ViewportWidth, m_ViewportHeight, true);
}

namespace {

void put_be32(std::vector<unsigned char>& v, uint32_t x)
{
    v.push_back(static_cast<unsigned char>(x >> 24)); v.push_back(static_cast<unsigned char>(x >> 16));
    v.push_back(static_cast<unsigned char>(x >> 8)); v.push_back(static_cast<unsigned char>(x));
}

void png_chunk(std::vector<unsigned char>& out, const char* tag, const unsigned char* data, size_t n)
{
    put_be32(out, static_cast<uint32_t>(n));
    const size_t start = out.size();
    out.insert(out.end(), tag, tag + 4);
    out.insert(out.end(), data, data + n);
    put_be32(out, static_cast<uint32_t>(crc32(0L, out.data() + start, static_cast<uInt>(n + 4))));
}

template <typename T>
void put_le(std::vector<unsigned char>& v, T x)
{
    unsigned char b[sizeof(T)];
    std::memcpy(b, &x, sizeof(T));  // the targets are little endian
    v.insert(v.end(), b, b + sizeof(T));
}

void exr_attr(std::vector<unsigned char>& v, const char* name, const char* type, const std::vector<unsigned char>& value)
{
    v.insert(v.end(), name, name + std::strlen(name) + 1);
    v.insert(v.end(), type, type + std::strlen(type) + 1);
    put_le<int32_t>(v, static_cast<int32_t>(value.size()));
    v.insert(v.end(), value.begin(), value.end());
}

bool write_file(const std::string& path, const std::vector<unsigned char>& bytes)
{
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    return std::fclose(f) == 0 && ok;
}

}  // namespace

bool WritePNG(const std::string& path, const uint32_t* rgba8, uint32_t width, uint32_t height, bool flipVertically)
{
    if (!rgba8 || width == 0 || height == 0) return false;
    std::vector<unsigned char> raw;
    raw.reserve((static_cast<size_t>(width) * 4 + 1) * height);
    for (uint32_t y = 0; y < height; y++) {
        const uint32_t* row = rgba8 + static_cast<size_t>(flipVertically ? height - 1 - y : y) * width;
        raw.push_back(0);  // filter: none
        const unsigned char* b = reinterpret_cast<const unsigned char*>(row);  // R in the low byte: R G B A in memory
        raw.insert(raw.end(), b, b + static_cast<size_t>(width) * 4);
    }
    uLongf bound = compressBound(static_cast<uLong>(raw.size()));
    std::vector<unsigned char> packed(bound);
    if (compress2(packed.data(), &bound, raw.data(), static_cast<uLong>(raw.size()), 6) != Z_OK) return false;
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    const unsigned char tail[5] = {8, 6, 0, 0, 0};  // 8 bits, RGBA, deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    png_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    png_chunk(out, "IDAT", packed.data(), bound);
    png_chunk(out, "IEND", nullptr, 0);
    return write_file(path, out);
}

bool WriteEXR(const std::string& path, const float* rgb, uint32_t width, uint32_t height, bool flipVertically)
{
    if (!rgb || width == 0 || height == 0) return false;
    std::vector<unsigned char> out;
    put_le<uint32_t>(out, 20000630u);  // magic
    put_le<uint32_t>(out, 2u);         // version 2, single-part scanline
    std::vector<unsigned char> v;
    for (const char* ch : {"B", "G", "R"}) {  // channel list, alphabetical: name, pixel type FLOAT = 2, pLinear + reserved, sampling 1 x 1
        v.insert(v.end(), ch, ch + 2);
        put_le<int32_t>(v, 2);
        put_le<uint32_t>(v, 0u);
        put_le<int32_t>(v, 1);
        put_le<int32_t>(v, 1);
    }
    v.push_back(0);
    exr_attr(out, "channels", "chlist", v);
    exr_attr(out, "compression", "compression", {0});  // NO_COMPRESSION
    v.clear();
    for (int32_t x : {0, 0, static_cast<int32_t>(width) - 1, static_cast<int32_t>(height) - 1}) put_le<int32_t>(v, x);
    exr_attr(out, "dataWindow", "box2i", v);
    exr_attr(out, "displayWindow", "box2i", v);
    exr_attr(out, "lineOrder", "lineOrder", {0});  // increasing y
    v.clear();
    put_le<float>(v, 1.0f);
    exr_attr(out, "pixelAspectRatio", "float", v);
    v.clear();
    put_le<float>(v, 0.0f);
    put_le<float>(v, 0.0f);
    exr_attr(out, "screenWindowCenter", "v2f", v);
    v.clear();
    put_le<float>(v, 1.0f);
    exr_attr(out, "screenWindowWidth", "float", v);
    out.push_back(0);  // end of header
    const size_t lineBytes = static_cast<size_t>(width) * 3 * 4, chunk = 8 + lineBytes;
    const uint64_t first = out.size() + static_cast<uint64_t>(height) * 8;
    for (uint32_t y = 0; y < height; y++) put_le<uint64_t>(out, first + static_cast<uint64_t>(y) * chunk);
    out.reserve(out.size() + chunk * height);
    for (uint32_t y = 0; y < height; y++) {
        const float* row = rgb + static_cast<size_t>(flipVertically ? height - 1 - y : y) * width * 3;
        put_le<int32_t>(out, static_cast<int32_t>(y));
        put_le<int32_t>(out, static_cast<int32_t>(lineBytes));
        for (int c = 2; c >= 0; c--)  // B, G, R planes of the line
            for (uint32_t x = 0; x < width; x++) put_le<float>(out, row[3 * static_cast<size_t>(x) + static_cast<size_t>(c)]);
    }
    return write_file(path, out);
}

}  // namespace nexus
