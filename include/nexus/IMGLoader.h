// nexus/IMGLoader.h — image files to RGBA8 textures for the kept API surface.
// Mirrors /root/reference/Nexus/src/Assets/IMGLoader.h (IMGLoader::LoadIMG(path) / LoadIMG(embedded texture)), which wraps
// stb_image with 4 requested channels.  stb is an absent third-party dependency; this reader implements the PNG
// specification (ISO/IEC 15948) directly — zlib inflate, the five scanline filters, all colour types, 1-16 bit depth,
// palette and colour-key transparency — and yields what stbi_load(..., 4) yields: RGBA8, row 0 = top row, 16-bit samples
// reduced to their high byte.  Radiance .hdr files (RGBE, flat or run-length encoded: what the viewer's "Load HDR map" feeds
// through the same call, Renderer.cpp:104-117, Scene.cpp:93-97) are read too and reduced to 8 bits exactly as stbi_load
// does (gamma 2.2).  Not handled: Adam7 interlacing, JPEG files (an error message, never a crash).
#pragma once

#include <cstddef>
#include <string>

#include "Assets.h"

namespace nexus {

class IMGLoader {
public:
    // Throw std::runtime_error with a message on malformed / unsupported input.
    static Texture LoadIMG(const std::string& filepath);
    static Texture LoadIMG(const unsigned char* data, size_t size);
};

}  // namespace nexus
