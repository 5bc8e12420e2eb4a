// IMGLoader.cpp — image decoders behind nexus/IMGLoader.h: PNG (all colour types and bit depths, Adam7 interlacing), Radiance
// .hdr and — JPEGDecoder.cpp — JPEG, each reduced to RGBA8 exactly as stbi_load(…, 4) does (the reference's IMGLoader.cpp:17-41).
#include "nexus/IMGLoader.h"

#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <stdexcept>
#include <vector>

namespace nexus {

namespace jpeg {
Texture decode(const unsigned char* data, size_t size);  // JPEGDecoder.cpp
}

namespace {

[[noreturn]] void fail(const std::string& msg) { throw std::runtime_error("IMGLoader: " + msg); }

uint32_t be32(const unsigned char* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

}  // namespace

namespace {

// Radiance .hdr (RGBE, flat or run-length encoded scanlines) reduced to 8 bits the way stbi_load does for an HDR file
// (stb_image: stbi__hdr_load then stbi__hdr_to_ldr with gamma 2.2, scale 1): c8 = clamp(pow(c, 1 / 2.2) * 255 + 0.5), alpha 255.
Texture load_radiance_hdr(const unsigned char* data, size_t size)
{
    size_t pos = 0;
    auto line = [&]() {
        std::string l;
        while (pos < size && data[pos] != '\n') l.push_back(static_cast<char>(data[pos++]));
        if (pos < size) pos++;
        return l;
    };
    const std::string magic = line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE") fail("not a Radiance HDR file");
    bool format = false;
    for (;;) {
        if (pos >= size) fail("HDR header is not terminated");
        const std::string l = line();
        if (l.empty()) break;
        if (l == "FORMAT=32-bit_rle_rgbe") format = true;
    }
    if (!format) fail("unsupported HDR format (need 32-bit_rle_rgbe)");
    const std::string res = line();
    int h = 0, w = 0;
    if (std::sscanf(res.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || w > 32768 || h > 32768) fail("unsupported HDR orientation / size");
    std::vector<unsigned char> rgbe(static_cast<size_t>(w) * h * 4);
    for (int y = 0; y < h; y++) {
        unsigned char* row = rgbe.data() + static_cast<size_t>(y) * w * 4;
        if (w >= 8 && w < 32768 && pos + 4 <= size && data[pos] == 2 && data[pos + 1] == 2 && !(data[pos + 2] & 0x80) && ((data[pos + 2] << 8) | data[pos + 3]) == w) {
            pos += 4;  // new-style RLE: the four components of the line one after the other
            for (int c = 0; c < 4; c++) {
                int x = 0;
                while (x < w) {
                    if (pos >= size) fail("HDR data ends early");
                    int count = data[pos++];
                    if (count > 128) {  // run
                        count -= 128;
                        if (count == 0 || x + count > w || pos >= size) fail("corrupt HDR run");
                        const unsigned char v = data[pos++];
                        for (int k = 0; k < count; k++) row[4 * (x++) + c] = v;
                    } else {  // literal
                        if (count == 0 || x + count > w || pos + static_cast<size_t>(count) > size) fail("corrupt HDR literal");
                        for (int k = 0; k < count; k++) row[4 * (x++) + c] = data[pos++];
                    }
                }
            }
        } else {  // flat
            if (pos + static_cast<size_t>(w) * 4 > size) fail("HDR data ends early");
            std::memcpy(row, data + pos, static_cast<size_t>(w) * 4);
            pos += static_cast<size_t>(w) * 4;
        }
    }
    Texture tex;
    tex.width = static_cast<uint32_t>(w);
    tex.height = static_cast<uint32_t>(h);
    tex.channels = 3;
    tex.pixels.resize(static_cast<size_t>(w) * h * 4);
    for (size_t i = 0; i < static_cast<size_t>(w) * h; i++) {
        const unsigned char* p = &rgbe[4 * i];
        for (int c = 0; c < 3; c++) {
            float f = 0.0f;
            if (p[3] != 0) f = static_cast<float>(p[c]) * std::ldexp(1.0f, static_cast<int>(p[3]) - (128 + 8));
            float z = static_cast<float>(std::pow(static_cast<double>(f), static_cast<double>(1.0f / 2.2f))) * 255.0f + 0.5f;  // stb: (float)pow(x, gamma)
            if (z < 0.0f) z = 0.0f;
            if (z > 255.0f) z = 255.0f;
            tex.pixels[4 * i + static_cast<size_t>(c)] = static_cast<unsigned char>(static_cast<int>(z));
        }
        tex.pixels[4 * i + 3] = 255;
    }
    return tex;
}

}  // namespace

Texture IMGLoader::LoadIMG(const unsigned char* data, size_t size)
{
    if (size >= 2 && data[0] == '#' && data[1] == '?') return load_radiance_hdr(data, size);
    static const unsigned char kSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (size >= 3 && data[0] == 0xff && data[1] == 0xd8) return jpeg::decode(data, size);
    if (size < 8 || std::memcmp(data, kSig, 8) != 0) fail("not a PNG file");
    uint32_t width = 0, height = 0;
    int depth = 0, colour = -1;
    bool interlaced = false;
    std::vector<unsigned char> idat, palette, trns;
    bool seenHeader = false, seenEnd = false;
    for (size_t off = 8; off + 12 <= size && !seenEnd;) {
        const uint32_t len = be32(data + off);
        const unsigned char* type = data + off + 4;
        const unsigned char* body = data + off + 8;
        if (len > size - off - 12) fail("chunk runs past the end of the file");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) fail("bad IHDR");
            width = be32(body);
            height = be32(body + 4);
            depth = body[8];
            colour = body[9];
            if (body[10] != 0 || body[11] != 0) fail("unknown compression / filter method");
            if (body[12] > 1) fail("unknown interlace method");
            interlaced = body[12] == 1;
            seenHeader = true;
        } else if (!std::memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) seenEnd = true;
        else if (!(type[0] & 0x20)) fail("unknown critical chunk");  // (ancillary chunks — lower-case first letter — are skipped)
        off += 12 + static_cast<size_t>(len);
    }
    if (!seenHeader || idat.empty()) fail("no IHDR / IDAT chunk");
    if (!seenEnd) fail("the file ends before its IEND chunk");
    if (width == 0 || height == 0 || width > (1u << 15) || height > (1u << 15)) fail("unreasonable image size");
    const int samples = colour == 0 ? 1 : colour == 2 ? 3 : colour == 3 ? 1 : colour == 4 ? 2 : colour == 6 ? 4 : 0;
    if (!samples) fail("unknown colour type");
    const bool depthOk = (colour == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                         (colour == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                         ((colour == 2 || colour == 4 || colour == 6) && (depth == 8 || depth == 16));
    if (!depthOk) fail("bit depth not allowed for this colour type");
    if (colour == 3 && (palette.size() < 3 || palette.size() % 3)) fail("palette image without a valid PLTE chunk");

    const size_t bitsPerPixel = static_cast<size_t>(samples) * depth;
    const size_t bpp = std::max<size_t>(1, bitsPerPixel / 8);  // filter distance in bytes
    auto stride_of = [&](uint32_t w) { return (static_cast<size_t>(w) * bitsPerPixel + 7) / 8; };
    // the sub-images the data stream carries one after the other: the image itself, or the seven Adam7 passes
    // (pass p holds the pixels (x0 + i * dx, y0 + j * dy); empty passes carry no bytes)
    struct Pass { uint32_t x0, y0, dx, dy, w, h; };
    std::vector<Pass> passes;
    if (!interlaced) passes.push_back(Pass{0, 0, 1, 1, width, height});
    else {
        static const uint32_t kX0[7] = {0, 4, 0, 2, 0, 1, 0}, kY0[7] = {0, 0, 4, 0, 2, 0, 1}, kDx[7] = {8, 8, 4, 4, 2, 2, 1}, kDy[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int p = 0; p < 7; p++) {
            const uint32_t w = (width + kDx[p] - 1 - kX0[p]) / kDx[p], h = (height + kDy[p] - 1 - kY0[p]) / kDy[p];
            if (width > kX0[p] && height > kY0[p] && w && h) passes.push_back(Pass{kX0[p], kY0[p], kDx[p], kDy[p], w, h});
        }
    }
    size_t rawSize = 0;
    for (const Pass& p : passes) rawSize += (stride_of(p.w) + 1) * p.h;
    std::vector<unsigned char> raw(rawSize);
    uLongf rawLen = static_cast<uLongf>(raw.size());
    const int zrc = uncompress(raw.data(), &rawLen, idat.data(), static_cast<uLong>(idat.size()));
    if (zrc != Z_OK || rawLen != raw.size()) fail("corrupt image data (zlib)");

    // undo the scanline filters in place (filter byte first on every line; every pass starts with a zero line above it)
    {
        size_t off = 0;
        for (const Pass& p : passes) {
            const size_t stride = stride_of(p.w);
            std::vector<unsigned char> prev(stride, 0);
            for (uint32_t y = 0; y < p.h; y++) {
                unsigned char* line = raw.data() + off + (stride + 1) * y;
                const int filter = line[0];
                unsigned char* cur = line + 1;
                for (size_t i = 0; i < stride; i++) {
                    const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
                    int v = cur[i];
                    switch (filter) {
                    case 0: break;
                    case 1: v += a; break;
                    case 2: v += b; break;
                    case 3: v += (a + b) / 2; break;
                    case 4: v += paeth(a, b, c); break;
                    default: fail("unknown scanline filter");
                    }
                    cur[i] = static_cast<unsigned char>(v);
                }
                std::memcpy(prev.data(), cur, stride);
            }
            off += (stride + 1) * p.h;
        }
    }

    Texture tex;
    tex.width = width;
    tex.height = height;
    tex.channels = static_cast<uint32_t>(colour == 3 ? (trns.empty() ? 3 : 4) : samples + ((colour == 0 || colour == 2) && !trns.empty() ? 1 : 0));
    tex.pixels.resize(static_cast<size_t>(width) * height * 4);
    auto sample = [&](const unsigned char* row, size_t index) -> uint32_t {  // sample `index` of a row, full bit depth
        if (depth == 8) return row[index];
        if (depth == 16) return (uint32_t(row[2 * index]) << 8) | row[2 * index + 1];
        const size_t bit = index * depth;
        return (row[bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1u);
    };
    auto to8 = [&](uint32_t v) -> unsigned char {  // stb_image: 16 -> 8 keeps the high byte; 1/2/4-bit grey is scaled to 0..255
        if (depth == 16) return static_cast<unsigned char>(v >> 8);
        if (depth == 8) return static_cast<unsigned char>(v);
        return static_cast<unsigned char>(v * (255u / ((1u << depth) - 1u)));
    };
    uint32_t key[3] = {0, 0, 0};
    const bool haveKey = (colour == 0 && trns.size() >= 2) || (colour == 2 && trns.size() >= 6);
    if (haveKey)
        for (int k = 0; k < (colour == 0 ? 1 : 3); k++) key[k] = (uint32_t(trns[2 * k]) << 8) | trns[2 * k + 1];
    size_t passOff = 0;
    for (const Pass& ps : passes) {
      const size_t stride = stride_of(ps.w);
      for (uint32_t py = 0; py < ps.h; py++) {
        const unsigned char* row = raw.data() + passOff + (stride + 1) * py + 1;
        for (uint32_t x = 0; x < ps.w; x++) {
            unsigned char* out = tex.pixels.data() + (static_cast<size_t>(ps.y0 + py * ps.dy) * width + (ps.x0 + x * ps.dx)) * 4;
            switch (colour) {
            case 0: {
                const uint32_t g = sample(row, x);
                out[0] = out[1] = out[2] = to8(g);
                out[3] = haveKey && g == key[0] ? 0 : 255;
                break;
            }
            case 2: {
                const uint32_t r = sample(row, 3 * x), g = sample(row, 3 * x + 1), b = sample(row, 3 * x + 2);
                out[0] = to8(r); out[1] = to8(g); out[2] = to8(b);
                out[3] = haveKey && r == key[0] && g == key[1] && b == key[2] ? 0 : 255;
                break;
            }
            case 3: {
                const uint32_t i = sample(row, x);
                if (3 * static_cast<size_t>(i) + 2 >= palette.size()) fail("palette index out of range");
                out[0] = palette[3 * i]; out[1] = palette[3 * i + 1]; out[2] = palette[3 * i + 2];
                out[3] = i < trns.size() ? trns[i] : 255;
                break;
            }
            case 4:
                out[0] = out[1] = out[2] = to8(sample(row, 2 * x));
                out[3] = to8(sample(row, 2 * x + 1));
                break;
            default:
                for (int k = 0; k < 4; k++) out[k] = to8(sample(row, 4 * x + k));
            }
        }
      }
      passOff += (stride + 1) * ps.h;
    }
    return tex;
}

Texture IMGLoader::LoadIMG(const std::string& filepath)
{
    std::ifstream f(filepath, std::ios::binary);
    if (!f) fail("cannot open " + filepath);
    const std::vector<unsigned char> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return LoadIMG(data.data(), data.size());
}

}  // namespace nexus
