// JPEGDecoder.cpp — baseline / extended-sequential / progressive Huffman JPEG (ITU-T T.81, 8 bits per sample) to RGBA8, for
// glTF images and texture files: the reference decodes them with stbi_load(…, 4) (/root/reference/Nexus/src/Assets/
// IMGLoader.cpp:17-41 -> vendor/stb/stb_image.h), so the pixel values here are stb_image's, bit for bit:
//   * inverse DCT: the 13-bit fixed-point "islow" butterfly with stb's constants and rounding (stb_image.h:2437-2520),
//     columns first with two guard bits, an all-zero-AC column short cut, +128 level shift folded into the row pass;
//   * chroma upsampling: per output row, the 3:1 / 9:3:3:1 tent filters for 2x factors, nearest for others, with stb's
//     edge rules and its choice of the "near" and "far" source row (stb_image.h:3455-3560, 3922-3940);
//   * YCbCr -> RGB: 20-bit fixed point with the chroma-blue term of green truncated to 16 bits (stb_image.h:3655-3680);
//   * component semantics: 1 component = grey, 3 = YCbCr unless the ids are 'R','G','B' or an Adobe marker says transform 0
//     without a JFIF marker, 4 = CMYK / YCCK through the Adobe transform.
// tests/test_image_decoders.py compares this decoder with stb_image itself (a test-side build of the reference's vendored
// header) on files of every kind, intact and damaged, and with committed stb outputs.  Entropy decoding is written from T.81 (canonical Huffman
// codes by length, EXTEND, restart intervals, spectral selection / successive approximation with EOB runs).
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "nexus/Assets.h"

namespace nexus {
namespace jpeg {

namespace {

[[noreturn]] void fail(const std::string& msg) { throw std::runtime_error("IMGLoader: JPEG: " + msg); }

// zig-zag position -> natural (row-major) position.  Fifteen more entries, all 63: a corrupt run length can step past the end
// of a block, and such a file is to decode to the same (garbage) pixels as through stb_image, which lets the index run on
constexpr uint8_t kNatural[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                       41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                       30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                       63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct HuffTable {
    bool defined = false;
    // canonical code per T.81 annex C / F.2.2.3: for each length l, codes [minCode[l], maxCode[l]] map to values[valPtr[l] ...]
    int32_t minCode[17], maxCode[17], valPtr[17];
    uint8_t values[256];
    // AC tables: for every 9-bit window whose leading code word AND the magnitude bits behind it fit into the window (and whose
    // value fits a signed byte), the decoded coefficient at once: value * 256 + run * 16 + bits consumed; 0 = no such short cut.
    // stb_image decodes such symbols in one step and treats a shortage of bits differently on this path than on the general
    // one (see BitReader), so which symbols take it is part of the behaviour to reproduce on damaged files.
    int16_t shortcut[512];
    // (symbol, length) of the code word at the top of a 16-bit window; length 0: none
    bool lookup(uint32_t window16, int& symbol, int& length) const
    {
        int32_t code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | static_cast<int32_t>((window16 >> (16 - l)) & 1u);
            if (maxCode[l] >= 0 && code <= maxCode[l] && code >= minCode[l]) {
                symbol = values[valPtr[l] + (code - minCode[l])];
                length = l;
                return true;
            }
        }
        length = 0;
        return false;
    }
    void build(const uint8_t counts[16], const uint8_t* vals, int n, bool isAc)
    {
        std::memcpy(values, vals, static_cast<size_t>(n));
        int32_t code = 0;
        int k = 0;
        for (int l = 1; l <= 16; l++) {
            valPtr[l] = k;
            minCode[l] = code;
            code += counts[l - 1];
            k += counts[l - 1];
            maxCode[l] = counts[l - 1] ? code - 1 : -1;
            if (counts[l - 1] && code - 1 >= (1 << l)) fail("Huffman code lengths over-subscribe the code space");
            code <<= 1;
        }
        defined = true;
        std::memset(shortcut, 0, sizeof shortcut);
        if (!isAc) return;
        for (int w = 0; w < 512; w++) {
            int sym = 0, len = 0;
            if (!lookup(static_cast<uint32_t>(w) << 7, sym, len) || len > 9) continue;
            const int run = sym >> 4, mag = sym & 15;
            if (!mag || len + mag > 9) continue;
            int v = ((w << len) & 511) >> (9 - mag);
            if (v < (1 << (mag - 1))) v -= (1 << mag) - 1;
            if (v >= -128 && v <= 127) shortcut[w] = static_cast<int16_t>(v * 256 + run * 16 + len + mag);
        }
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;   // Huffman tables of the current scan
    int width = 0, height = 0;    // samples that belong to the image
    int planeW = 0, planeH = 0;   // allocated: whole MCUs
    int blocksW = 0, blocksH = 0;
    int dcPred = 0;
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff;   // progressive: all coefficients, natural order, 64 per block
};

// Entropy-coded segment reader: bytes of the scan with 0xFF00 unstuffed, fill bytes skipped.  Behaviour at the edges follows
// stb_image (stbi__grow_buffer_unsafe and its callers), because a damaged file is to decode to the same pixels or fail alike:
//   * beyond the end of the file every byte reads as zero;
//   * the refill that runs into a marker stops there WITHOUT adding bits; every later refill feeds zero bytes;
//   * a code word longer than the bits at hand is an error; magnitude bits that are not at hand (after one refill) read as a
//     zero value; on the short-cut path (HuffTable::shortcut) code word + magnitude together must be at hand.
struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int bits = 0;
    int marker = -1;
    bool noMore = false;
    void reset() { acc = 0; bits = 0; marker = -1; noMore = false; }
    int byte() { return p < end ? *p++ : 0; }
    void refill()
    {
        do {
            const uint32_t b = noMore ? 0u : static_cast<uint32_t>(byte());
            if (b == 0xff) {
                int c = byte();
                while (c == 0xff) c = byte();  // fill bytes
                if (c != 0) {
                    marker = c;
                    noMore = true;
                    return;
                }
            }
            acc |= b << (24 - bits);
            bits += 8;
        } while (bits <= 24);
    }
    void want16() { if (bits < 16) refill(); }
    int bit()
    {
        if (bits < 1) refill();
        if (bits < 1) return 0;
        const int b = static_cast<int>(acc >> 31);
        acc <<= 1;
        bits--;
        return b;
    }
    int take(int n)  // n in [1, 16]
    {
        if (bits < n) refill();
        if (bits < n) return 0;
        const int v = static_cast<int>(acc >> (32 - n));
        acc <<= n;
        bits -= n;
        return v;
    }
    int decode(const HuffTable& t)
    {
        want16();
        int sym = 0, len = 0;
        if (!t.lookup(acc >> 16, sym, len)) fail("bad Huffman code");
        if (len > bits) fail("bad Huffman code");
        acc <<= len;
        bits -= len;
        return sym;
    }
    // RECEIVE + EXTEND (T.81 F.2.2.1): an n-bit magnitude category value
    int receiveExtend(int n)
    {
        if (n == 0) return 0;
        if (bits < n) refill();
        if (bits < n) return 0;
        const int v = static_cast<int>(acc >> (32 - n));
        acc <<= n;
        bits -= n;
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    // the short cut of an AC table for the window at the cursor: 0 = none, else value * 256 + run * 16 + bits (consumed here)
    int shortcut(const HuffTable& t)
    {
        const int r = t.shortcut[acc >> 23];
        if (!r) return 0;
        const int n = r & 15;
        if (n > bits) fail("bad Huffman code");
        acc <<= n;
        bits -= n;
        return r;
    }
};

inline uint8_t clamp8(int x) { return static_cast<uint8_t>(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// ---- inverse DCT, stb_image's arithmetic --------------------------------------------------------------------------------
constexpr int fx(double x) { return static_cast<int>(x * 4096 + 0.5); }

struct Butterfly {
    int x0, x1, x2, x3, t0, t1, t2, t3;
};
// 32-bit two's-complement arithmetic spelled without signed overflow: on a well-formed file nothing here leaves the int range,
// on a damaged one stb_image's int arithmetic wraps, and the same (garbage) pixels are to come out here — defined behaviour
using W = long long;
inline int wrap(W v) { return static_cast<int>(static_cast<uint32_t>(static_cast<unsigned long long>(v))); }
// the 1-D even / odd part of the islow IDCT on eight inputs, results scaled by 4096
inline Butterfly idct1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7)
{
    Butterfly r;
    int p1 = wrap((W(s2) + s6) * fx(0.5411961f));
    const int e2 = wrap(W(p1) + W(s6) * fx(-1.847759065f));
    const int e3 = wrap(W(p1) + W(s2) * fx(0.765366865f));
    const int e0 = wrap((W(s0) + s4) * 4096), e1 = wrap((W(s0) - s4) * 4096);
    r.x0 = wrap(W(e0) + e3);
    r.x3 = wrap(W(e0) - e3);
    r.x1 = wrap(W(e1) + e2);
    r.x2 = wrap(W(e1) - e2);
    int t0 = s7, t1 = s5, t2 = s3, t3 = s1;
    int p3 = wrap(W(t0) + t2), p4 = wrap(W(t1) + t3);
    p1 = wrap(W(t0) + t3);
    int p2 = wrap(W(t1) + t2);
    const int p5 = wrap((W(p3) + p4) * fx(1.175875602f));
    t0 = wrap(W(t0) * fx(0.298631336f));
    t1 = wrap(W(t1) * fx(2.053119869f));
    t2 = wrap(W(t2) * fx(3.072711026f));
    t3 = wrap(W(t3) * fx(1.501321110f));
    p1 = wrap(W(p5) + W(p1) * fx(-0.899976223f));
    p2 = wrap(W(p5) + W(p2) * fx(-2.562915447f));
    p3 = wrap(W(p3) * fx(-1.961570560f));
    p4 = wrap(W(p4) * fx(-0.390180644f));
    r.t3 = wrap(W(t3) + p1 + p4);
    r.t2 = wrap(W(t2) + p2 + p3);
    r.t1 = wrap(W(t1) + p2 + p4);
    r.t0 = wrap(W(t0) + p1 + p3);
    return r;
}

void idct8x8(uint8_t* out, int stride, const int16_t d[64])
{
    int tmp[64];
    for (int c = 0; c < 8; c++) {
        const int16_t* s = d + c;
        int* v = tmp + c;
        if (!s[8] && !s[16] && !s[24] && !s[32] && !s[40] && !s[48] && !s[56]) {
            const int dc = s[0] * 4;
            for (int r = 0; r < 8; r++) v[8 * r] = dc;
            continue;
        }
        Butterfly b = idct1d(s[0], s[8], s[16], s[24], s[32], s[40], s[48], s[56]);
        const W round = 512;  // keep two extra bits through the row pass
        v[0] = wrap(W(b.x0) + round + b.t3) >> 10;
        v[56] = wrap(W(b.x0) + round - b.t3) >> 10;
        v[8] = wrap(W(b.x1) + round + b.t2) >> 10;
        v[48] = wrap(W(b.x1) + round - b.t2) >> 10;
        v[16] = wrap(W(b.x2) + round + b.t1) >> 10;
        v[40] = wrap(W(b.x2) + round - b.t1) >> 10;
        v[24] = wrap(W(b.x3) + round + b.t0) >> 10;
        v[32] = wrap(W(b.x3) + round - b.t0) >> 10;
    }
    for (int r = 0; r < 8; r++) {
        const int* v = tmp + 8 * r;
        uint8_t* o = out + static_cast<size_t>(r) * stride;
        Butterfly b = idct1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        const W bias = 65536 + (128 << 17);  // rounding of the 17-bit shift + the level shift
        o[0] = clamp8(wrap(W(b.x0) + bias + b.t3) >> 17);
        o[7] = clamp8(wrap(W(b.x0) + bias - b.t3) >> 17);
        o[1] = clamp8(wrap(W(b.x1) + bias + b.t2) >> 17);
        o[6] = clamp8(wrap(W(b.x1) + bias - b.t2) >> 17);
        o[2] = clamp8(wrap(W(b.x2) + bias + b.t1) >> 17);
        o[5] = clamp8(wrap(W(b.x2) + bias - b.t1) >> 17);
        o[3] = clamp8(wrap(W(b.x3) + bias + b.t0) >> 17);
        o[4] = clamp8(wrap(W(b.x3) + bias - b.t0) >> 17);
    }
}

// ---- chroma upsampling, one output row at a time ------------------------------------------------------------------------
// `w` = samples of the low-resolution row that belong to the image; the rows are plane rows, so reading is safe
const uint8_t* upsample_row(uint8_t* out, const uint8_t* nearRow, const uint8_t* farRow, int w, int hs, int vs)
{
    if (hs == 1 && vs == 1) return nearRow;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; i++) out[i] = static_cast<uint8_t>((3 * nearRow[i] + farRow[i] + 2) >> 2);
        return out;
    }
    if (hs == 2 && vs == 1) {
        const uint8_t* in = nearRow;
        if (w == 1) {
            out[0] = out[1] = in[0];
            return out;
        }
        out[0] = in[0];
        out[1] = static_cast<uint8_t>((in[0] * 3 + in[1] + 2) >> 2);
        int i = 1;
        for (; i < w - 1; i++) {
            const int n = 3 * in[i] + 2;
            out[2 * i] = static_cast<uint8_t>((n + in[i - 1]) >> 2);
            out[2 * i + 1] = static_cast<uint8_t>((n + in[i + 1]) >> 2);
        }
        out[2 * i] = static_cast<uint8_t>((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[2 * i + 1] = in[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) {
            out[0] = out[1] = static_cast<uint8_t>((3 * nearRow[0] + farRow[0] + 2) >> 2);
            return out;
        }
        int t1 = 3 * nearRow[0] + farRow[0];
        out[0] = static_cast<uint8_t>((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int t0 = t1;
            t1 = 3 * nearRow[i] + farRow[i];
            out[2 * i - 1] = static_cast<uint8_t>((3 * t0 + t1 + 8) >> 4);
            out[2 * i] = static_cast<uint8_t>((3 * t1 + t0 + 8) >> 4);
        }
        out[2 * w - 1] = static_cast<uint8_t>((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++)
        for (int j = 0; j < hs; j++) out[i * hs + j] = nearRow[i];
    return out;
}

inline uint8_t mul8(uint8_t x, uint8_t y)  // x * y / 255, Blinn's rounding
{
    const unsigned t = static_cast<unsigned>(x) * y + 128;
    return static_cast<uint8_t>((t + (t >> 8)) >> 8);
}

constexpr int chroma_fixed(double x) { return static_cast<int>(x * 4096.0 + 0.5) << 8; }

void ycc_to_rgba(uint8_t* out, const uint8_t* y, const uint8_t* cbRow, const uint8_t* crRow, int count)
{
    for (int i = 0; i < count; i++, out += 4) {
        const int yf = (y[i] << 20) + (1 << 19);
        const int cr = crRow[i] - 128, cb = cbRow[i] - 128;
        int r = yf + cr * chroma_fixed(1.40200f);
        int g = yf + cr * -chroma_fixed(0.71414f) + static_cast<int>(static_cast<unsigned>(cb * -chroma_fixed(0.34414f)) & 0xffff0000u);
        int b = yf + cb * chroma_fixed(1.77200f);
        out[0] = clamp8(r >> 20);
        out[1] = clamp8(g >> 20);
        out[2] = clamp8(b >> 20);
        out[3] = 255;
    }
}

struct Decoder {
    const uint8_t* data;
    size_t size;
    size_t pos = 0;
    int width = 0, height = 0, nComp = 0;
    bool progressive = false, sawFrame = false;
    bool jfif = false;
    int adobeTransform = -1;
    int rgbIds = 0;
    int hMax = 1, vMax = 1, mcuX = 0, mcuY = 0;
    int restartInterval = 0;
    uint16_t quant[4][64];
    bool quantDefined[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    Component comp[4];
    // scan
    int scanN = 0, order[4];
    int ss = 0, se = 63, ah = 0, al = 0;
    int eobRun = 0;

    Decoder(const uint8_t* d, size_t n) : data(d), size(n) {}

    bool eof() const { return pos >= size; }
    int u8() { return pos < size ? data[pos++] : 0; }  // beyond the end of the file every byte reads as zero (stb_image's stbi__get8)
    int u16()
    {
        const int a = u8();
        return (a << 8) | u8();
    }
    static constexpr int kNoMarker = 0xff;
    int nextMarker()  // the marker at the cursor, kNoMarker when the cursor is not at one
    {
        int c = u8();
        if (c != 0xff) return kNoMarker;
        while (c == 0xff) c = u8();
        return c;
    }

    void tables(int m)
    {
        if (m == kNoMarker) fail("expected a marker");
        if (m == 0xdd) {  // DRI
            if (u16() != 4) fail("bad DRI length");
            restartInterval = u16();
            return;
        }
        if (m == 0xdb) {  // DQT
            int len = u16() - 2;
            while (len > 0) {
                const int q = u8(), precision = q >> 4, t = q & 15;
                if (precision > 1) fail("bad DQT precision");
                if (t > 3) fail("bad DQT table id");
                for (int i = 0; i < 64; i++) quant[t][kNatural[i]] = static_cast<uint16_t>(precision ? u16() : u8());
                quantDefined[t] = true;
                len -= precision ? 129 : 65;
            }
            if (len != 0) fail("bad DQT length");
            return;
        }
        if (m == 0xc4) {  // DHT
            int len = u16() - 2;
            while (len > 0) {
                const int q = u8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) fail("bad DHT header");
                uint8_t counts[16];
                int n = 0;
                for (int i = 0; i < 16; i++) {
                    counts[i] = static_cast<uint8_t>(u8());
                    n += counts[i];
                }
                if (n > 256) fail("bad DHT header");
                uint8_t vals[256];
                for (int i = 0; i < n; i++) vals[i] = static_cast<uint8_t>(u8());
                (tc ? ac[th] : dc[th]).build(counts, vals, n, tc != 0);
                len -= 17 + n;
            }
            if (len != 0) fail("bad DHT length");
            return;
        }
        if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) {  // APPn, COM
            int len = u16();
            if (len < 2) fail("bad APP / COM length");
            len -= 2;
            if (m == 0xe0 && len >= 5) {
                static const char tag[5] = {'J', 'F', 'I', 'F', 0};
                bool ok = true;
                for (int i = 0; i < 5; i++) ok = (u8() == static_cast<uint8_t>(tag[i])) && ok;
                len -= 5;
                if (ok) jfif = true;
            } else if (m == 0xee && len >= 12) {
                static const char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
                bool ok = true;
                for (int i = 0; i < 6; i++) ok = (u8() == static_cast<uint8_t>(tag[i])) && ok;
                len -= 6;
                if (ok) {
                    u8(); u16(); u16();
                    adobeTransform = u8();
                    len -= 6;
                }
            }
            if (static_cast<size_t>(len) > size - pos) fail("segment runs past the end of the file");
            pos += static_cast<size_t>(len);
            return;
        }
        fail("unknown marker");
    }

    void frame(int m)
    {
        progressive = m == 0xc2;
        const int len = u16();
        if (len < 11) fail("bad SOF length");
        if (u8() != 8) fail("only 8 bits per sample are supported");
        height = u16();
        width = u16();
        if (height == 0) fail("no image height in the frame header");
        if (width == 0) fail("zero width");
        if (width > (1 << 24) || height > (1 << 24)) fail("image too large");
        nComp = u8();
        if (nComp != 1 && nComp != 3 && nComp != 4) fail("bad component count");
        if (len != 8 + 3 * nComp) fail("bad SOF length");
        rgbIds = 0;
        for (int i = 0; i < nComp; i++) {
            Component& c = comp[i];
            c.id = u8();
            if (nComp == 3 && c.id == "RGB"[i]) rgbIds++;
            const int q = u8();
            c.h = q >> 4;
            c.v = q & 15;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) fail("bad sampling factor");
            c.tq = u8();
            if (c.tq > 3) fail("bad quantisation table id");
            hMax = c.h > hMax ? c.h : hMax;
            vMax = c.v > vMax ? c.v : vMax;
        }
        for (int i = 0; i < nComp; i++)
            if (hMax % comp[i].h || vMax % comp[i].v) fail("fractional sampling ratios are not supported");
        if (static_cast<uint64_t>(width) * height > (1ull << 28)) fail("image too large");
        mcuX = (width + 8 * hMax - 1) / (8 * hMax);
        mcuY = (height + 8 * vMax - 1) / (8 * vMax);
        for (int i = 0; i < nComp; i++) {
            Component& c = comp[i];
            c.width = (width * c.h + hMax - 1) / hMax;
            c.height = (height * c.v + vMax - 1) / vMax;
            c.blocksW = mcuX * c.h;
            c.blocksH = mcuY * c.v;
            c.planeW = c.blocksW * 8;
            c.planeH = c.blocksH * 8;
            c.plane.assign(static_cast<size_t>(c.planeW) * c.planeH, 0);
            if (progressive) c.coeff.assign(static_cast<size_t>(c.blocksW) * c.blocksH * 64, 0);
        }
        sawFrame = true;
    }

    void scanHeader()
    {
        const int len = u16();
        scanN = u8();
        if (scanN < 1 || scanN > 4 || scanN > nComp) fail("bad SOS component count");
        if (len != 6 + 2 * scanN) fail("bad SOS length");
        for (int i = 0; i < scanN; i++) {
            const int id = u8(), q = u8();
            int which = 0;
            while (which < nComp && comp[which].id != id) which++;
            if (which == nComp) fail("SOS names an unknown component");
            comp[which].td = q >> 4;
            comp[which].ta = q & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) fail("bad Huffman table id");
            order[i] = which;
        }
        ss = u8();
        se = u8();
        const int a = u8();
        ah = a >> 4;
        al = a & 15;
        if (progressive) {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) fail("bad SOS parameters");
        } else {
            if (ss != 0 || ah != 0 || al != 0) fail("bad SOS parameters");
            se = 63;
        }
    }

    // ---- block decoders -------------------------------------------------------------------------------------------------
    static int16_t to16(long long v, const char* what)
    {
        if (v < -32768 || v > 32767) fail(what);
        return static_cast<int16_t>(v);
    }

    void blockSequential(BitReader& br, Component& c, int16_t out[64])
    {
        if (!dc[c.td].defined || !ac[c.ta].defined) fail("scan uses an undefined Huffman table");
        br.want16();
        const int t = br.decode(dc[c.td]);
        if (t > 15) fail("bad DC magnitude category");
        std::memset(out, 0, 64 * sizeof(int16_t));
        const int diff = br.receiveExtend(t);
        const long long pred = static_cast<long long>(c.dcPred) + diff;
        if (pred < INT32_MIN || pred > INT32_MAX) fail("bad DC delta");
        c.dcPred = static_cast<int>(pred);
        const uint16_t* q = quant[c.tq];
        out[0] = to16(static_cast<long long>(c.dcPred) * static_cast<int>(q[0]), "DC coefficient out of range");
        const HuffTable& t2 = ac[c.ta];
        for (int k = 1; k < 64;) {
            br.want16();
            if (const int sc = br.shortcut(t2)) {
                k += (sc >> 4) & 15;
                const int nat = kNatural[k++];
                out[nat] = static_cast<int16_t>((sc >> 8) * static_cast<int>(q[nat]));
                continue;
            }
            const int rs = br.decode(t2);
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (rs != 0xf0) break;  // end of block
                k += 16;
            } else {
                k += r;
                const int nat = kNatural[k++];
                out[nat] = static_cast<int16_t>(br.receiveExtend(s) * static_cast<int>(q[nat]));
            }
        }
    }

    void blockProgressiveDC(BitReader& br, Component& c, int16_t* blk)
    {
        if (se != 0) fail("a DC scan may not carry AC coefficients");
        br.want16();
        if (ah == 0) {
            if (!dc[c.td].defined) fail("scan uses an undefined Huffman table");
            std::memset(blk, 0, 64 * sizeof(int16_t));
            const int t = br.decode(dc[c.td]);
            if (t > 15) fail("bad DC magnitude category");
            const int diff = br.receiveExtend(t);
            const long long pred = static_cast<long long>(c.dcPred) + diff;
            if (pred < INT32_MIN || pred > INT32_MAX) fail("bad DC delta");
            c.dcPred = static_cast<int>(pred);
            blk[0] = to16(static_cast<long long>(c.dcPred) * (1 << al), "DC coefficient out of range");
        } else if (br.bit()) {
            blk[0] = static_cast<int16_t>(blk[0] + static_cast<int16_t>(1 << al));
        }
    }

    void blockProgressiveAC(BitReader& br, Component& c, int16_t* blk)
    {
        if (ss == 0) fail("an AC scan may not carry the DC coefficient");
        if (!ac[c.ta].defined) fail("scan uses an undefined Huffman table");
        const HuffTable& t = ac[c.ta];
        if (ah == 0) {
            if (eobRun) {
                eobRun--;
                return;
            }
            for (int k = ss; k <= se;) {
                br.want16();
                if (const int sc = br.shortcut(t)) {
                    k += (sc >> 4) & 15;
                    blk[kNatural[k++]] = static_cast<int16_t>((sc >> 8) * (1 << al));
                    continue;
                }
                const int rs = br.decode(t);
                const int r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) {
                        eobRun = (1 << r) - 1;
                        if (r) eobRun += br.take(r);
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    blk[kNatural[k++]] = static_cast<int16_t>(br.receiveExtend(s) * (1 << al));
                }
            }
            return;
        }
        // refinement of coefficients that are already non-zero, new +-1 coefficients in between (T.81 G.1.2.3)
        const int16_t bit = static_cast<int16_t>(1 << al);
        auto refine = [&](int16_t& v) {
            if (br.bit() && (v & bit) == 0) v = static_cast<int16_t>(v > 0 ? v + bit : v - bit);
        };
        if (eobRun) {
            eobRun--;
            for (int k = ss; k <= se; k++) {
                int16_t& v = blk[kNatural[k]];
                if (v != 0) refine(v);
            }
            return;
        }
        int k = ss;
        do {
            const int rs = br.decode(t);
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobRun = (1 << r) - 1;
                    if (r) eobRun += br.take(r);
                    r = 64;  // nothing new in the rest of the band: only refinements
                }
            } else {
                if (s != 1) fail("bad refinement code");
                s = br.bit() ? bit : -bit;
            }
            while (k <= se) {
                int16_t& v = blk[kNatural[k++]];
                if (v != 0) refine(v);
                else {
                    if (r == 0) {
                        v = static_cast<int16_t>(s);
                        break;
                    }
                    r--;
                }
            }
        } while (k <= se);
    }

    void scan()
    {
        BitReader br{data + pos, data + size};
        for (int i = 0; i < nComp; i++) comp[i].dcPred = 0;
        eobRun = 0;
        int todo = restartInterval ? restartInterval : 0x7fffffff;
        int16_t tmp[64];
        auto one = [&](Component& c, int bx, int by) {
            if (progressive) {
                int16_t* blk = c.coeff.data() + 64 * (static_cast<size_t>(by) * c.blocksW + bx);
                if (ss == 0) blockProgressiveDC(br, c, blk);
                else blockProgressiveAC(br, c, blk);
            } else {
                if (!quantDefined[c.tq]) fail("component uses an undefined quantisation table");
                blockSequential(br, c, tmp);
                idct8x8(c.plane.data() + static_cast<size_t>(by) * 8 * c.planeW + static_cast<size_t>(bx) * 8, c.planeW, tmp);
            }
        };
        auto restart = [&]() -> bool {  // after every restart interval: RSTn expected, predictors and bit reader start over
            if (--todo > 0) return true;
            if (br.bits < 24) br.refill();
            if (br.marker < 0xd0 || br.marker > 0xd7) return false;  // no restart marker: the scan ends here
            br.reset();
            for (int i = 0; i < nComp; i++) comp[i].dcPred = 0;
            eobRun = 0;
            todo = restartInterval;
            return true;
        };
        bool more = true;
        if (scanN == 1) {
            // non-interleaved: the component's own blocks, only those that cover the image
            Component& c = comp[order[0]];
            const int w = (c.width + 7) >> 3, h = (c.height + 7) >> 3;
            if (progressive && ss != 0 && scanN != 1) fail("AC scans carry one component");
            for (int by = 0; by < h && more; by++)
                for (int bx = 0; bx < w && more; bx++) {
                    one(c, bx, by);
                    more = restart();
                }
        } else {
            if (progressive && ss != 0) fail("AC scans carry one component");
            for (int my = 0; my < mcuY && more; my++)
                for (int mx = 0; mx < mcuX && more; mx++) {
                    for (int k = 0; k < scanN; k++) {
                        Component& c = comp[order[k]];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++) one(c, mx * c.h + x, my * c.v + y);
                    }
                    more = restart();
                }
        }
        // The file cursor goes behind what the entropy decoder has consumed; the marker it ran into (if any) is the next one.
        pos = static_cast<size_t>(br.p - data);
        pendingMarker = br.marker >= 0 ? br.marker : kNoMarker;
        if (pendingMarker == kNoMarker) {
            // bytes behind the scan that the decoder did not need: skip to what looks like a marker (0xff, then neither a
            // stuffed zero nor another 0xff)
            while (!eof()) {
                int x = u8();
                while (x == 0xff) {
                    if (eof()) return;
                    x = u8();
                    if (x != 0x00 && x != 0xff) {
                        pendingMarker = x;
                        return;
                    }
                }
            }
        }
    }
    int pendingMarker = kNoMarker;
    int markerAfterScan()
    {
        const int m = pendingMarker != kNoMarker ? pendingMarker : nextMarker();
        pendingMarker = kNoMarker;
        return m;
    }

    void finishProgressive()
    {
        int16_t tmp[64];
        for (int n = 0; n < nComp; n++) {
            Component& c = comp[n];
            if (!quantDefined[c.tq]) fail("component uses an undefined quantisation table");
            const uint16_t* q = quant[c.tq];
            const int w = (c.width + 7) >> 3, h = (c.height + 7) >> 3;
            for (int by = 0; by < h; by++)
                for (int bx = 0; bx < w; bx++) {
                    const int16_t* blk = c.coeff.data() + 64 * (static_cast<size_t>(by) * c.blocksW + bx);
                    for (int i = 0; i < 64; i++) tmp[i] = static_cast<int16_t>(blk[i] * static_cast<int>(q[i]));
                    idct8x8(c.plane.data() + static_cast<size_t>(by) * 8 * c.planeW + static_cast<size_t>(bx) * 8, c.planeW, tmp);
                }
        }
    }

    Texture run()
    {
        pos = 0;
        if (nextMarker() != 0xd8) fail("no SOI marker");
        int m = nextMarker();
        while (m != 0xc0 && m != 0xc1 && m != 0xc2) {
            tables(m);  // (lossless / hierarchical / arithmetic-coding frames end here as unknown markers)
            m = nextMarker();
            while (m == kNoMarker) {  // padding between segments
                if (eof()) fail("no frame header");
                m = nextMarker();
            }
        }
        frame(m);
        m = nextMarker();
        while (m != 0xd9) {
            if (m == 0xda) {
                scanHeader();
                scan();
                m = markerAfterScan();
                if (m >= 0xd0 && m <= 0xd7) m = nextMarker();
            } else if (m == 0xdc) {  // DNL
                if (u16() != 4) fail("bad DNL length");
                if (u16() != height) fail("bad DNL height");
                m = nextMarker();
            } else {
                // Anything else that cannot be read behind the frame header — a damaged table, a truncated file, trailing
                // bytes — ends the image with what has been decoded (stb_image: stbi__decode_jpeg_image returns success).
                // (stb_image leaves before its dequantise + IDCT step for progressive files: such a file yields the flat image
                //  of its untouched sample planes, and so it does here)
                try {
                    tables(m);
                } catch (const std::runtime_error&) {
                    return output();
                }
                m = nextMarker();
            }
        }
        if (progressive) finishProgressive();
        return output();
    }

    Texture output()
    {
        Texture tex;
        tex.width = static_cast<uint32_t>(width);
        tex.height = static_cast<uint32_t>(height);
        tex.channels = nComp >= 3 ? 3u : 1u;
        tex.pixels.assign(static_cast<size_t>(width) * height * 4, 255);
        const bool isRgb = nComp == 3 && (rgbIds == 3 || (adobeTransform == 0 && !jfif));
        struct Up {
            int hs, vs, ystep, wLores, ypos;
            const uint8_t *line0, *line1;
            std::vector<uint8_t> buf;
        } up[4];
        for (int k = 0; k < nComp; k++) {
            Up& r = up[k];
            r.hs = hMax / comp[k].h;
            r.vs = vMax / comp[k].v;
            r.ystep = r.vs >> 1;
            r.wLores = (width + r.hs - 1) / r.hs;
            r.ypos = 0;
            r.line0 = r.line1 = comp[k].plane.data();
            r.buf.assign(static_cast<size_t>(width) + 3 + 8, 0);
        }
        const uint8_t* rows[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int j = 0; j < height; j++) {
            uint8_t* out = tex.pixels.data() + static_cast<size_t>(j) * width * 4;
            for (int k = 0; k < nComp; k++) {
                Up& r = up[k];
                const bool bottom = r.ystep >= (r.vs >> 1);
                rows[k] = upsample_row(r.buf.data(), bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.wLores, r.hs, r.vs);
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < comp[k].height) r.line1 += comp[k].planeW;
                }
            }
            if (nComp == 1) {
                for (int i = 0; i < width; i++, out += 4) out[0] = out[1] = out[2] = rows[0][i];
            } else if (nComp == 3) {
                if (isRgb)
                    for (int i = 0; i < width; i++, out += 4) { out[0] = rows[0][i]; out[1] = rows[1][i]; out[2] = rows[2][i]; }
                else ycc_to_rgba(out, rows[0], rows[1], rows[2], width);
            } else if (adobeTransform == 0) {  // CMYK
                for (int i = 0; i < width; i++, out += 4) {
                    const uint8_t k = rows[3][i];
                    out[0] = mul8(rows[0][i], k); out[1] = mul8(rows[1][i], k); out[2] = mul8(rows[2][i], k);
                }
            } else if (adobeTransform == 2) {  // YCCK
                ycc_to_rgba(out, rows[0], rows[1], rows[2], width);
                for (int i = 0; i < width; i++, out += 4) {
                    const uint8_t k = rows[3][i];
                    out[0] = mul8(static_cast<uint8_t>(255 - out[0]), k); out[1] = mul8(static_cast<uint8_t>(255 - out[1]), k); out[2] = mul8(static_cast<uint8_t>(255 - out[2]), k);
                }
            } else {
                ycc_to_rgba(out, rows[0], rows[1], rows[2], width);  // four components without an Adobe marker: the fourth is ignored
            }
        }
        return tex;
    }
};

}  // namespace

Texture decode(const unsigned char* data, size_t size)
{
    if (size < 4 || data[0] != 0xff || data[1] != 0xd8) fail("no SOI marker");
    Decoder d(data, size);
    return d.run();
}

}  // namespace jpeg
}  // namespace nexus
