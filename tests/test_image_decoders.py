"""The product's image decoders (nexus::IMGLoader: PNG incl. Adam7, Radiance .hdr; nexus::jpeg: baseline and progressive
JPEG) against the decoder the reference itself uses: stb_image, stbi_load(..., 4) (/root/reference/Nexus/src/Assets/
IMGLoader.cpp:17-41).  Two pins:
  * committed outputs of stb_image on the committed fixture files (tests/golden/images_golden.npz, written by
    tests/golden/make_image_golden.py with the reference build of oracle/stb_ref.c) — run everywhere;
  * stb_image itself, where the reference build exists (oracle/_ref/libstbref.so: this container, and the GPU box, which
    receives the built library): generated PNG files of every colour type / bit depth / interlacing, HDR files, the JPEG
    fixtures, and damaged copies of all of them (both decoders must refuse a file, or agree on every pixel).
No GPU involved."""
import ctypes as C
import glob
import os
import struct
import zlib

import numpy as np
import pytest

from nexus_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
IMAGES = os.path.join(HERE, "golden", "images")
REF_LIB = os.path.join(HERE, "..", "oracle", "_ref", "libstbref.so")


def _ref():
    if not os.path.exists(REF_LIB):
        pytest.skip("no reference build (make -C oracle ref needs /root/reference)")
    L = C.CDLL(REF_LIB)
    L.nxref_image_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    return L


def stb_decode(L, data):
    w, h, c = C.c_int(0), C.c_int(0), C.c_int(0)
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    if L.nxref_image_size(buf, len(data), C.byref(w), C.byref(h), C.byref(c)) != 0:
        return None
    px = np.zeros((h.value, w.value, 4), np.uint8)
    assert L.nxref_image_decode(buf, len(data), px.ctypes.data_as(C.c_void_p), px.size) == 0
    return px, c.value


def ours(data):
    try:
        return capi.decode_image(data)
    except capi.NexusError:
        return None


def _fixtures():
    return sorted(glob.glob(os.path.join(IMAGES, "*.jpg")))


def test_jpeg_fixtures_decode_to_the_committed_stb_outputs():
    gold = np.load(os.path.join(HERE, "golden", "images_golden.npz"))
    files = _fixtures()
    assert len(files) >= 20
    kinds = set()
    for f in files:
        name = os.path.splitext(os.path.basename(f))[0]
        px, ch = capi.decode_image(open(f, "rb").read())
        want = gold[name]
        assert px.shape == want.shape, name
        assert np.array_equal(px, want), "%s: %d of %d bytes differ from stb_image" % (name, int((px != want).sum()), px.size)
        assert ch == int(gold[name + "__channels"]), name
        kinds.add(name.split("_")[0] + "_" + name.split("_")[1])
    assert {"base_444", "base_420", "base_422", "base_grey", "prog_444", "prog_420", "prog_422", "prog_grey", "base_cmyk"} <= kinds


def test_jpeg_fixtures_against_the_reference_build():
    L = _ref()
    for f in _fixtures():
        data = open(f, "rb").read()
        want = stb_decode(L, data)
        got = ours(data)
        assert want is not None and got is not None, f
        assert np.array_equal(got[0], want[0]) and got[1] == want[1], f


# ---- PNG files of every kind, written here -------------------------------------------------------------------------------

ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def _pack_rows(smp, depth):
    """[h][w][samples] integer samples -> [h][stride] bytes at `depth` bits per sample"""
    h, w, n = smp.shape
    stride = (w * n * depth + 7) // 8
    rows = np.zeros((h, stride), np.uint8)
    for y in range(h):
        flat = smp[y].reshape(-1)
        if depth == 8:
            rows[y] = flat
        elif depth == 16:
            rows[y, 0::2] = flat >> 8
            rows[y, 1::2] = flat & 255
        else:
            b = np.zeros(stride * 8, np.uint8)
            for k in range(depth):
                b[k: len(flat) * depth: depth] = (flat >> (depth - 1 - k)) & 1
            rows[y] = np.packbits(b)
    return rows


def _filter_rows(rows, bpp, filters, start):
    out = bytearray()
    stride = rows.shape[1]
    prev = np.zeros(stride, np.int32)
    for y in range(rows.shape[0]):
        f = filters[(start + y) % len(filters)]
        cur = rows[y].astype(np.int32)
        shift = lambda a: np.concatenate([np.zeros(bpp, np.int32), a[:-bpp]]) if stride > bpp else np.zeros(stride, np.int32)
        left, upleft = shift(cur), shift(prev)
        if f == 0:
            enc = cur
        elif f == 1:
            enc = cur - left
        elif f == 2:
            enc = cur - prev
        elif f == 3:
            enc = cur - (left + prev) // 2
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            enc = cur - pred
        out.append(f)
        out += bytes((enc & 255).astype(np.uint8))
        prev = cur
    return bytes(out)


def make_png(smp, colour, depth, interlace=False, palette=None, trns=None, filters=(0, 1, 2, 3, 4)):
    h, w, n = smp.shape
    bpp = max(1, n * depth // 8)
    raw = b""
    if not interlace:
        raw = _filter_rows(_pack_rows(smp, depth), bpp, filters, 0)
    else:
        for k, (x0, y0, dx, dy) in enumerate(ADAM7):
            sub = smp[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += _filter_rows(_pack_rows(sub, depth), bpp, filters, k)

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)

    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    comp = zlib.compress(raw, 6)
    third = max(1, len(comp) // 3)
    for i in range(0, len(comp), third):  # several IDAT chunks
        out += chunk(b"IDAT", comp[i:i + third])
    return out + chunk(b"IEND", b"")


def png_cases(rng):
    cases = []
    sizes = [(1, 1), (2, 3), (5, 4), (8, 8), (9, 17), (33, 7), (16, 9)]
    kinds = [(0, d, 1) for d in (1, 2, 4, 8, 16)] + [(2, d, 3) for d in (8, 16)] + [(3, d, 1) for d in (1, 2, 4, 8)] + [(4, d, 2) for d in (8, 16)] + [(6, d, 4) for d in (8, 16)]
    for colour, depth, n in kinds:
        for interlace in (False, True):
            for (w, h) in sizes[:: 1 if interlace else 2]:
                palette = trns = None
                maxv = (1 << depth) - 1
                if colour == 3:
                    entries = min(1 << depth, 1 + rng.randint(1, 1 << depth))
                    palette = rng.randint(0, 256, 3 * entries).astype(np.uint8)
                    maxv = entries - 1
                    if rng.rand() < 0.5:
                        trns = rng.randint(0, 256, rng.randint(1, entries + 1)).astype(np.uint8)
                smp = rng.randint(0, maxv + 1, size=(h, w, n)).astype(np.uint32)
                if colour in (0, 2) and rng.rand() < 0.5:  # a colour key that occurs in the image
                    key = smp[rng.randint(h), rng.randint(w)]
                    trns = b"".join(struct.pack(">H", int(v)) for v in key)
                cases.append(("png c%d d%d %s %dx%d" % (colour, depth, "adam7" if interlace else "plain", w, h), make_png(smp, colour, depth, interlace, palette, trns)))
    return cases


def hdr_cases(rng):
    cases = []
    for (w, h, rle) in ((7, 5, False), (40, 9, True), (9, 3, True), (130, 4, True)):
        rgbe = rng.randint(0, 256, size=(h, w, 4)).astype(np.uint8)
        rgbe[..., 3] = rng.randint(110, 140, size=(h, w))
        body = b""
        for y in range(h):
            if not rle or w < 8 or w >= 32768:
                body += rgbe[y].tobytes()
            else:
                body += bytes([2, 2, w >> 8, w & 255])
                for ch in range(4):
                    row = rgbe[y, :, ch]
                    x = 0
                    while x < w:  # literal runs of up to 128, a repeat run wherever three equal bytes start
                        run = 1
                        while x + run < w and run < 127 and row[x + run] == row[x]:
                            run += 1
                        if run >= 3:
                            body += bytes([128 + run, int(row[x])])
                            x += run
                        else:
                            n = min(128, w - x)
                            body += bytes([n]) + row[x:x + n].tobytes()
                            x += n
        cases.append(("hdr %dx%d rle=%s" % (w, h, rle), b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w) + body))
    return cases


def test_png_and_hdr_files_of_every_kind_against_the_reference_build():
    L = _ref()
    rng = np.random.RandomState(5)
    cases = png_cases(rng) + hdr_cases(rng)
    assert len(cases) > 100
    for name, data in cases:
        want = stb_decode(L, data)
        got = ours(data)
        assert want is not None, name + ": the reference decoder refuses the generated file"
        assert got is not None, name + ": refused"
        assert got[0].shape == want[0].shape and np.array_equal(got[0], want[0]), name
        assert got[1] == want[1], name + ": channel count"


def test_damaged_files_are_refused_or_decoded_like_the_reference_build():
    """Random byte edits, truncations and insertions in JPEG and PNG files of every kind.  The product may be stricter than
    stb_image about a broken container, never laxer, and whenever it does decode a damaged file the pixels are stb_image's
    (its portable code paths, zero-filled buffers: oracle/stb_ref.c).  That covers the entropy decoder's behaviour at
    markers and at the end of the data, runs past the end of a block, out-of-range coefficients through the inverse DCT,
    truncated progressive files.  Damaged .hdr files only have to be survived: stb_image re-reads a scanline it cannot
    make sense of as flat pixels from the top of the image, which the product does not imitate."""
    L = _ref()
    rng = np.random.RandomState(9)
    sources = [(os.path.basename(f), open(f, "rb").read()) for f in _fixtures()]
    sources += png_cases(np.random.RandomState(6))[::9] + hdr_cases(np.random.RandomState(7))[:2]
    decoded = refused = stricter = 0
    for name, data in sources:
        for trial in range(30):
            b = bytearray(data)
            kind = trial % 3
            if kind == 0:
                for _ in range(rng.randint(1, 4)):
                    b[rng.randint(len(b))] = rng.randint(256)
            elif kind == 1:
                del b[rng.randint(max(1, len(b) // 2), len(b)):]
            else:
                at = rng.randint(len(b))
                b[at:at] = bytes(rng.randint(0, 256, rng.randint(1, 5)).astype(np.uint8))
            b = bytes(b)
            want = stb_decode(L, b)
            got = ours(b)
            if name.startswith("hdr"):
                continue
            if got is None:
                refused += 1
                stricter += want is not None
                continue
            decoded += 1
            assert want is not None, "%s trial %d: decoded a file the reference decoder refuses" % (name, trial)
            assert got[0].shape == want[0].shape and np.array_equal(got[0], want[0]), "%s trial %d: pixels differ from the reference decoder's" % (name, trial)
    assert decoded > 100 and refused > 100
