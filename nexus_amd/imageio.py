"""PNG output of the RGBA8 render buffer (the reference's only persisted artefact is a PNG screenshot,
/root/reference/Nexus/src/Renderer/Renderer.cpp:183-215, written there through stb_image_write)."""
import struct
import zlib

import numpy as np


def write_png(path, rgba8, width, height, flip_y=True):
    """rgba8: uint32 array (R in the low byte, as the accumulate kernel packs it) of width*height pixels."""
    img = np.ascontiguousarray(rgba8, dtype=np.uint32).reshape(height, width).view(np.uint8).reshape(height, width, 4)
    if flip_y:  # image row 0 is the bottom of the viewport (the camera's lowerLeftCorner)
        img = img[::-1]
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(height))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def write_pfm(path, rgb, width, height):
    """Float accumulation as a Portable Float Map (little endian, bottom row first — the accumulation's own row order)."""
    img = np.ascontiguousarray(rgb, dtype="<f4").reshape(height, width, 3)
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (width, height))
        f.write(img.tobytes())


def read_pfm(path):
    with open(path, "rb") as f:
        if f.readline().strip() != b"PF":
            raise ValueError("not a colour PFM file")
        width, height = (int(x) for x in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(width * height * 12), dtype="<f4" if scale < 0 else ">f4")
    return data.reshape(height * width, 3).astype(np.float32), width, height


def write_exr(path, rgb, width, height, flip_y=True):
    """Float image as OpenEXR: single-part scanline file, no compression, 32-bit float channels B, G, R (the format's
    alphabetical channel order); rows top to bottom."""
    img = np.ascontiguousarray(rgb, dtype="<f4").reshape(height, width, 3)
    if flip_y:
        img = img[::-1]

    def attr(name, typ, value):
        return name + b"\0" + typ + b"\0" + struct.pack("<i", len(value)) + value

    chlist = b"".join(c + b"\0" + struct.pack("<iIii", 2, 0, 1, 1) for c in (b"B", b"G", b"R")) + b"\0"
    window = struct.pack("<iiii", 0, 0, width - 1, height - 1)
    header = (struct.pack("<II", 20000630, 2) + attr(b"channels", b"chlist", chlist) + attr(b"compression", b"compression", b"\0")
              + attr(b"dataWindow", b"box2i", window) + attr(b"displayWindow", b"box2i", window) + attr(b"lineOrder", b"lineOrder", b"\0")
              + attr(b"pixelAspectRatio", b"float", struct.pack("<f", 1.0)) + attr(b"screenWindowCenter", b"v2f", struct.pack("<ff", 0.0, 0.0))
              + attr(b"screenWindowWidth", b"float", struct.pack("<f", 1.0)) + b"\0")
    line = width * 12
    first = len(header) + 8 * height
    with open(path, "wb") as f:
        f.write(header)
        f.write(b"".join(struct.pack("<Q", first + y * (8 + line)) for y in range(height)))
        for y in range(height):
            f.write(struct.pack("<ii", y, line))
            f.write(np.ascontiguousarray(img[y, :, ::-1].T).tobytes())  # B, G, R planes


def read_exr(path):
    """Reader for the files write_exr / nexus::WriteEXR produce (uncompressed scanline, float32 channels): (rows top to
    bottom [h][w][3] RGB, width, height)."""
    data = open(path, "rb").read()
    magic, version = struct.unpack_from("<II", data, 0)
    if magic != 20000630 or (version & 0xFF) != 2:
        raise ValueError("not an OpenEXR 2 file")
    off, attrs = 8, {}
    while data[off] != 0:
        end = data.index(b"\0", off)
        name = data[off:end]
        off = end + 1
        end = data.index(b"\0", off)
        typ = data[off:end]
        off = end + 1
        (n,) = struct.unpack_from("<i", data, off)
        attrs[name] = (typ, data[off + 4: off + 4 + n])
        off += 4 + n
    off += 1
    if attrs[b"compression"][1] != b"\0":
        raise ValueError("compressed EXR files are not supported")
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs[b"dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names, c = [], attrs[b"channels"][1]
    p = 0
    while c[p] != 0:
        e = c.index(b"\0", p)
        names.append(c[p:e])
        if struct.unpack_from("<i", c, e + 1)[0] != 2:
            raise ValueError("only FLOAT channels are supported")
        p = e + 1 + 16
    offsets = struct.unpack_from("<%dQ" % h, data, off)
    img = np.zeros((h, w, 3), dtype=np.float32)
    for k in range(h):
        y, n = struct.unpack_from("<ii", data, offsets[k])
        planes = np.frombuffer(data, dtype="<f4", count=w * len(names), offset=offsets[k] + 8).reshape(len(names), w)
        for ci, nm in enumerate(names):
            if nm in (b"R", b"G", b"B"):
                img[y - y0, :, {b"R": 0, b"G": 1, b"B": 2}[nm]] = planes[ci]
    return img, w, h
