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
