#!/usr/bin/env python3
"""Compare two RGBA8 PNGs written by nexus_amd.imageio (same size): prints whether they are identical."""
import struct
import sys
import zlib

import numpy as np


def load(p):
    d = open(p, "rb").read()
    w, h = struct.unpack(">II", d[16:24])
    i, dat = 8, b""
    while i < len(d):
        (n,) = struct.unpack(">I", d[i:i + 4])
        if d[i + 4:i + 8] == b"IDAT":
            dat += d[i + 8:i + 8 + n]
        i += 12 + n
    return np.frombuffer(zlib.decompress(dat), np.uint8).reshape(h, 1 + w * 4)[:, 1:]


a, b = load(sys.argv[1]), load(sys.argv[2])
print("identical:", np.array_equal(a, b), "max diff:", int(np.abs(a.astype(int) - b.astype(int)).max()), "differing bytes:", int((a != b).sum()))
