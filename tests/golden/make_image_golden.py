"""Writes tests/golden/images/: small JPEG files of every kind the product's decoder supports (Pillow / libjpeg is only the
ENCODER here; it is not needed to run the tests) and, beside them, what the reference's own decoder makes of each —
stb_image through stbi_load_from_memory(..., 4), built from the reference's vendored header by `make -C oracle ref`
(oracle/stb_ref.c) — as images_golden.npz.  Run in the build container (needs /root/reference for the reference build):

    make -C oracle ref && python tests/golden/make_image_golden.py
"""
import ctypes as C
import io
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "images")
REF = os.path.join(HERE, "..", "..", "oracle", "_ref", "libstbref.so")


def stb_decode(data):
    L = C.CDLL(REF)
    w, h, c = C.c_int(0), C.c_int(0), C.c_int(0)
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    if L.nxref_image_size(buf, len(data), C.byref(w), C.byref(h), C.byref(c)) != 0:
        return None
    px = np.zeros((h.value, w.value, 4), np.uint8)
    L.nxref_image_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    assert L.nxref_image_decode(buf, len(data), px.ctypes.data_as(C.c_void_p), px.size) == 0
    return px, c.value


def test_image(w, h, seed, grey=False):
    """smooth gradients + an edge + noise: exercises DC prediction, many AC magnitudes and the chroma filters"""
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([128 + 100 * np.sin(x / 5.0 + seed) * np.cos(y / 7.0), 255 * x / max(1, w - 1), 255 * y / max(1, h - 1)], -1)
    img[h // 3:, w // 2:] = 255 - img[h // 3:, w // 2:]
    img += rng.normal(0, 12, img.shape)
    img = np.clip(img, 0, 255).astype(np.uint8)
    return img[..., 0] if grey else img


def main():
    from PIL import Image

    os.makedirs(OUT, exist_ok=True)
    cases = []

    def add(name, arr, mode=None, **kw):
        im = Image.fromarray(arr, mode) if mode else Image.fromarray(arr)
        b = io.BytesIO()
        im.save(b, "JPEG", **kw)
        cases.append((name, b.getvalue()))

    add("base_444_q90_37x29", test_image(37, 29, 1), quality=90, subsampling=0)
    add("base_420_q75_37x29", test_image(37, 29, 2), quality=75, subsampling=2)
    add("base_422_q60_50x21", test_image(50, 21, 3), quality=60, subsampling=1)
    add("base_420_q30_64x48", test_image(64, 48, 4), quality=30, subsampling=2)
    add("base_420_q95_17x33_opt", test_image(17, 33, 5), quality=95, subsampling=2, optimize=True)
    add("base_grey_q80_41x23", test_image(41, 23, 6, grey=True), quality=80)
    add("base_444_1x1", test_image(1, 1, 7), quality=85, subsampling=0)
    add("base_420_1x1", test_image(1, 1, 8), quality=85, subsampling=2)
    add("base_420_9x2", test_image(9, 2, 9), quality=85, subsampling=2)
    add("base_420_2x9", test_image(2, 9, 10), quality=85, subsampling=2)
    add("base_420_16x16", test_image(16, 16, 11), quality=50, subsampling=2)
    add("base_420_q80_restart_70x40", test_image(70, 40, 12), quality=80, subsampling=2, restart_marker_blocks=3)
    add("base_444_q80_restart_rows_40x40", test_image(40, 40, 13), quality=80, subsampling=0, restart_marker_rows=1)
    add("prog_444_q85_37x29", test_image(37, 29, 14), quality=85, subsampling=0, progressive=True)
    add("prog_420_q70_45x31", test_image(45, 31, 15), quality=70, subsampling=2, progressive=True)
    add("prog_422_q40_64x33", test_image(64, 33, 16), quality=40, subsampling=1, progressive=True)
    add("prog_grey_q75_33x33", test_image(33, 33, 17, grey=True), quality=75, progressive=True)
    add("prog_420_q90_restart_48x48", test_image(48, 48, 18), quality=90, subsampling=2, progressive=True, restart_marker_blocks=2)
    add("prog_420_q20_80x24", test_image(80, 24, 19), quality=20, subsampling=2, progressive=True)
    cmyk = np.concatenate([test_image(24, 18, 20), test_image(24, 18, 21)[..., :1]], -1)
    add("base_cmyk_q80_24x18", cmyk, mode="CMYK", quality=80)
    # RGB stored without a colour transform (component ids R, G, B are not what libjpeg writes; the Adobe marker with
    # transform 0 is): keep_rgb needs a recent Pillow
    try:
        add("base_rgb_notransform_30x20", test_image(30, 20, 22), quality=85, keep_rgb=True)
    except Exception as e:  # pragma: no cover
        print("skipped keep_rgb case:", e)
    # 4:1:1 and 4:4:0 sampling through explicit factors, where Pillow accepts them
    for name, ss in (("base_411_q80_66x19", "4:1:1"), ("base_440_q80_21x40", "4:4:0")):
        try:
            add(name, test_image(*[int(v) for v in name.rsplit("_", 1)[1].split("x")], 23), quality=80, subsampling=ss)
        except Exception as e:  # pragma: no cover
            print("skipped", name, e)

    golden = {}
    for name, data in cases:
        with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
            f.write(data)
        r = stb_decode(data)
        assert r is not None, name
        golden[name] = r[0]
        golden[name + "__channels"] = np.int32(r[1])
        print("%-36s %6d bytes  %dx%d  channels %d" % (name, len(data), r[0].shape[1], r[0].shape[0], r[1]))
    np.savez_compressed(os.path.join(HERE, "images_golden.npz"), **golden)


if __name__ == "__main__":
    main()
