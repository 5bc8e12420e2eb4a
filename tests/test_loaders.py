"""Scene ingestion: the .glb reader on the reference's own Cornell box asset (facts from SURVEY.md Appendix C) and the
.obj reader on a generated file."""
import os

import numpy as np
import pytest

from nexus_amd import loaders, pod
from tests import scene_helpers as SH


def test_cornell_glb_facts():
    ls = loaders.load_glb(os.path.join(SH.GOLDEN, "cornell_box.glb"))
    assert [len(m) for m in ls.meshes] == [2, 2, 2, 2, 2, 10, 10, 2]  # floor ceiling back right left shortBox tallBox light
    assert sum(len(m) for m in ls.meshes) == 32 and len(ls.instances) == 8
    assert ls.material_names == ["floor", "ceiling", "backWall", "rightWall", "leftWall", "shortBox", "tallBox", "light"]
    assert np.allclose(ls.materials["u"][3][:3], (0.14, 0.45, 0.091), atol=1e-6)
    assert np.allclose(ls.materials["u"][4][:3], (0.63, 0.065, 0.05), atol=1e-6)
    light = ls.materials[7]
    assert np.allclose(light["emissive"], 1.0) and light["intensity"] == 35.0
    assert all(abs(m["u"][3] - 0.9) < 1e-4 and m["u"][4] == 1.0 and m["type"] == pod.MAT_PLASTIC for m in ls.materials)
    for inst in ls.instances:  # node rotation quaternion (0.7071, 0, 0, 0.7071) = +90 degrees about X
        assert np.allclose(inst["rotation"], (90, 0, 0), atol=1e-3) and np.allclose(inst["scale"], 1, atol=1e-6) and np.allclose(inst["position"], 0)


def test_cornell_scene_bounds_and_lights():
    sc = SH.cornell_scene(32, 32)
    assert len(sc.lights) == 1 and sc.lights[0]["meshId"] == 7
    # after the rotation the box spans x [-1.02, 1], y [0, 1.99], z [-1.04, 0.99] (open side towards +z)
    pts = []
    for inst, (nodes, tris, idx) in zip(sc.instances, sc.blas):
        m = inst["transform"].reshape(4, 4)
        for k in ("pos0", "pos1", "pos2"):
            p = tris[k] @ m[:3, :3].T + m[:3, 3]
            pts.append(p)
    pts = np.concatenate(pts)
    assert np.allclose(pts.min(0), (-1.02, 0.0, -1.04), atol=2e-3) and np.allclose(pts.max(0), (1.0, 1.99, 0.99), atol=2e-3)


def test_obj_reader(tmp_path):
    p = tmp_path / "quad.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4//1 -3//1 -2//1\n")
    ls = loaders.load_obj(str(p))
    assert len(ls.meshes) == 1 and len(ls.meshes[0]) == 3  # quad fan = 2 triangles + 1
    t = ls.meshes[0]
    assert np.allclose(t["pos0"][0], (0, 0, 0)) and np.allclose(t["pos2"][1], (0, 1, 0))
    assert np.allclose(t["normal0"], (0, 0, 1))


# ---- the C++ reader (nexus::OBJLoader through the C-ABI) against the Python one ------------------------------------

def _same_scene(cpp, py):
    meshes, mats, insts = cpp
    assert len(meshes) == len(py.meshes)
    for a, b in zip(meshes, py.meshes):
        assert a.tobytes() == np.ascontiguousarray(b).tobytes()
    assert mats.tobytes() == np.ascontiguousarray(py.materials).tobytes()
    assert len(insts) == len(py.instances)
    for a, b in zip(insts, py.instances):
        assert a["mesh"] == b["mesh"] and a["material"] == b["material"]
        for k in ("position", "rotation", "scale"):
            assert np.allclose(a[k], b[k], rtol=0, atol=1e-5), (k, a[k], b[k])


def test_cpp_glb_reader_equals_python_reader():
    from nexus_amd import capi

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_box.glb")
    cpp = capi.load_scene_file(path)
    py = loaders.load_glb(path)
    _same_scene(cpp, py)
    assert len(cpp[0]) == 8 and sum(len(m) for m in cpp[0]) == 32


def test_cpp_obj_reader_equals_python_reader(tmp_path):
    from nexus_amd import capi

    rng = np.random.RandomState(3)
    p = tmp_path / "soup.obj"
    lines = []
    for _ in range(40):
        lines.append("v %.6f %.6f %.6f" % tuple(rng.uniform(-1, 1, 3)))
    for _ in range(10):
        lines.append("vn %.6f %.6f %.6f" % tuple(rng.uniform(-1, 1, 3)))
        lines.append("vt %.6f %.6f" % tuple(rng.uniform(0, 1, 2)))
    for _ in range(25):  # triangles, quads and a pentagon; positive and negative indices; all three index forms
        n = rng.choice([3, 3, 4, 5])
        vs = rng.randint(1, 41, n)
        lines.append("f " + " ".join("%d/%d/%d" % (v, rng.randint(1, 11), rng.randint(1, 11)) for v in vs))
    lines.append("f -1/-1/-1 -2/-2/-2 -3/-3/-3")
    p.write_text("\n".join(lines) + "\n")
    _same_scene(capi.load_scene_file(str(p)), loaders.load_obj(str(p)))
    # positions only: face normals are generated
    q = tmp_path / "plain.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1 3 4 2\n")
    _same_scene(capi.load_scene_file(str(q)), loaders.load_obj(str(q)))


def test_cpp_reader_rejects_bad_input(tmp_path):
    from nexus_amd import capi

    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"glTF" + b"\x02\x00\x00\x00" + b"\xff" * 24)
    with pytest.raises(capi.NexusError):
        capi.load_scene_file(str(bad))
    with pytest.raises(capi.NexusError):
        capi.load_scene_file(str(tmp_path / "missing.obj"))
    idx = tmp_path / "idx.obj"
    idx.write_text("v 0 0 0\nv 1 0 0\nf 1 2 9\n")
    with pytest.raises(capi.NexusError):
        capi.load_scene_file(str(idx))
    with pytest.raises(capi.NexusError):
        capi.load_scene_file(str(tmp_path / "scene.fbx"))


# ---- cornell_box_sphere.glb: the second data file the reference ships (assets/demo_scenes/cornell_box_sphere) -------

def test_sphere_glb_facts_and_cpp_reader():
    """2 188 triangles, two transmissive spheres (ior 1.45 and 2.5) that become DIELECTRIC through
    transmissionFactor > 0 (OBJLoader.cpp:97-102)."""
    from nexus_amd import capi

    path = os.path.join(SH.GOLDEN, "cornell_box_sphere.glb")
    ls = loaders.load_glb(path)
    assert sum(len(m) for m in ls.meshes) == 2188
    glass = [m for m in ls.materials if m["type"] == pod.MAT_DIELECTRIC]
    assert len(glass) == 2
    assert sorted(round(float(m["u"][4]), 3) for m in glass) == [1.45, 2.5]           # plastic/dielectric ior slot
    assert all(m["type"] in (pod.MAT_PLASTIC, pod.MAT_DIELECTRIC) for m in ls.materials)
    lights = [m for m in ls.materials if float(m["intensity"]) * float(np.max(m["emissive"])) > 0.0]
    assert len(lights) == 1
    assert ls.textures == [] and ls.warnings == []
    _same_scene(capi.load_scene_file(path), ls)


# ---- images: the PNG decoder (C++ nexus::IMGLoader and its Python twin) ---------------------------------------------

def _png(width, height, colour, depth, samples, rng, palette=None, trns=None, filters=(0, 1, 2, 3, 4)):
    """Encode a random image with the given colour type / bit depth, cycling through the scanline filters; returns
    (file bytes, sample array [h][w][samples] at full bit depth)."""
    import struct
    import zlib

    maxv = (1 << depth) - 1 if colour != 3 else len(palette) // 3 - 1
    smp = rng.randint(0, maxv + 1, size=(height, width, samples)).astype(np.uint32)
    bits = samples * depth
    stride = (width * bits + 7) // 8
    bpp = max(1, bits // 8)
    rows = np.zeros((height, stride), dtype=np.uint8)
    for y in range(height):
        flat = smp[y].reshape(-1)
        if depth == 8:
            rows[y] = flat
        elif depth == 16:
            rows[y, 0::2] = flat >> 8
            rows[y, 1::2] = flat & 255
        else:
            b = np.zeros(stride * 8, dtype=np.uint8)
            for k in range(depth):
                b[k: len(flat) * depth: depth] = (flat >> (depth - 1 - k)) & 1
            rows[y] = np.packbits(b)
    raw = bytearray()
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(height):
        f = filters[y % len(filters)]
        cur = rows[y].astype(np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if stride > bpp else np.zeros(stride, np.int32)
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if stride > bpp else np.zeros(stride, np.int32)
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
        raw.append(f)
        raw += bytes((enc & 255).astype(np.uint8))
        prev = cur

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)

    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, colour, 0, 0, 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    comp = zlib.compress(bytes(raw), 6)
    out += chunk(b"IDAT", comp[: len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b"")
    return out, smp


def _to8(v, depth):
    return (v >> 8).astype(np.uint8) if depth == 16 else v.astype(np.uint8) if depth == 8 else (v * (255 // ((1 << depth) - 1))).astype(np.uint8)


def test_png_decoders_agree_with_the_encoded_samples():
    from nexus_amd import capi

    rng = np.random.RandomState(11)
    cases = [(0, d, 1) for d in (1, 2, 4, 8, 16)] + [(2, 8, 3), (2, 16, 3), (4, 8, 2), (4, 16, 2), (6, 8, 4), (6, 16, 4)] + [(3, d, 1) for d in (1, 2, 4, 8)]
    for colour, depth, samples in cases:
        for (w, h) in ((1, 1), (5, 7), (33, 9)):
            palette = trns = None
            if colour == 3:
                n = 1 << depth
                palette = rng.randint(0, 256, size=3 * n).astype(np.uint8)
                trns = rng.randint(0, 256, size=n // 2 + 1).astype(np.uint8)
            data, smp = _png(w, h, colour, depth, samples, rng, palette, trns)
            want = np.zeros((h, w, 4), dtype=np.uint8)
            want[..., 3] = 255
            if colour == 0:
                want[..., 0] = want[..., 1] = want[..., 2] = _to8(smp[..., 0], depth)
            elif colour == 2:
                want[..., :3] = _to8(smp, depth)
            elif colour == 3:
                pal = palette.reshape(-1, 3)
                want[..., :3] = pal[smp[..., 0]]
                alpha = np.full(256, 255, np.uint8)
                alpha[: len(trns)] = trns
                want[..., 3] = alpha[smp[..., 0]]
            elif colour == 4:
                want[..., 0] = want[..., 1] = want[..., 2] = _to8(smp[..., 0], depth)
                want[..., 3] = _to8(smp[..., 1], depth)
            else:
                want[...] = _to8(smp, depth)
            py, _ = loaders.decode_png(data)
            cpp, _ = capi.decode_png(data)
            assert np.array_equal(py, want), (colour, depth, w, h)
            assert np.array_equal(cpp, want), (colour, depth, w, h)
    # colour-key transparency (tRNS on grey / RGB) and what must be refused
    import struct
    data, smp2 = _png(6, 4, 2, 8, 3, np.random.RandomState(5), trns=struct.pack(">HHH", 7, 7, 7))
    for dec in (loaders.decode_png, capi.decode_png):
        img, ch = dec(data)
        assert ch == 4 and np.array_equal(img[..., 3] == 0, np.all(smp2 == 7, axis=-1))
    with pytest.raises(Exception):
        capi.decode_png(b"\xff\xd8\xff\xe0 not a png")
    with pytest.raises(Exception):
        capi.decode_png(data[:40])
    bad = bytearray(data)
    bad[8 + 8 + 12] = 1  # interlace flag of IHDR
    with pytest.raises(Exception):
        capi.decode_png(bytes(bad))


def _textured_glb(path, rng, images=None):
    """A two-triangle quad with UVs, a base-colour texture and an emissive texture (PNG unless `images` gives the two files'
    bytes, embedded through bufferViews)."""
    import json
    import struct

    if images:
        base_png, emis_png = images
    else:
        base_png, _ = _png(16, 8, 6, 8, 4, rng)
        emis_png, _ = _png(4, 4, 2, 8, 3, rng)
    mime = ["image/jpeg" if b[:2] == b"\xff\xd8" else "image/png" for b in (base_png, emis_png)]
    pos = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    nrm = np.tile(np.array([0, 1, 0], np.float32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    parts, views = [], []
    for b in (pos.tobytes(), nrm.tobytes(), uv.tobytes(), idx.tobytes(), base_png, emis_png):
        off = sum(len(p) for p in parts)
        parts.append(b + b"\0" * (-len(b) % 4))
        views.append({"buffer": 0, "byteOffset": off, "byteLength": len(b)})
    blob = b"".join(parts)
    doc = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0, "name": "quad"}],
        "meshes": [{"name": "quad", "primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0}]}],
        "materials": [{"name": "tex", "pbrMetallicRoughness": {"baseColorFactor": [1, 1, 1, 1], "baseColorTexture": {"index": 0}},
                       "emissiveTexture": {"index": 1}, "emissiveFactor": [1, 1, 1]}],
        "textures": [{"source": 0}, {"source": 1}], "images": [{"bufferView": 4, "mimeType": mime[0]}, {"bufferView": 5, "mimeType": mime[1]}],
        "buffers": [{"byteLength": len(blob)}], "bufferViews": views,
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3", "min": [-1, 0, -1], "max": [1, 0, 1]},
                      {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 2, "componentType": 5126, "count": 4, "type": "VEC2"},
                      {"bufferView": 3, "componentType": 5123, "count": 6, "type": "SCALAR"}],
    }
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(blob), 0x004E4942) + blob
    open(path, "wb").write(struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body)
    return base_png, emis_png


def test_glb_textures_cpp_equals_python_and_reach_the_asset_manager(tmp_path):
    from nexus_amd import capi

    path = str(tmp_path / "textured.glb")
    base_png, emis_png = _textured_glb(path, np.random.RandomState(21))
    py = loaders.load_glb(path)
    assert [k for k, _ in py.textures] == ["diffuse", "emissive"] and py.material_diffuse_texture == [0] and py.material_emissive_texture == [1]
    assert np.array_equal(py.textures[0][1], loaders.decode_png(base_png)[0]) and py.textures[0][1].shape == (8, 16, 4)
    texs, dt, et, warns = capi.load_scene_textures(path)
    assert warns == [] and list(dt) == [0] and list(et) == [1]
    for (ka, pa), (kb, pb) in zip(texs, py.textures):
        assert ka == kb and np.array_equal(pa, pb)
    _same_scene(capi.load_scene_file(path), py)
    # Scene::CreateMeshInstanceFromFile: the maps get ids in the manager's diffuse / emissive lists and the material carries them;
    # an emissive map makes the instance a light (Scene.cpp:166-175)
    sc = capi.Scene(32, 32)
    sc.load_file(str(tmp_path) + os.sep, "textured.glb")
    sc.update()
    assert sc.light_count() == 1
    # an image that cannot be decoded (here: a JPEG signature in front of PNG bytes) is dropped with a warning, the scene still loads
    raw = bytearray(open(path, "rb").read())
    at = raw.find(base_png[:8])
    raw[at: at + 4] = b"\xff\xd8\xff\xe0"
    broken = str(tmp_path / "jpeg.glb")
    open(broken, "wb").write(bytes(raw))
    texs, dt, et, warns = capi.load_scene_textures(broken)
    assert len(texs) == 1 and list(dt) == [-1] and list(et) == [0] and len(warns) == 1 and "JPEG" in warns[0]
    assert loaders.load_glb(broken).material_diffuse_texture == [-1]


def test_glb_with_jpeg_images_loads_like_through_stb_image(tmp_path):
    """glTF's most common image encoding: a baseline 4:2:0 and a progressive JPEG as base-colour and emissive textures.  Both
    readers decode them, to the pixels stb_image produces (tests/golden/images_golden.npz: the reference's decoder on the same
    files), and the scene built from the file has its light."""
    from nexus_amd import capi

    img_dir = os.path.join(SH.GOLDEN, "images")
    names = ("base_420_q75_37x29", "prog_444_q85_37x29")
    files = [open(os.path.join(img_dir, n + ".jpg"), "rb").read() for n in names]
    gold = np.load(os.path.join(SH.GOLDEN, "images_golden.npz"))
    path = str(tmp_path / "jpeg_textures.glb")
    _textured_glb(path, None, images=files)
    texs, dt, et, warns = capi.load_scene_textures(path)
    assert warns == [] and list(dt) == [0] and list(et) == [1]
    py = loaders.load_glb(path)
    assert py.warnings == []
    for (kind, px), (kind_py, px_py), n in zip(texs, py.textures, names):
        assert kind == kind_py and np.array_equal(px, px_py)
        assert np.array_equal(px, gold[n])
    sc = capi.Scene(32, 32)
    sc.load_file(str(tmp_path) + os.sep, "jpeg_textures.glb")
    sc.update()
    assert sc.light_count() == 1


def test_interlaced_png_python_twin_equals_cpp():
    from nexus_amd import capi
    from tests.test_image_decoders import make_png

    rng = np.random.RandomState(31)
    for colour, depth, n, (w, h) in ((0, 1, 1, (9, 5)), (2, 8, 3, (13, 11)), (3, 4, 1, (7, 9)), (4, 16, 2, (3, 2)), (6, 8, 4, (17, 8)), (2, 16, 3, (1, 1))):
        palette = rng.randint(0, 256, 48).astype(np.uint8) if colour == 3 else None
        maxv = 15 if colour == 3 else (1 << depth) - 1
        smp = rng.randint(0, maxv + 1, size=(h, w, n)).astype(np.uint32)
        data = make_png(smp, colour, depth, interlace=True, palette=palette)
        a, ca = capi.decode_image(data)
        b, cb = loaders.decode_png(data)
        assert ca == cb and np.array_equal(a, b)
        plain, _ = capi.decode_image(make_png(smp, colour, depth, interlace=False, palette=palette))
        assert np.array_equal(a, plain)


def test_cpp_decoders_survive_mutated_files(tmp_path):
    """Byte-level mutations of valid files: the C++ PNG decoder and .glb reader must either decode or raise NexusError,
    never read out of bounds (this test is also part of the AddressSanitizer run, tools/run_sanitized_cpu_tests.sh)."""
    import struct
    from nexus_amd import capi
    rng = np.random.RandomState(11)
    pngs = [_png(9, 7, 6, 8, 4, rng)[0], _png(5, 6, 0, 1, 1, rng)[0], _png(6, 4, 2, 16, 3, rng)[0],
            _png(7, 5, 3, 4, 1, rng, palette=bytes(rng.randint(0, 256, size=48).astype(np.uint8)))[0]]
    # JPEG files (baseline with restart markers, progressive) and an interlaced PNG go through the same mill
    from tests.test_image_decoders import make_png
    for n in ("base_420_q80_restart_70x40", "prog_420_q70_45x31", "base_cmyk_q80_24x18"):
        pngs.append(open(os.path.join(SH.GOLDEN, "images", n + ".jpg"), "rb").read())
    pngs.append(make_png(rng.randint(0, 256, size=(9, 11, 3)).astype(np.uint32), 2, 8, interlace=True))
    outcomes = {"ok": 0, "refused": 0}
    for data in pngs:
        for _ in range(60):
            bad = bytearray(data)
            for _ in range(rng.randint(1, 4)):
                k = rng.randint(8, len(bad))
                bad[k] = rng.randint(0, 256)
            if rng.randint(0, 4) == 0:
                bad = bad[: rng.randint(8, len(bad))]
            try:
                img, ch = capi.decode_png(bytes(bad))
                assert img.ndim == 3 and ch in (1, 2, 3, 4)
                outcomes["ok"] += 1
            except capi.NexusError:
                outcomes["refused"] += 1
    glb = open(os.path.join(SH.GOLDEN, "cornell_box_sphere.glb"), "rb").read()
    json_len = struct.unpack_from("<I", glb, 12)[0]
    for i in range(40):
        bad = bytearray(glb)
        # mostly inside the JSON chunk (accessor counts, offsets, indices), sometimes in the binary chunk / the headers
        lo, hi = (20, 20 + json_len) if i % 4 else (0, len(bad))
        for _ in range(rng.randint(1, 3)):
            k = rng.randint(lo, hi)
            bad[k] = rng.randint(32, 127) if i % 4 else rng.randint(0, 256)
        path = tmp_path / ("m%d.glb" % i)
        path.write_bytes(bytes(bad))
        try:
            capi.load_scene_file(str(path))
            outcomes["ok"] += 1
        except capi.NexusError:
            outcomes["refused"] += 1
    assert outcomes["refused"] > 20 and outcomes["ok"] > 20 and outcomes["ok"] + outcomes["refused"] == len(pngs) * 60 + 40


def test_host_bvh_builder_survives_triangles_that_are_not_numbers():
    """A damaged mesh (NaN / infinite vertices) through the host builder: it returns, and the tree it returns still holds every
    triangle exactly once (the reference indexes its bins with a converted NaN and drops children whose slot costs are not
    finite; `tests/test_gpu_lbvh.py` has the device builders' version of this test)."""
    from nexus_amd import capi, pod, scenegen

    tris = np.ascontiguousarray(scenegen.random_soup(3000, seed=12, extent=1.0, size=0.05), dtype=pod.TRI_DT)
    rng = np.random.RandomState(3)
    victims = rng.choice(len(tris), 40, replace=False)
    for kind in ("nan", "inf", "mixed"):
        bad = tris.copy()
        if kind in ("nan", "mixed"):
            bad["pos0"][victims[:15], 0] = np.nan
        if kind in ("inf", "mixed"):
            bad["pos1"][victims[15:30]] = np.inf
            bad["pos2"][victims[30:], 2] = -np.inf
        nodes, idx = capi.bvh8_build(bad, threads=2)
        seen, stack = [], [0]
        while stack:
            n = nodes[stack.pop()]
            inner = 0
            for s in range(8):
                m = int(n["meta"][s])
                if (int(n["imask"]) >> s) & 1:
                    stack.append(int(n["childBaseIdx"]) + inner)
                    inner += 1
                elif m:
                    count = bin(m >> 5).count("1")
                    assert 1 <= count <= 3
                    first = int(n["triangleBaseIdx"]) + (m & 31)
                    seen += idx[first:first + count].tolist()
        assert sorted(seen) == list(range(len(bad))), kind


def test_host_tlas_builder_survives_instance_bounds_that_are_not_numbers():
    """Instances whose world bounds hold NaNs or infinities (a degenerate transform): the agglomerative clustering still ends —
    its nearest-neighbour chain needs a symmetric finite measure, so such boxes are made conservative finite ones first — and
    every instance is in the tree once.  (The reference indexes its table with -1 in this case; this builder used to spin.)"""
    from nexus_amd import capi, pod

    rng = np.random.RandomState(1)
    n = 50
    ident = np.eye(4, dtype=np.float32).reshape(16)
    inst = np.zeros(n, dtype=pod.INST_DT)
    for i in range(n):
        inst[i]["transform"] = ident
        inst[i]["invTransform"] = ident
        lo = rng.uniform(-5, 5, 3).astype(np.float32)
        inst[i]["boundsMin"], inst[i]["boundsMax"] = lo, lo + 1
    clean_nodes, _ = capi.tlas_build(inst)
    for kind in ("nan", "inf", "allnan"):
        bad = inst.copy()
        if kind == "nan":
            bad["boundsMin"][3, 0] = np.nan
            bad["boundsMax"][7] = np.nan
        if kind == "inf":
            bad["boundsMax"][3] = np.inf
            bad["boundsMin"][9, 1] = -np.inf
        if kind == "allnan":
            bad["boundsMin"][:] = np.nan
            bad["boundsMax"][:] = np.nan
        nodes, idx = capi.tlas_build(bad)
        assert sorted(idx.tolist()) == list(range(n)), kind
        assert len(nodes) >= 1
    again, _ = capi.tlas_build(inst)
    assert again.tobytes() == clean_nodes.tobytes()


def test_cpp_obj_reader_survives_mutated_files(tmp_path):
    """Text-level mutations of a valid .obj + .mtl (bytes flipped, tokens inserted — huge and negative indices, nan, inf, NULs —
    ranges deleted, files cut short): the C++ reader either loads the scene or raises NexusError (also part of the
    AddressSanitizer run; 1 600 such files were run under it when the test was written)."""
    from nexus_amd import capi, scenegen

    tris = scenegen.displaced_torus(12, 6, seed=2)
    loaders.write_obj(str(tmp_path / "base.obj"), tris)
    base = b"mtllib base.mtl\nusemtl m0\n" + (tmp_path / "base.obj").read_bytes() + b"\nusemtl m1\nf 1/1/1 2/2/2 3/3/3\nf -1 -2 -3\nf 1//1 2//2 3//3 4//4 5//5\n"
    (tmp_path / "base.mtl").write_bytes(b"newmtl m0\nKd 0.8 0.2 0.1\nKe 1 1 1\nNs 50\nd 0.5\nmap_Kd missing.png\nnewmtl m1\nKd 0.1 0.1 0.9\n")
    tokens = [b"f", b"v", b"vn", b"vt", b"usemtl", b"mtllib", b"-", b"/", b"//", b"999999999999", b"-999999", b"0", b"nan", b"inf", b"1e40", b"\n", b" ", b"\x00", b"\xff"]
    rng = np.random.RandomState(5)
    outcomes = {"ok": 0, "refused": 0}
    for _ in range(80):
        bad = bytearray(base)
        for _ in range(rng.randint(1, 6)):
            k = rng.randint(0, len(bad))
            how = rng.randint(0, 4)
            if how == 0:
                bad[k] = rng.randint(0, 256)
            elif how == 1:
                bad[k:k] = tokens[rng.randint(0, len(tokens))]
            elif how == 2:
                del bad[k:k + rng.randint(1, 20)]
            else:
                bad[k:k + 1] = tokens[rng.randint(0, len(tokens))]
        if rng.randint(0, 5) == 0:
            bad = bad[:rng.randint(1, len(bad))]
        (tmp_path / "m.obj").write_bytes(bytes(bad))
        try:
            capi.load_scene_file(str(tmp_path / "m.obj"))
            outcomes["ok"] += 1
        except capi.NexusError:
            outcomes["refused"] += 1
    assert outcomes["ok"] > 5 and outcomes["refused"] > 5
