"""Scene ingestion without Assimp: a minimal binary glTF (.glb) and Wavefront .obj reader.

The reference imports everything through Assimp (/root/reference/Nexus/src/Assets/OBJLoader.cpp:8-239), which is
not available here (empty submodule).  This reader reproduces what that path yields for the files the reference
ships: one mesh + BVH per glTF *primitive* (Assimp splits primitives into aiMeshes, OBJLoader.cpp:165-181), one
instance per (node, primitive) placed with the node's TRS decomposed to Euler degrees (OBJLoader.cpp:183-211), and
the material heuristics of OBJLoader.cpp:71-163 (PLASTIC by default, DIELECTRIC if transmission > 0, emissive =
emissiveFactor with intensity = KHR_materials_emissive_strength, ior from KHR_materials_ior, roughness from the
glTF roughnessFactor — Assimp maps it to shininess (1-r)^2*1000 and the reference maps that back with
1 - sqrt(shininess)/31.62278).
"""
import json
import os
import struct
import zlib

import numpy as np

from . import pod

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _quat_to_matrix(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def _node_local_matrix(node):
    if "matrix" in node:
        return np.array(node["matrix"], dtype=np.float64).reshape(4, 4).T  # glTF stores column major
    m = np.eye(4)
    r = _quat_to_matrix(node.get("rotation", [0, 0, 0, 1]))
    s = np.array(node.get("scale", [1, 1, 1]), dtype=np.float64)
    m[:3, :3] = r * s[None, :]
    m[:3, 3] = node.get("translation", [0, 0, 0])
    return m


def decompose_trs(m):
    """aiMatrix4x4::Decompose into (position, euler XYZ in degrees, scale), as the reference consumes it."""
    pos = m[:3, 3].copy()
    cols = [m[:3, 0].copy(), m[:3, 1].copy(), m[:3, 2].copy()]
    scale = np.array([np.linalg.norm(c) for c in cols])
    if np.linalg.det(m[:3, :3]) < 0:
        scale = -scale
    cols = [c / s if s != 0 else c for c, s in zip(cols, scale)]
    eps = 1e-10
    ry = np.arcsin(-cols[0][2])
    c = np.cos(ry)
    if abs(c) > eps:
        rx = np.arctan2(cols[1][2], cols[2][2])
        rz = np.arctan2(cols[0][1], cols[0][0])
    else:
        rx = 0.0
        rz = np.arctan2(-cols[1][0], cols[1][1])
    return pos, np.degrees([rx, ry, rz]), scale


class LoadedScene:
    """meshes: list of TRI_DT arrays; materials: MAT_DT array; instances: list of dict(mesh, material, position,
    rotation (degrees), scale, name)."""

    def __init__(self):
        self.meshes = []
        self.mesh_names = []
        self.materials = np.zeros(0, dtype=pod.MAT_DT)
        self.material_names = []
        self.instances = []
        self.textures = []            # list of (kind "diffuse" / "emissive", HxWx4 uint8 RGBA, row 0 = top)
        self.material_diffuse_texture = []   # per material: index into textures, -1 = none
        self.material_emissive_texture = []
        self.warnings = []


def decode_hdr(data):
    """Radiance .hdr (RGBE, flat or run-length encoded) -> HxWx4 uint8 the way stbi_load reduces an HDR file to 8 bits
    (stb_image stbi__hdr_to_ldr: gamma 2.2, scale 1, alpha 255) — what Scene::AddHDRMap gets from IMGLoader (Scene.cpp:93-97)."""
    data = bytes(data)
    pos = 0

    def line():
        nonlocal pos
        end = data.index(b"\n", pos)
        l = data[pos:end]
        pos = end + 1
        return l

    if line() not in (b"#?RADIANCE", b"#?RGBE"):
        raise ValueError("not a Radiance HDR file")
    fmt = False
    while True:
        l = line()
        if not l:
            break
        fmt = fmt or l == b"FORMAT=32-bit_rle_rgbe"
    if not fmt:
        raise ValueError("unsupported HDR format")
    parts = line().split()
    if len(parts) != 4 or parts[0] != b"-Y" or parts[2] != b"+X":
        raise ValueError("unsupported HDR orientation")
    h, w = int(parts[1]), int(parts[3])
    rgbe = np.zeros((h, w, 4), dtype=np.uint8)
    for y in range(h):
        if 8 <= w < 32768 and data[pos] == 2 and data[pos + 1] == 2 and not (data[pos + 2] & 0x80) and ((data[pos + 2] << 8) | data[pos + 3]) == w:
            pos += 4
            for c in range(4):
                x = 0
                while x < w:
                    count = data[pos]
                    pos += 1
                    if count > 128:
                        count -= 128
                        rgbe[y, x: x + count, c] = data[pos]
                        pos += 1
                    else:
                        rgbe[y, x: x + count, c] = np.frombuffer(data, dtype=np.uint8, count=count, offset=pos)
                        pos += count
                    x += count
        else:
            rgbe[y] = np.frombuffer(data, dtype=np.uint8, count=4 * w, offset=pos).reshape(w, 4)
            pos += 4 * w
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e != 0, np.ldexp(np.float32(1.0), e - 136), np.float32(0.0)).astype(np.float32)
    f = rgbe[..., :3].astype(np.float32) * scale[..., None]
    z = np.power(f.astype(np.float64), np.float64(np.float32(1.0) / np.float32(2.2))).astype(np.float32) * np.float32(255.0) + np.float32(0.5)
    out = np.full((h, w, 4), 255, dtype=np.uint8)
    out[..., :3] = np.clip(z, 0.0, 255.0).astype(np.int32).astype(np.uint8)
    return out, 3


ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))  # x0, y0, dx, dy per pass


def decode_image(data):
    """An image file in memory -> (HxWx4 uint8, channels), as the reference's IMGLoader gets it from stbi_load(..., 4).
    PNG and Radiance .hdr are decoded here (Python twins of the C++ decoders, compared with them byte for byte by the
    tests); JPEG goes through the product's C++ decoder (nexus::jpeg, checked against stb_image by
    tests/test_image_decoders.py) — there is no Python twin of that one."""
    data = bytes(data)
    if data[:2] == b"\xff\xd8":
        from . import capi

        return capi.decode_image(data)
    if data[:2] == b"#?":
        return decode_hdr(data), 3
    return decode_png(data)


def decode_png(data):
    """PNG (ISO/IEC 15948) -> (HxWx4 uint8, channels of the file), what stbi_load(..., 4) returns and the reference's
    IMGLoader hands to Texture (Assets/IMGLoader.cpp:17-41): all colour types, 1-16 bits, palette / colour-key
    transparency, 16-bit samples reduced to their high byte, Adam7 interlacing."""
    data = bytes(data)
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    off, idat, palette, trns, hdr, ended = 8, b"", b"", b"", None, False
    while off + 12 <= len(data):
        (n,), typ = struct.unpack_from(">I", data, off), data[off + 4: off + 8]
        if n > len(data) - off - 12:
            raise ValueError("chunk runs past the end of the file")
        body = data[off + 8: off + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            palette = body
        elif typ == b"tRNS":
            trns = body
        elif typ == b"IDAT":
            idat += body
        elif typ == b"IEND":
            ended = True
            break
        elif not typ[0] & 0x20:
            raise ValueError("unknown critical chunk")
        off += 12 + n
    if hdr is None or not idat:
        raise ValueError("no IHDR / IDAT chunk")
    if not ended:
        raise ValueError("the file ends before its IEND chunk")
    w, h, depth, colour, _comp, _filt, interlace = hdr
    if interlace > 1:
        raise ValueError("unknown interlace method")
    samples = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[colour]
    bits = samples * depth
    bpp = max(1, bits // 8)
    raw = zlib.decompress(idat)
    passes = [(0, 0, 1, 1, w, h)]
    if interlace:
        passes = [(x0, y0, dx, dy, (w + dx - 1 - x0) // dx, (h + dy - 1 - y0) // dy) for (x0, y0, dx, dy) in ADAM7 if w > x0 and h > y0]
    if len(raw) != sum(((pw * bits + 7) // 8 + 1) * ph for (_x0, _y0, _dx, _dy, pw, ph) in passes):
        raise ValueError("corrupt image data")

    def sub_image(offset, pw, ph):
        """one filtered sub-image of the stream -> [ph][pw][samples] at full bit depth"""
        stride = (pw * bits + 7) // 8
        rows = np.zeros((ph, stride), dtype=np.uint8)
        prev = np.zeros(stride, dtype=np.int32)
        for y in range(ph):
            f = raw[offset + (stride + 1) * y]
            cur = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=offset + (stride + 1) * y + 1).astype(np.int32)
            if f == 0:
                pass
            elif f == 2:
                cur = (cur + prev) & 255
            else:  # 1, 3, 4 depend on the already reconstructed byte bpp to the left: a scan
                out = cur.copy()
                for i in range(stride):
                    a = out[i - bpp] if i >= bpp else 0
                    b = prev[i]
                    c = prev[i - bpp] if i >= bpp else 0
                    if f == 1:
                        out[i] = (out[i] + a) & 255
                    elif f == 3:
                        out[i] = (out[i] + (a + b) // 2) & 255
                    elif f == 4:
                        pp = a + b - c
                        pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                        out[i] = (out[i] + (a if (pa <= pb and pa <= pc) else (b if pb <= pc else c))) & 255
                    else:
                        raise ValueError("unknown scanline filter")
                cur = out
            rows[y] = cur
            prev = cur
        if depth == 8:
            sm = rows[:, : pw * samples].astype(np.uint32)
        elif depth == 16:
            sm = (rows[:, : 2 * pw * samples: 2].astype(np.uint32) << 8) | rows[:, 1: 2 * pw * samples: 2]
        else:
            bitsarr = np.unpackbits(rows, axis=1)[:, : pw * samples * depth].reshape(ph, pw * samples, depth)
            sm = np.zeros((ph, pw * samples), dtype=np.uint32)
            for k in range(depth):
                sm = (sm << 1) | bitsarr[:, :, k]
        return sm.reshape(ph, pw, samples), (stride + 1) * ph

    smp = np.zeros((h, w, samples), dtype=np.uint32)
    offset = 0
    for (x0, y0, dx, dy, pw, ph) in passes:
        if pw == 0 or ph == 0:
            continue
        sub, used = sub_image(offset, pw, ph)
        smp[y0::dy, x0::dx] = sub
        offset += used

    def to8(v):
        if depth == 16:
            return (v >> 8).astype(np.uint8)
        if depth == 8:
            return v.astype(np.uint8)
        return (v * (255 // ((1 << depth) - 1))).astype(np.uint8)

    out = np.zeros((h, w, 4), dtype=np.uint8)
    out[..., 3] = 255
    channels = samples
    if colour == 0:
        out[..., 0] = out[..., 1] = out[..., 2] = to8(smp[..., 0])
        if len(trns) >= 2:
            key = struct.unpack(">H", trns[:2])[0]
            out[..., 3] = np.where(smp[..., 0] == key, 0, 255)
            channels = 2
    elif colour == 2:
        out[..., :3] = to8(smp)
        if len(trns) >= 6:
            key = struct.unpack(">HHH", trns[:6])
            out[..., 3] = np.where((smp[..., 0] == key[0]) & (smp[..., 1] == key[1]) & (smp[..., 2] == key[2]), 0, 255)
            channels = 4
    elif colour == 3:
        pal = np.frombuffer(palette, dtype=np.uint8).reshape(-1, 3)
        idx = smp[..., 0]
        if idx.max() >= len(pal):
            raise ValueError("palette index out of range")
        out[..., :3] = pal[idx]
        alpha = np.full(256, 255, dtype=np.uint8)
        alpha[: len(trns)] = np.frombuffer(trns, dtype=np.uint8)
        out[..., 3] = alpha[idx]
        channels = 4 if trns else 3
    elif colour == 4:
        out[..., 0] = out[..., 1] = out[..., 2] = to8(smp[..., 0])
        out[..., 3] = to8(smp[..., 1])
    else:
        out[...] = to8(smp)
    return out, channels


def _gltf_material(m):
    pbr = m.get("pbrMetallicRoughness", {})
    ext = m.get("extensions", {})
    base = pbr.get("baseColorFactor", [1, 1, 1, 1])
    rough = float(pbr.get("roughnessFactor", 1.0))
    shininess = (1.0 - rough) ** 2 * 1000.0
    roughness = float(np.clip(1.0 - np.sqrt(shininess) / 31.62278, 0.0, 1.0))
    ior = float(ext.get("KHR_materials_ior", {}).get("ior", 1.45))
    transmission = float(ext.get("KHR_materials_transmission", {}).get("transmissionFactor", 0.0))
    mtype = pod.MAT_DIELECTRIC if transmission > 0.0 else pod.MAT_PLASTIC
    emissive = m.get("emissiveFactor", [0, 0, 0])
    intensity = float(ext.get("KHR_materials_emissive_strength", {}).get("emissiveStrength", 1.0))
    return pod.make_material(type=mtype, albedo=base[:3], roughness=roughness, ior=ior, emissive=emissive, intensity=intensity,
                             opacity=float(base[3]) if len(base) > 3 else 1.0)


def load_glb(path):
    data = open(path, "rb").read()
    magic, version, _length = struct.unpack_from("<III", data, 0)
    if magic != 0x46546C67 or version != 2:
        raise ValueError("not a glTF 2.0 binary file")
    off = 12
    doc, blob = None, b""
    while off < len(data):
        clen, ctype = struct.unpack_from("<II", data, off)
        chunk = data[off + 8: off + 8 + clen]
        if ctype == 0x4E4F534A:
            doc = json.loads(chunk)
        elif ctype == 0x004E4942:
            blob = chunk
        off += 8 + clen
    if doc is None:
        raise ValueError("glb without a JSON chunk")

    def accessor(i):
        a = doc["accessors"][i]
        bv = doc["bufferViews"][a["bufferView"]]
        dt = np.dtype(_COMPONENT[a["componentType"]])
        nc = _NCOMP[a["type"]]
        start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or dt.itemsize * nc
        count = a["count"]
        if stride == dt.itemsize * nc:
            arr = np.frombuffer(blob, dtype=dt, count=count * nc, offset=start).reshape(count, nc)
        else:
            arr = np.stack([np.frombuffer(blob, dtype=dt, count=nc, offset=start + k * stride) for k in range(count)])
        return arr

    out = LoadedScene()
    decoded = {}

    def load_texture(ref, kind):
        key = (ref["index"], kind)
        if key in decoded:
            return decoded[key]
        result = -1
        try:
            img = doc["images"][doc["textures"][ref["index"]]["source"]]
            if "bufferView" in img:
                bv = doc["bufferViews"][img["bufferView"]]
                raw = blob[bv.get("byteOffset", 0): bv.get("byteOffset", 0) + bv["byteLength"]]
            elif "uri" in img and not img["uri"].startswith("data:"):
                raw = open(os.path.join(os.path.dirname(path), img["uri"]), "rb").read()
            else:
                raise ValueError("image without bufferView or file uri")
            px, _ch = decode_image(raw)
            result = len(out.textures)
            out.textures.append((kind, px))
        except Exception as e:  # the reference prints and carries on (IMGLoader.cpp:24-25)
            out.warnings.append("texture %d: %s" % (ref["index"], e))
        decoded[key] = result
        return result

    for m in doc.get("materials", []):
        bt = m.get("pbrMetallicRoughness", {}).get("baseColorTexture")
        et = m.get("emissiveTexture")
        out.material_diffuse_texture.append(load_texture(bt, "diffuse") if bt else -1)
        out.material_emissive_texture.append(load_texture(et, "emissive") if et else -1)
    if not doc.get("materials"):
        out.material_diffuse_texture, out.material_emissive_texture = [-1], [-1]
    mats = [_gltf_material(m) for m in doc.get("materials", [])]
    out.material_names = [m.get("name", "") for m in doc.get("materials", [])]
    out.materials = np.array(mats, dtype=pod.MAT_DT) if mats else np.array([pod.make_material()], dtype=pod.MAT_DT)

    # one mesh per primitive (what Assimp hands the reference)
    prim_mesh = {}
    for mi, mesh in enumerate(doc.get("meshes", [])):
        for pi, prim in enumerate(mesh["primitives"]):
            if prim.get("mode", 4) != 4:
                continue
            pos = accessor(prim["attributes"]["POSITION"]).astype(np.float32)
            nrm = accessor(prim["attributes"]["NORMAL"]).astype(np.float32) if "NORMAL" in prim["attributes"] else None
            uv = accessor(prim["attributes"]["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in prim["attributes"] else None
            idx = accessor(prim["indices"]).reshape(-1).astype(np.int64) if "indices" in prim else np.arange(len(pos))
            tri = idx.reshape(-1, 3)
            normals = nrm[tri] if nrm is not None else np.zeros((len(tri), 3, 3), np.float32)
            if uv is not None:
                uv = uv.copy()
                uv[:, 1] = 1.0 - uv[:, 1]  # aiProcess_FlipUVs, OBJLoader.cpp:219-220
            t = pod.make_triangles(pos[tri], normals=normals, uvs=uv[tri] if uv is not None else None)
            prim_mesh[(mi, pi)] = (len(out.meshes), prim.get("material", 0))
            out.meshes.append(t)
            out.mesh_names.append(mesh.get("name", "mesh%d" % mi))

    def walk(ni, parent):
        node = doc["nodes"][ni]
        m = parent @ _node_local_matrix(node)
        if "mesh" in node:
            pos, rot, scale = decompose_trs(m)
            for (mi, pi), (mesh_id, mat_id) in prim_mesh.items():
                if mi == node["mesh"]:
                    out.instances.append(dict(mesh=mesh_id, material=mat_id, position=pos.astype(np.float32), rotation=rot.astype(np.float32),
                                              scale=scale.astype(np.float32), name=node.get("name", "")))
        for c in node.get("children", []):
            walk(c, m)

    scene = doc["scenes"][doc.get("scene", 0)]
    for ni in scene["nodes"]:
        walk(ni, np.eye(4))
    return out


def load_obj(path):
    """Triangulated Wavefront .obj (v / vn / vt / f, fan triangulation), one mesh, default material."""
    v, vn, vt, faces = [], [], [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                v.append([float(x) for x in p[1:4]])
            elif p[0] == "vn":
                vn.append([float(x) for x in p[1:4]])
            elif p[0] == "vt":
                vt.append([float(x) for x in p[1:3]])
            elif p[0] == "f":
                corners = []
                for tok in p[1:]:
                    parts = tok.split("/")
                    vi = int(parts[0])
                    ti = int(parts[1]) if len(parts) > 1 and parts[1] else 0
                    ni = int(parts[2]) if len(parts) > 2 and parts[2] else 0
                    corners.append((vi, ti, ni))
                for k in range(1, len(corners) - 1):
                    faces.append((corners[0], corners[k], corners[k + 1]))
    v = np.array(v, np.float32)
    vn = np.array(vn, np.float32) if vn else None
    vt = np.array(vt, np.float32) if vt else None

    def fix(i, n):
        return i - 1 if i > 0 else n + i

    pos = np.array([[v[fix(c[0], len(v))] for c in f] for f in faces], np.float32)
    normals = None
    if vn is not None and all(c[2] for f in faces for c in f):
        normals = np.array([[vn[fix(c[2], len(vn))] for c in f] for f in faces], np.float32)
    uvs = None
    if vt is not None and all(c[1] for f in faces for c in f):
        uvs = np.array([[vt[fix(c[1], len(vt))] for c in f] for f in faces], np.float32)
        uvs[..., 1] = 1.0 - uvs[..., 1]
    out = LoadedScene()
    out.meshes = [pod.make_triangles(pos, normals=normals, uvs=uvs)]
    out.mesh_names = [path]
    out.materials = np.array([pod.make_material(type=pod.MAT_PLASTIC, albedo=(0.6, 0.6, 0.6), roughness=1.0 - np.sqrt(20.0) / 31.62278)], dtype=pod.MAT_DT)
    out.material_names = ["default"]
    out.material_diffuse_texture, out.material_emissive_texture = [-1], [-1]
    out.instances = [dict(mesh=0, material=0, position=np.zeros(3, np.float32), rotation=np.zeros(3, np.float32), scale=np.ones(3, np.float32), name=path)]
    return out


def write_obj(path, tris):
    """TRI_DT triangles as an indexed Wavefront .obj (v / vn / vt / f with all three indices; equal corners share one index).
    Floats are written with 9 significant digits, which a float32 survives exactly; vt carries 1 - v because readers
    (Assimp's aiProcess_FlipUVs in the reference, OBJLoader.cpp:213-239; `load_obj` above; the C++ reader) flip it back."""
    t = np.ascontiguousarray(tris, dtype=pod.TRI_DT)
    n = len(t)
    corners = np.zeros((n, 3, 8), np.float32)
    for k in range(3):
        corners[:, k, 0:3] = t["pos%d" % k]
        corners[:, k, 3:6] = t["normal%d" % k]
        corners[:, k, 6:8] = t["texCoord%d" % k]
    flat = corners.reshape(n * 3, 8)
    keys = np.ascontiguousarray(flat).view(np.dtype((np.void, 32))).reshape(-1)
    _, first, inverse = np.unique(keys, return_index=True, return_inverse=True)
    verts = flat[first]
    idx = inverse.reshape(n, 3) + 1
    with open(path, "w") as f:
        f.write("# %d triangles, %d vertices\n" % (n, len(verts)))
        f.write("".join("v %.9g %.9g %.9g\n" % (a, b, c) for a, b, c in verts[:, 0:3].tolist()))
        f.write("".join("vn %.9g %.9g %.9g\n" % (a, b, c) for a, b, c in verts[:, 3:6].tolist()))
        uv = verts[:, 6:8].copy()
        uv[:, 1] = np.float32(1.0) - uv[:, 1]
        f.write("".join("vt %.9g %.9g\n" % (a, b) for a, b in uv.tolist()))
        f.write("".join("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, b, b, b, c, c, c) for a, b, c in idx.tolist()))
    return len(verts)
