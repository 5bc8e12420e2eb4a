"""numpy views of the POD contracts in ``include/nexus_pod.h``.

Byte layouts follow the reference's device PODs (file:line relative to /root/reference/Nexus/src):
``D_BVH8Node`` Cuda/BVH/BVH8.cuh:47-63, ``D_Triangle`` Cuda/Geometry/Triangle.cuh:25-49,
``D_BVHInstance`` Cuda/BVH/BVHInstance.cuh:7-14, ``D_Material`` Cuda/Scene/Material.cuh:5-51,
``D_Light`` Cuda/Scene/Light.cuh:4-33, ``D_Camera`` Cuda/Scene/Camera.cuh:5-15,
``D_RenderSettings`` Cuda/Scene/Scene.cuh:10-17, ``D_Intersection`` Cuda/Geometry/Ray.cuh:5-15.
"""
import numpy as np

NODE_DT = np.dtype(
    [
        ("p", "<f4", 3),
        ("e", "u1", 3),
        ("imask", "u1"),
        ("childBaseIdx", "<u4"),
        ("triangleBaseIdx", "<u4"),
        ("meta", "u1", 8),
        ("qlox", "u1", 8),
        ("qloy", "u1", 8),
        ("qloz", "u1", 8),
        ("qhix", "u1", 8),
        ("qhiy", "u1", 8),
        ("qhiz", "u1", 8),
    ]
)
assert NODE_DT.itemsize == 80

TRI_DT = np.dtype(
    [
        ("pos0", "<f4", 3),
        ("pos1", "<f4", 3),
        ("pos2", "<f4", 3),
        ("normal0", "<f4", 3),
        ("normal1", "<f4", 3),
        ("normal2", "<f4", 3),
        ("texCoord0", "<f4", 2),
        ("texCoord1", "<f4", 2),
        ("texCoord2", "<f4", 2),
    ]
)
assert TRI_DT.itemsize == 96

INST_DT = np.dtype(
    [
        ("bvhIdx", "<u4"),
        ("invTransform", "<f4", 16),
        ("transform", "<f4", 16),
        ("boundsMin", "<f4", 3),
        ("boundsMax", "<f4", 3),
        ("materialId", "<i4"),
    ]
)
assert INST_DT.itemsize == 160

# The 28-byte union is exposed as raw floats: diffuse/dielectric/plastic use u[0:3]=albedo, u[3]=roughness,
# u[4]=ior; conductor uses u[0:3]=ior, u[3:6]=k, u[6]=roughness.
MAT_DT = np.dtype(
    {
        "names": ["u", "emissive", "intensity", "opacity", "diffuseMapId", "emissiveMapId", "type"],
        "formats": [("<f4", 7), ("<f4", 3), "<f4", "<f4", "<i4", "<i4", "i1"],
        "offsets": [0, 28, 40, 44, 48, 52, 56],
        "itemsize": 60,
    }
)

LIGHT_DT = np.dtype(
    {"names": ["meshId", "aux", "type"], "formats": ["<u4", "<u4", "i1"], "offsets": [0, 4, 8], "itemsize": 12}
)

CAM_DT = np.dtype(
    {
        "names": ["position", "right", "up", "lensRadius", "lowerLeftCorner", "viewportX", "viewportY", "resolution"],
        "formats": [("<f4", 3), ("<f4", 3), ("<f4", 3), "<f4", ("<f4", 3), ("<f4", 3), ("<f4", 3), ("<u4", 2)],
        "offsets": [0, 12, 24, 36, 40, 52, 64, 80],
        "itemsize": 88,
    }
)

SETTINGS_DT = np.dtype(
    {
        "names": ["useMIS", "pathLength", "backgroundColor", "backgroundIntensity"],
        "formats": ["u1", "u1", ("<f4", 3), "<f4"],
        "offsets": [0, 1, 4, 16],
        "itemsize": 20,
    }
)

RAY_DT = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3)])
HIT_DT = np.dtype([("hitDistance", "<f4"), ("u", "<f4"), ("v", "<f4"), ("triIdx", "<u4"), ("instanceIdx", "<u4")])
LOADED_INST_DT = np.dtype([("mesh", "<i4"), ("material", "<i4"), ("position", "<f4", 3), ("rotation", "<f4", 3), ("scale", "<f4", 3)])
assert LOADED_INST_DT.itemsize == 44
# BSDF test hooks (nx_bsdf_query 32 B, nx_bsdf_result 48 B)
BSDF_QUERY_DT = np.dtype([("wi", "<f4", 3), ("rng", "<u4"), ("wo", "<f4", 3), ("pad_", "<u4")])
BSDF_RESULT_DT = np.dtype([("wo", "<f4", 3), ("pdf", "<f4"), ("throughput", "<f4", 3), ("ok", "<u4"), ("rngOut", "<u4"), ("pad_", "<u4", 3)])
assert BSDF_QUERY_DT.itemsize == 32 and BSDF_RESULT_DT.itemsize == 48

MAT_DIFFUSE, MAT_DIELECTRIC, MAT_PLASTIC, MAT_CONDUCTOR = 0, 1, 2, 3
LIGHT_POINT, LIGHT_AREA, LIGHT_MESH = 0, 1, 2
RNG_REFERENCE_SLOT, RNG_PIXEL_KEYED = 0, 1
COMPACT_FAST, COMPACT_ORDERED = 0, 1
CONDUCTOR_REFERENCE, CONDUCTOR_EXTENDED = 0, 1
ORDER_ROWS, ORDER_TILES = 0, 1  # nxhip_set_pixel_order
PATH_MAX_LENGTH = 100
# include/nexus_fmath.h NXF_OP_*: the shared transcendental functions, by number (double: 0-6; float, carried in doubles: 7-12)
NXF_OPS = {"sin": 0, "cos": 1, "exp": 2, "log": 3, "pow": 4, "atan2": 5, "asin": 6, "sinf": 7, "cosf": 8, "expf": 9, "logf": 10, "atan2f": 11, "asinf": 12}
MISS_DISTANCE = np.float32(1e30)


def make_triangles(pos, normals=None, uvs=None):
    """pos: (n,3,3) float array of vertex positions -> TRI_DT array (normals default to the face normal)."""
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    n = pos.shape[0]
    t = np.zeros(n, dtype=TRI_DT)
    t["pos0"], t["pos1"], t["pos2"] = pos[:, 0], pos[:, 1], pos[:, 2]
    if normals is None:
        fn = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0])
        ln = np.linalg.norm(fn, axis=1, keepdims=True)
        fn = np.where(ln > 0, fn / np.maximum(ln, 1e-30), np.array([0, 0, 1], np.float32)).astype(np.float32)
        t["normal0"] = t["normal1"] = t["normal2"] = fn
    else:
        normals = np.asarray(normals, dtype=np.float32)
        t["normal0"], t["normal1"], t["normal2"] = normals[:, 0], normals[:, 1], normals[:, 2]
    if uvs is not None:
        uvs = np.asarray(uvs, dtype=np.float32)
        t["texCoord0"], t["texCoord1"], t["texCoord2"] = uvs[:, 0], uvs[:, 1], uvs[:, 2]
    return t


def make_material(type=MAT_DIFFUSE, albedo=(0.8, 0.8, 0.8), roughness=0.0, ior=1.45, emissive=(0, 0, 0), intensity=0.0,
                  opacity=1.0, conductor_ior=None, conductor_k=None, diffuse_map=-1, emissive_map=-1):
    m = np.zeros((), dtype=MAT_DT)
    if type == MAT_CONDUCTOR:
        m["u"][0:3] = conductor_ior if conductor_ior is not None else (0.2, 0.9, 1.1)
        m["u"][3:6] = conductor_k if conductor_k is not None else (3.9, 2.4, 2.2)
        m["u"][6] = roughness
    else:
        m["u"][0:3] = albedo
        m["u"][3] = roughness
        m["u"][4] = ior
    m["emissive"] = emissive
    m["intensity"] = intensity
    m["opacity"] = opacity
    m["diffuseMapId"] = diffuse_map
    m["emissiveMapId"] = emissive_map
    m["type"] = type
    return m
