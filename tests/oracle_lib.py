"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg.
The product package ``nexus_amd`` never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from nexus_amd import pod

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_ORACLE_DIR, f) for f in os.listdir(_ORACLE_DIR) if f.endswith((".c", ".h"))]
    srcs += [os.path.join(_ORACLE_DIR, "..", "include", f) for f in ("nexus_pod.h", "nexus_fmath.h")]  # shared with the product
    stale = force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _ORACLE_DIR, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


class _Bvh2(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("nodeCount", C.c_uint32), ("triIdx", C.c_void_p), ("triCount", C.c_uint32)]


class _Bvh8(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("nodeCount", C.c_uint32), ("primIdx", C.c_void_p), ("primCount", C.c_uint32)]


class _Blas(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("tris", C.c_void_p), ("triIdx", C.c_void_p), ("nodeCount", C.c_uint32),
                ("triCount", C.c_uint32)]


class _TexDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba8", C.c_void_p)]


class _Scene(C.Structure):
    _fields_ = [
        ("tlasNodes", C.c_void_p), ("tlasInstIdx", C.c_void_p), ("tlasNodeCount", C.c_uint32),
        ("instances", C.c_void_p), ("instanceCount", C.c_uint32),
        ("blas", C.c_void_p), ("blasCount", C.c_uint32),
        ("materials", C.c_void_p), ("materialCount", C.c_uint32),
        ("lights", C.c_void_p), ("lightCount", C.c_uint32),
        ("diffuseMaps", C.c_void_p), ("emissiveMaps", C.c_void_p), ("hdrMap", C.c_void_p),
        ("camera", C.c_uint8 * 88), ("settings", C.c_uint8 * 20),
        ("envSampling", C.c_int32), ("envMarginalCdf", C.c_void_p), ("envRowCdf", C.c_void_p), ("envDensity", C.c_void_p),
    ]


class TraceStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes", C.c_uint64), ("tris", C.c_uint64), ("instances", C.c_uint64),
                ("maxStack", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class QueueSizes(C.Structure):
    _fields_ = [(n, C.c_int32 * pod.PATH_MAX_LENGTH) for n in
                ("traceSize", "traceShadowSize", "diffuseSize", "plasticSize", "dielectricSize", "conductorSize")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_bvh2_build.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_Bvh2)]
        L.orc_bvh2_free.argtypes = [C.POINTER(_Bvh2)]
        L.orc_bvh8_build.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(_Bvh8)]
        L.orc_bvh8_free.argtypes = [C.POINTER(_Bvh8)]
        L.orc_tlas_build.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_Bvh8)]
        L.orc_mat4_invert.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_mat4_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mat4_from_trs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_instance_init.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_camera_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, C.c_float,
                                      C.c_float]
        L.orc_trace_closest.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(TraceStats)]
        L.orc_trace_any.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(TraceStats)]
        L.orc_trace_closest_mt.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
        L.orc_brute_closest.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_brute_any.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_bvh2_trace_closest.argtypes = [C.POINTER(_Bvh2), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_child_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.orc_jenkins.argtypes = [C.c_uint32]
        L.orc_jenkins.restype = C.c_uint32
        for f in ("orc_rng_init_pixel", "orc_rng_init_keyed"):
            getattr(L, f).argtypes = [C.c_uint32] * 4
            getattr(L, f).restype = C.c_uint32
        L.orc_rng_init_index.argtypes = [C.c_uint32] * 3
        L.orc_rng_init_index.restype = C.c_uint32
        L.orc_rand.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_rand.restype = C.c_float
        L.orc_bsdf_sample.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.orc_bsdf_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.orc_bsdf_sample_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_bsdf_eval_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_tex2d.argtypes = [C.POINTER(_TexDesc), C.c_float, C.c_float, C.c_void_p]
        L.orc_fmath_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_wavefront_create.argtypes = [C.POINTER(_Scene), C.c_uint32, C.c_void_p, C.c_int, C.c_int]
        L.orc_wavefront_create.restype = C.c_void_p
        L.orc_wavefront_destroy.argtypes = [C.c_void_p]
        L.orc_wavefront_render.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.orc_wavefront_accumulate.argtypes = [C.c_void_p, C.c_uint32]
        for f in ("orc_wavefront_radiance", "orc_wavefront_accumulation", "orc_wavefront_rgba8", "orc_wavefront_queue_sizes"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_void_p
        L.orc_wavefront_trace_stats.argtypes = [C.c_void_p, C.POINTER(TraceStats), C.POINTER(TraceStats)]
        L.orc_tonemap_rgba8.argtypes = [C.c_void_p]
        L.orc_tonemap_rgba8.restype = C.c_uint32
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def fmath_batch(op, a, b=None):
    """include/nexus_fmath.h as the oracle compiles it: out[i] = nxf_apply(op, a[i], b[i])"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
    out = np.zeros(len(a), dtype=np.float64)
    lib().orc_fmath_batch(int(op), _ptr(a), None if bb is None else _ptr(bb), len(a), _ptr(out))
    return out


def bsdf_sample_batch(material, queries):
    """orc_bsdf_sample over an array of pod.BSDF_QUERY_DT -> pod.BSDF_RESULT_DT (the oracle twin of capi.Context.bsdf_sample_batch)"""
    m = np.array([material], dtype=pod.MAT_DT)
    q = np.ascontiguousarray(queries, dtype=pod.BSDF_QUERY_DT)
    out = np.zeros(len(q), dtype=pod.BSDF_RESULT_DT)
    lib().orc_bsdf_sample_batch(_ptr(m), _ptr(q), len(q), _ptr(out))
    return out


def bsdf_eval_batch(material, queries):
    m = np.array([material], dtype=pod.MAT_DT)
    q = np.ascontiguousarray(queries, dtype=pod.BSDF_QUERY_DT)
    out = np.zeros(len(q), dtype=pod.BSDF_RESULT_DT)
    lib().orc_bsdf_eval_batch(_ptr(m), _ptr(q), len(q), _ptr(out))
    return out


def _copy_out(addr, count, dtype):
    if count == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


def bvh2_build(tris):
    tris = np.ascontiguousarray(tris, dtype=pod.TRI_DT)
    b = _Bvh2()
    assert lib().orc_bvh2_build(_ptr(tris), len(tris), C.byref(b)) == 0
    node_dt = np.dtype([("aabbMin", "<f4", 3), ("aabbMax", "<f4", 3), ("leftFirst", "<u4"), ("triCount", "<u4")])
    nodes = _copy_out(b.nodes, b.nodeCount, node_dt)
    idx = _copy_out(b.triIdx, b.triCount, np.uint32)
    lib().orc_bvh2_free(C.byref(b))
    return nodes, idx


def bvh8_build(tris, clamp_qhi=1):
    tris = np.ascontiguousarray(tris, dtype=pod.TRI_DT)
    b = _Bvh8()
    assert lib().orc_bvh8_build(_ptr(tris), len(tris), int(clamp_qhi), C.byref(b)) == 0
    nodes = _copy_out(b.nodes, b.nodeCount, pod.NODE_DT)
    idx = _copy_out(b.primIdx, b.primCount, np.uint32)
    lib().orc_bvh8_free(C.byref(b))
    return nodes, idx


def tlas_build(instances):
    instances = np.ascontiguousarray(instances, dtype=pod.INST_DT)
    b = _Bvh8()
    assert lib().orc_tlas_build(_ptr(instances), len(instances), C.byref(b)) == 0
    nodes = _copy_out(b.nodes, b.nodeCount, pod.NODE_DT)
    idx = _copy_out(b.primIdx, b.primCount, np.uint32)
    lib().orc_bvh8_free(C.byref(b))
    return nodes, idx


def mat4_from_trs(pos=(0, 0, 0), rot_deg=(0, 0, 0), scale=(1, 1, 1)):
    out = np.zeros(16, np.float32)
    p, r, s = (np.asarray(x, np.float32) for x in (pos, rot_deg, scale))
    lib().orc_mat4_from_trs(_ptr(p), _ptr(r), _ptr(s), _ptr(out))
    return out


def mat4_invert(m):
    m = np.ascontiguousarray(m, np.float32)
    out = np.zeros(16, np.float32)
    lib().orc_mat4_invert(_ptr(m), _ptr(out))
    return out


def instance_init(bvh_idx, material_id, transform, blas_root_node):
    inst = np.zeros(1, dtype=pod.INST_DT)
    t = np.ascontiguousarray(transform, np.float32)
    root = np.ascontiguousarray(blas_root_node, dtype=pod.NODE_DT).reshape(1)
    lib().orc_instance_init(_ptr(inst), int(bvh_idx), int(material_id), _ptr(t), _ptr(root))
    return inst[0]


def camera_init(position, forward, hfov_deg, width, height, focus_dist=5.0, defocus_deg=0.0):
    cam = np.zeros(1, dtype=pod.CAM_DT)
    p = np.asarray(position, np.float32)
    f = np.asarray(forward, np.float32)
    lib().orc_camera_init(_ptr(cam), _ptr(p), _ptr(f), hfov_deg, width, height, focus_dist, defocus_deg)
    return cam[0]


class OracleScene:
    """Owns the numpy buffers behind an ``orc_scene``."""

    def __init__(self, blas_list, instances, tlas_nodes, tlas_idx, materials=None, lights=None, camera=None, settings=None,
                 diffuse_maps=(), emissive_maps=(), hdr_map=None, env_sampling=False):
        # blas_list: list of (nodes NODE_DT[], tris TRI_DT[], triIdx u32[])
        self.keep = []
        self.blas_arr = (_Blas * max(1, len(blas_list)))()
        for i, (nodes, tris, idx) in enumerate(blas_list):
            nodes = np.ascontiguousarray(nodes, dtype=pod.NODE_DT)
            tris = np.ascontiguousarray(tris, dtype=pod.TRI_DT)
            idx = np.ascontiguousarray(idx, dtype=np.uint32)
            self.keep += [nodes, tris, idx]
            self.blas_arr[i] = _Blas(_ptr(nodes), _ptr(tris), _ptr(idx), len(nodes), len(tris))
        self.instances = np.ascontiguousarray(instances, dtype=pod.INST_DT)
        self.tlas_nodes = np.ascontiguousarray(tlas_nodes, dtype=pod.NODE_DT)
        self.tlas_idx = np.ascontiguousarray(tlas_idx, dtype=np.uint32)
        self.materials = np.ascontiguousarray(materials if materials is not None else np.zeros(1, pod.MAT_DT), dtype=pod.MAT_DT)
        self.lights = np.ascontiguousarray(lights if lights is not None else np.zeros(0, pod.LIGHT_DT), dtype=pod.LIGHT_DT)
        s = _Scene()
        s.tlasNodes, s.tlasInstIdx, s.tlasNodeCount = _ptr(self.tlas_nodes), _ptr(self.tlas_idx), len(self.tlas_nodes)
        s.instances, s.instanceCount = _ptr(self.instances), len(self.instances)
        s.blas, s.blasCount = C.cast(self.blas_arr, C.c_void_p), len(blas_list)
        s.materials, s.materialCount = _ptr(self.materials), len(self.materials)
        s.lights, s.lightCount = (_ptr(self.lights) if len(self.lights) else None), len(self.lights)
        self._dmaps = self._tex_array(diffuse_maps)
        self._emaps = self._tex_array(emissive_maps)
        s.diffuseMaps = C.cast(self._dmaps, C.c_void_p) if self._dmaps is not None else None
        s.emissiveMaps = C.cast(self._emaps, C.c_void_p) if self._emaps is not None else None
        self._hdr = self._tex_array([hdr_map]) if hdr_map is not None else None
        s.hdrMap = C.cast(self._hdr, C.c_void_p) if self._hdr is not None else None
        s.envSampling = 0
        if env_sampling and hdr_map is not None:  # the extension of nxhip_set_env_sampling, restated in orc_wavefront.c
            h, w = np.asarray(hdr_map).shape[:2]
            self.env_marginal = np.zeros(h, np.float32)
            self.env_row = np.zeros(h * w, np.float32)
            self.env_density = np.zeros(h * w, np.float32)
            lib().orc_env_distribution.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            lib().orc_env_distribution(C.cast(self._hdr, C.c_void_p), _ptr(self.env_marginal), _ptr(self.env_row), _ptr(self.env_density))
            s.envSampling = 1
            s.envMarginalCdf, s.envRowCdf, s.envDensity = _ptr(self.env_marginal), _ptr(self.env_row), _ptr(self.env_density)
        self.c = s
        if camera is not None:
            self.set_camera(camera)
        self.set_settings(settings if settings is not None else make_settings())

    def _tex_array(self, maps):
        maps = list(maps)
        if not maps:
            return None
        arr = (_TexDesc * len(maps))()
        for i, img in enumerate(maps):
            img = np.ascontiguousarray(img, dtype=np.uint8)
            assert img.ndim == 3 and img.shape[2] == 4
            self.keep.append(img)
            arr[i] = _TexDesc(img.shape[1], img.shape[0], _ptr(img))
        return arr

    def set_camera(self, cam):
        cam = np.ascontiguousarray(cam, dtype=pod.CAM_DT).reshape(1)
        C.memmove(self.c.camera, _ptr(cam), 88)

    def set_settings(self, st):
        st = np.ascontiguousarray(st, dtype=pod.SETTINGS_DT).reshape(1)
        C.memmove(self.c.settings, _ptr(st), 20)

    # ---- traversal
    def trace_closest(self, rays, stats=None, threads=1):
        rays = np.ascontiguousarray(rays, dtype=pod.RAY_DT)
        hits = np.zeros(len(rays), dtype=pod.HIT_DT)
        if threads > 1:
            lib().orc_trace_closest_mt(C.byref(self.c), _ptr(rays), len(rays), _ptr(hits), threads)
        else:
            lib().orc_trace_closest(C.byref(self.c), _ptr(rays), len(rays), _ptr(hits), C.byref(stats) if stats is not None else None)
        return hits

    def trace_any(self, rays, tmax, stats=None):
        rays = np.ascontiguousarray(rays, dtype=pod.RAY_DT)
        tmax = np.ascontiguousarray(tmax, dtype=np.float32)
        occ = np.zeros(len(rays), dtype=np.uint8)
        lib().orc_trace_any(C.byref(self.c), _ptr(rays), _ptr(tmax), len(rays), _ptr(occ), C.byref(stats) if stats is not None else None)
        return occ

    def brute_closest(self, rays):
        rays = np.ascontiguousarray(rays, dtype=pod.RAY_DT)
        hits = np.zeros(len(rays), dtype=pod.HIT_DT)
        lib().orc_brute_closest(C.byref(self.c), _ptr(rays), len(rays), _ptr(hits))
        return hits

    def brute_any(self, rays, tmax):
        rays = np.ascontiguousarray(rays, dtype=pod.RAY_DT)
        tmax = np.ascontiguousarray(tmax, dtype=np.float32)
        occ = np.zeros(len(rays), dtype=np.uint8)
        lib().orc_brute_any(C.byref(self.c), _ptr(rays), _ptr(tmax), len(rays), _ptr(occ))
        return occ


def make_settings(use_mis=True, path_length=4, background=(1, 1, 1), background_intensity=0.0):
    s = np.zeros((), dtype=pod.SETTINGS_DT)
    s["useMIS"] = 1 if use_mis else 0
    s["pathLength"] = path_length
    s["backgroundColor"] = background
    s["backgroundIntensity"] = background_intensity
    return s


class Wavefront:
    def __init__(self, scene: OracleScene, local_count, pixel_map=None, rng_mode=pod.RNG_REFERENCE_SLOT,
                 conductor_mode=pod.CONDUCTOR_REFERENCE):
        self.scene = scene
        self.n = int(local_count)
        self.pixel_map = None if pixel_map is None else np.ascontiguousarray(pixel_map, dtype=np.uint32)
        self.h = lib().orc_wavefront_create(C.byref(scene.c), self.n, _ptr(self.pixel_map) if self.pixel_map is not None else None,
                                            rng_mode, conductor_mode)

    def render(self, frame, threads=1):
        lib().orc_wavefront_render(self.h, frame, threads)

    def accumulate(self, frame):
        lib().orc_wavefront_accumulate(self.h, frame)

    def radiance(self):
        return _copy_out(lib().orc_wavefront_radiance(self.h), self.n * 3, np.float32).reshape(self.n, 3)

    def accumulation(self):
        return _copy_out(lib().orc_wavefront_accumulation(self.h), self.n * 3, np.float32).reshape(self.n, 3)

    def rgba8(self):
        return _copy_out(lib().orc_wavefront_rgba8(self.h), self.n, np.uint32)

    def queue_sizes(self):
        q = QueueSizes.from_address(lib().orc_wavefront_queue_sizes(self.h))
        return {n: np.array(getattr(q, n)[:], dtype=np.int32) for n, _ in QueueSizes._fields_}

    def trace_stats(self):
        a, b = TraceStats(), TraceStats()
        lib().orc_wavefront_trace_stats(self.h, C.byref(a), C.byref(b))
        return a.as_dict(), b.as_dict()

    def close(self):
        if self.h:
            lib().orc_wavefront_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
