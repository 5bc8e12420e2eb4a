"""ctypes binding of ``nexus_amd/lib/libnexus_amd.so`` — the C-ABI declared in ``include/nexus_hip.h`` (device
layer) and ``include/nexus_host.h`` (host builders).

There is no Python / CPU fallback: if the library is missing or a device call fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

from . import pod

_HERE = os.path.dirname(os.path.abspath(__file__))
# NEXUS_AMD_LIB: another build of the same library (tools/ab_bench.sh compares kernel variants); still no fallback
LIB_PATH = os.environ.get("NEXUS_AMD_LIB") or os.path.join(_HERE, "lib", "libnexus_amd.so")

# every symbol include/nexus_hip.h and include/nexus_host.h declare
HIP_SYMBOLS = [
    "nxhip_last_error", "nxhip_device_count", "nxhip_create", "nxhip_destroy", "nxhip_resize", "nxhip_sync",
    "nxhip_upload_blas", "nxhip_clear_blas", "nxhip_set_tlas", "nxhip_set_materials", "nxhip_set_lights",
    "nxhip_upload_texture", "nxhip_clear_textures", "nxhip_set_camera", "nxhip_set_render_settings", "nxhip_set_modes",
    "nxhip_set_pixel_map", "nxhip_set_frames_per_pass", "nxhip_reset_frame_number", "nxhip_set_frame_number", "nxhip_frame_number",
    "nxhip_render_frame", "nxhip_accumulate", "nxhip_render", "nxhip_read_radiance", "nxhip_read_accumulation",
    "nxhip_read_rgba8", "nxhip_write_accumulation", "nxhip_bind_radiance", "nxhip_read_full_accumulation", "nxhip_read_full_rgba8", "nxhip_radiance_device_ptr", "nxhip_accumulation_device_ptr", "nxhip_accumulate_external", "nxhip_compose_tiles",
    "nxhip_read_queue_sizes", "nxhip_set_pixel_query", "nxhip_get_selected_instance", "nxhip_trace_batch",
    "nxhip_trace_shadow_batch", "nxhip_bsdf_sample_batch", "nxhip_bsdf_eval_batch", "nxhip_tex2d_batch", "nxhip_enable_trace_stats", "nxhip_read_trace_stats", "nxhip_enable_kernel_timing",
    "nxhip_read_kernel_times", "nxhip_read_graph_timeline", "nxhip_has_gfx950_code", "nxhip_build_info", "nxhip_set_pixel_order", "nxhip_sync_timeout", "nxhip_debug_set_requeue", "nxhip_debug_thin_counts_of_pass", "nxhip_debug_write_blas_node", "nxhip_rebuild_tlas", "nxhip_read_tlas_index", "nxhip_release_queues", "nxhip_set_device_builder",
    "nxhip_set_instance_transforms", "nxhip_read_tlas", "nxhip_set_passes_in_flight", "nxhip_set_tail_bounce", "nxhip_set_entry_points", "nxhip_read_entry_states", "nxhip_debug_set_thin", "nxhip_debug_set_thin_pool", "nxhip_debug_thin_counts", "nxhip_build_blas", "nxhip_read_blas", "nxhip_set_env_sampling",
    "nxhip_tile_pixel_map", "nxhip_mgpu_unique_id", "nxhip_mgpu_init", "nxhip_mgpu_attach", "nxhip_mgpu_gather", "nxhip_mgpu_read_rgba8",
    "nxhip_mgpu_read_accumulation", "nxhip_mgpu_shutdown", "nxhip_fmath_batch", "nxhip_abi_stamp", "nxhip_check_library", "nxhip_build_blas_batch", "nxhip_read_blas_batch", "nxhip_debug_set_scan_epoch",
]
HOST_SYMBOLS = [
    "nxh_bvh8_build", "nxh_tlas_build", "nxh_tlas_refit", "nxh_bvh8_node_count", "nxh_bvh8_prim_count", "nxh_bvh8_nodes",
    "nxh_bvh8_prim_indices", "nxh_bvh8_free", "nxh_bvh2_build", "nxh_mat4_from_trs", "nxh_mat4_invert",
    "nxh_instance_init", "nxh_camera_init", "nxh_loaded_texture_count", "nxh_loaded_texture_info", "nxh_loaded_texture_pixels",
    "nxh_loaded_material_textures", "nxh_loaded_warning_count", "nxh_loaded_warning", "nxh_decode_png", "nxh_write_png", "nxh_write_exr",
    "nxs_scene_add_hdr_map_file", "nxs_renderer_create", "nxs_renderer_destroy", "nxs_renderer_render", "nxs_renderer_reset", "nxs_renderer_on_resize",
    "nxs_renderer_save_screenshot", "nxs_renderer_save_exr", "nxs_renderer_frame_number", "nxs_renderer_megasamples_per_second", "nxs_renderer_device_context",
    "nxs_renderer_set_modes",
    "nxh_load_scene_file", "nxh_loaded_scene_free", "nxh_loaded_mesh_count", "nxh_loaded_mesh_triangle_count", "nxh_loaded_mesh_triangles",
    "nxh_loaded_material_count", "nxh_loaded_materials", "nxh_loaded_instance_count", "nxh_loaded_instances", "nxs_scene_load_file", "nxs_scene_set_instance_transform", "nxs_scene_assign_material", "nxs_scene_set_tlas_refit", "nxs_scene_set_device_tlas", "nxs_pathtracer_set_device_blas_build",
    "nxs_last_error", "nxs_scene_create", "nxs_scene_destroy", "nxs_scene_add_material", "nxs_scene_add_texture", "nxs_scene_set_hdr_map",
    "nxs_scene_add_mesh", "nxs_scene_create_instance", "nxs_scene_set_camera", "nxs_scene_set_render_settings", "nxs_scene_update",
    "nxs_scene_light_count", "nxs_scene_instance_count", "nxs_pathtracer_create", "nxs_pathtracer_destroy", "nxs_pathtracer_set_modes", "nxs_pathtracer_set_frames_per_pass", "nxs_pathtracer_set_passes_in_flight", "nxs_pathtracer_set_pixel_order", "nxs_pathtracer_set_entry_points",
    "nxs_pathtracer_update_device_scene", "nxs_pathtracer_render", "nxs_pathtracer_reset_frame_number", "nxs_pathtracer_frame_number",
    "nxs_pathtracer_read_pixels", "nxs_pathtracer_device_context",
]


class NexusError(RuntimeError):
    pass


class QueueSizes(C.Structure):
    _fields_ = [(n, C.c_int32 * pod.PATH_MAX_LENGTH) for n in
                ("traceSize", "traceShadowSize", "diffuseSize", "plasticSize", "dielectricSize", "conductorSize")]


class TraceStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes", C.c_uint64), ("tris", C.c_uint64), ("instances", C.c_uint64),
                ("waveIters", C.c_uint64), ("lanesActive", C.c_uint64), ("lanesNode", C.c_uint64), ("lanesPrim", C.c_uint64), ("cycles", C.c_uint64 * 8)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "cycles"}
        d["cycles"] = [int(x) for x in self.cycles]
        return d


class KernelTimes(C.Structure):
    _fields_ = [("ms", C.c_double * 7), ("launches", C.c_uint64 * 7)]


KERNEL_CLASSES = ("generate", "trace", "shadow", "logic", "shade", "accumulate", "thin")

API_VERSION = 7  # NXHIP_API_VERSION of the include/nexus_hip.h these bindings were written against


def abi_words():
    """The list nxhip_header_abi_stamp() hashes (include/nexus_hip.h), from THIS module's mirrors of the C structs: API version
    and the size / key offsets of everything that crosses the boundary."""
    off = lambda dt, name: dt.fields[name][1]  # noqa: E731
    return [
        API_VERSION, pod.PATH_MAX_LENGTH,
        pod.NODE_DT.itemsize, off(pod.NODE_DT, "meta"), pod.TRI_DT.itemsize, off(pod.TRI_DT, "texCoord0"),
        pod.INST_DT.itemsize, off(pod.INST_DT, "transform"), off(pod.INST_DT, "materialId"),
        pod.MAT_DT.itemsize, off(pod.MAT_DT, "emissive"), off(pod.MAT_DT, "type"), pod.LIGHT_DT.itemsize, off(pod.LIGHT_DT, "type"),
        pod.CAM_DT.itemsize, off(pod.CAM_DT, "resolution"), pod.SETTINGS_DT.itemsize, off(pod.SETTINGS_DT, "backgroundColor"),
        pod.RAY_DT.itemsize, pod.HIT_DT.itemsize, pod.BSDF_QUERY_DT.itemsize, pod.BSDF_RESULT_DT.itemsize, off(pod.BSDF_RESULT_DT, "rngOut"),
        C.sizeof(QueueSizes), C.sizeof(TraceStats), TraceStats.cycles.offset, C.sizeof(KernelTimes), len(KERNEL_CLASSES),
    ]


def abi_stamp(words=None):
    """FNV-1a over the little-endian 8-byte words: nxhip_header_abi_stamp() as these bindings would compute it"""
    h = 0xcbf29ce484222325
    for w in (abi_words() if words is None else words):
        for _ in range(8):
            h = ((h ^ (w & 0xff)) * 0x100000001b3) & 0xffffffffffffffff
            w >>= 8
    return h


def check_library(L, stamp=None):
    """Refuse a library that does not match these bindings — a stale build reached through NEXUS_AMD_LIB, a library linked
    from objects of different source states — with an error instead of a GPU fault in the first launch."""
    if not hasattr(L, "nxhip_check_library"):
        raise NexusError(f"{LIB_PATH} predates the ABI stamp (no nxhip_check_library): rebuild it with `make`")
    L.nxhip_check_library.argtypes = [C.c_uint64]
    L.nxhip_abi_stamp.restype = C.c_uint64
    L.nxhip_last_error.restype = C.c_char_p
    rc = L.nxhip_check_library(abi_stamp() if stamp is None else stamp)
    if rc != 0:
        raise NexusError("%s: %s" % (LIB_PATH, (L.nxhip_last_error() or b"").decode()))

_lib = None


def lib():
    """Load the shared library (once).  Raises NexusError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NexusError(f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()); there is no fallback path")
    L = C.CDLL(LIB_PATH)
    check_library(L)
    vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int32, C.c_float
    L.nxhip_last_error.restype = C.c_char_p
    L.nxhip_create.argtypes = [C.c_int, u32, u32, vp, C.POINTER(vp)]
    L.nxhip_destroy.argtypes = [vp]
    L.nxhip_destroy.restype = None
    L.nxhip_resize.argtypes = [vp, u32, u32]
    L.nxhip_sync.argtypes = [vp]
    L.nxhip_upload_blas.argtypes = [vp, vp, u32, vp, u32, vp, C.POINTER(i32)]
    L.nxhip_clear_blas.argtypes = [vp]
    L.nxhip_set_tlas.argtypes = [vp, vp, u32, vp, vp, u32]
    L.nxhip_set_materials.argtypes = [vp, vp, u32]
    L.nxhip_set_lights.argtypes = [vp, vp, u32]
    L.nxhip_upload_texture.argtypes = [vp, C.c_int, vp, u32, u32, C.POINTER(i32)]
    L.nxhip_clear_textures.argtypes = [vp]
    L.nxhip_set_camera.argtypes = [vp, vp]
    L.nxhip_set_render_settings.argtypes = [vp, vp]
    L.nxhip_set_modes.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.nxhip_set_pixel_map.argtypes = [vp, vp, u32]
    L.nxhip_reset_frame_number.argtypes = [vp]
    L.nxhip_set_frame_number.argtypes = [vp, u32]
    L.nxhip_frame_number.argtypes = [vp]
    L.nxhip_frame_number.restype = u32
    L.nxhip_render_frame.argtypes = [vp]
    L.nxhip_accumulate.argtypes = [vp]
    L.nxhip_render.argtypes = [vp, u32]
    L.nxhip_read_radiance.argtypes = [vp, vp]
    L.nxhip_read_accumulation.argtypes = [vp, vp]
    L.nxhip_read_rgba8.argtypes = [vp, vp]
    L.nxhip_write_accumulation.argtypes = [vp, vp, u32]
    L.nxhip_radiance_device_ptr.argtypes = [vp]
    L.nxhip_radiance_device_ptr.restype = vp
    L.nxhip_accumulation_device_ptr.argtypes = [vp]
    L.nxhip_accumulation_device_ptr.restype = vp
    L.nxhip_accumulate_external.argtypes = [vp, vp, u32, u32, u32, u32, vp]
    L.nxhip_compose_tiles.argtypes = [vp, vp, u32, vp, vp, vp]
    L.nxhip_set_frames_per_pass.argtypes = [vp, u32]
    L.nxhip_bind_radiance.argtypes = [vp, vp, u32]
    L.nxhip_read_full_accumulation.argtypes = [vp, vp]
    L.nxhip_read_full_rgba8.argtypes = [vp, vp]
    L.nxhip_read_queue_sizes.argtypes = [vp, C.POINTER(QueueSizes)]
    L.nxhip_set_pixel_query.argtypes = [vp, u32, u32]
    L.nxhip_get_selected_instance.argtypes = [vp, C.POINTER(i32)]
    L.nxhip_trace_batch.argtypes = [vp, vp, u32, vp]
    L.nxhip_trace_shadow_batch.argtypes = [vp, vp, vp, u32, vp]
    L.nxhip_bsdf_sample_batch.argtypes = [vp, vp, vp, u32, vp]
    L.nxhip_bsdf_eval_batch.argtypes = [vp, vp, vp, u32, vp]
    L.nxhip_tex2d_batch.argtypes = [vp, C.c_int, C.c_int, vp, u32, vp]
    L.nxhip_fmath_batch.argtypes = [vp, C.c_int, vp, vp, u32, vp]
    L.nxhip_enable_trace_stats.argtypes = [vp, C.c_int]
    L.nxhip_read_trace_stats.argtypes = [vp, C.POINTER(TraceStats), C.POINTER(TraceStats), C.c_int]
    L.nxhip_enable_kernel_timing.argtypes = [vp, C.c_int]
    L.nxhip_read_kernel_times.argtypes = [vp, C.POINTER(KernelTimes), C.c_int]
    L.nxhip_read_graph_timeline.argtypes = [vp, vp, vp, vp, u32, C.POINTER(u32)]
    L.nxhip_set_instance_transforms.argtypes = [vp, vp, vp, u32]
    L.nxhip_read_tlas.argtypes = [vp, vp, u32, vp, u32]
    L.nxhip_tile_pixel_map.argtypes = [u32, u32, C.c_int, C.c_int, u32, C.c_int, vp, C.POINTER(u32)]
    L.nxhip_mgpu_unique_id.argtypes = [vp]
    L.nxhip_mgpu_init.argtypes = [vp, C.c_int, C.c_int, vp, u32]
    L.nxhip_mgpu_attach.argtypes = [vp, vp, C.c_int, C.c_int, u32]
    L.nxhip_mgpu_gather.argtypes = [vp]
    L.nxhip_mgpu_read_rgba8.argtypes = [vp, vp]
    L.nxhip_mgpu_read_accumulation.argtypes = [vp, vp]
    L.nxhip_mgpu_shutdown.argtypes = [vp]
    # host builders
    L.nxh_bvh8_build.argtypes = [vp, u32, u32, C.POINTER(vp)]
    L.nxh_tlas_build.argtypes = [vp, u32, C.POINTER(vp)]
    L.nxh_tlas_refit.argtypes = [vp, u32, vp, vp, u32]
    L.nxh_bvh8_node_count.argtypes = [vp]
    L.nxh_bvh8_node_count.restype = u32
    L.nxh_bvh8_prim_count.argtypes = [vp]
    L.nxh_bvh8_prim_count.restype = u32
    L.nxh_bvh8_nodes.argtypes = [vp]
    L.nxh_bvh8_nodes.restype = vp
    L.nxh_bvh8_prim_indices.argtypes = [vp]
    L.nxh_bvh8_prim_indices.restype = vp
    L.nxh_bvh8_free.argtypes = [vp]
    L.nxh_bvh8_free.restype = None
    L.nxh_bvh2_build.argtypes = [vp, u32, u32, vp, vp]
    L.nxh_mat4_from_trs.argtypes = [vp, vp, vp, vp]
    L.nxh_mat4_from_trs.restype = None
    L.nxh_mat4_invert.argtypes = [vp, vp]
    L.nxh_mat4_invert.restype = None
    L.nxh_instance_init.argtypes = [vp, u32, i32, vp, vp]
    L.nxh_camera_init.argtypes = [vp, vp, vp, f32, u32, u32, f32, f32]
    L.nxh_load_scene_file.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.nxh_loaded_scene_free.argtypes = [vp]
    L.nxh_loaded_scene_free.restype = None
    for f in ("nxh_loaded_mesh_count", "nxh_loaded_material_count", "nxh_loaded_instance_count"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = u32
    L.nxh_loaded_mesh_triangle_count.argtypes = [vp, u32]
    L.nxh_loaded_mesh_triangle_count.restype = u32
    L.nxh_loaded_mesh_triangles.argtypes = [vp, u32, vp]
    L.nxh_loaded_materials.argtypes = [vp, vp]
    L.nxh_loaded_instances.argtypes = [vp, vp]
    L.nxs_scene_load_file.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.nxs_scene_set_instance_transform.argtypes = [vp, u32, vp, vp, vp]
    L.nxs_scene_set_tlas_refit.argtypes = [vp, C.c_int]
    # Scene / PathTracer facade
    L.nxs_last_error.restype = C.c_char_p
    L.nxs_scene_create.argtypes = [u32, u32, C.POINTER(vp)]
    L.nxs_scene_destroy.argtypes = [vp]
    L.nxs_scene_destroy.restype = None
    L.nxs_scene_add_material.argtypes = [vp, vp, C.POINTER(i32)]
    L.nxs_scene_add_texture.argtypes = [vp, C.c_int, vp, u32, u32, C.POINTER(i32)]
    L.nxs_scene_set_hdr_map.argtypes = [vp, vp, u32, u32]
    L.nxs_scene_add_mesh.argtypes = [vp, vp, u32, i32, C.POINTER(i32)]
    L.nxs_scene_create_instance.argtypes = [vp, u32, i32, vp, vp, vp, C.POINTER(i32)]
    L.nxs_scene_set_camera.argtypes = [vp, vp, vp, f32, f32, f32]
    L.nxs_scene_set_render_settings.argtypes = [vp, vp]
    L.nxs_scene_update.argtypes = [vp]
    L.nxs_scene_light_count.argtypes = [vp]
    L.nxs_scene_light_count.restype = u32
    L.nxs_scene_instance_count.argtypes = [vp]
    L.nxs_scene_instance_count.restype = u32
    L.nxs_pathtracer_create.argtypes = [u32, u32, C.c_int, C.POINTER(vp)]
    L.nxs_pathtracer_destroy.argtypes = [vp]
    L.nxs_pathtracer_destroy.restype = None
    L.nxs_pathtracer_set_modes.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.nxs_pathtracer_set_frames_per_pass.argtypes = [vp, C.c_uint32]
    L.nxs_pathtracer_set_passes_in_flight.argtypes = [vp, C.c_uint32]
    L.nxs_pathtracer_update_device_scene.argtypes = [vp, vp]
    L.nxs_pathtracer_render.argtypes = [vp, vp]
    L.nxs_pathtracer_reset_frame_number.argtypes = [vp]
    L.nxs_pathtracer_frame_number.argtypes = [vp]
    L.nxs_pathtracer_frame_number.restype = u32
    L.nxs_pathtracer_read_pixels.argtypes = [vp, vp]
    L.nxs_pathtracer_device_context.argtypes = [vp]
    L.nxs_pathtracer_device_context.restype = vp
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def check(rc, what=""):
    if rc != 0:
        msg = lib().nxhip_last_error()
        raise NexusError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


def _copy_out(addr, count, dtype):
    if count == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


# ---- host builders --------------------------------------------------------------------------------

def bvh8_build(tris, threads=0):
    """BVH8Builder(tris).Init().Build() -> (nodes NODE_DT[], triIdx u32[])."""
    tris = np.ascontiguousarray(tris, dtype=pod.TRI_DT)
    h = C.c_void_p()
    rc = lib().nxh_bvh8_build(_ptr(tris), len(tris), threads, C.byref(h))
    if rc != 0:
        raise NexusError(f"nxh_bvh8_build failed ({rc})")
    try:
        nodes = _copy_out(lib().nxh_bvh8_nodes(h), lib().nxh_bvh8_node_count(h), pod.NODE_DT)
        idx = _copy_out(lib().nxh_bvh8_prim_indices(h), lib().nxh_bvh8_prim_count(h), np.uint32)
    finally:
        lib().nxh_bvh8_free(h)
    return nodes, idx


def tlas_build(instances):
    instances = np.ascontiguousarray(instances, dtype=pod.INST_DT)
    h = C.c_void_p()
    rc = lib().nxh_tlas_build(_ptr(instances), len(instances), C.byref(h))
    if rc != 0:
        raise NexusError(f"nxh_tlas_build failed ({rc})")
    try:
        nodes = _copy_out(lib().nxh_bvh8_nodes(h), lib().nxh_bvh8_node_count(h), pod.NODE_DT)
        idx = _copy_out(lib().nxh_bvh8_prim_indices(h), lib().nxh_bvh8_prim_count(h), np.uint32)
    finally:
        lib().nxh_bvh8_free(h)
    return nodes, idx


BVH2_NODE_DT = np.dtype([("aabbMin", "<f4", 3), ("aabbMax", "<f4", 3), ("leftFirst", "<u4"), ("triCount", "<u4")])


def bvh2_build(tris, threads=0):
    tris = np.ascontiguousarray(tris, dtype=pod.TRI_DT)
    nodes = np.zeros(2 * len(tris) - 1, dtype=BVH2_NODE_DT)
    idx = np.zeros(len(tris), dtype=np.uint32)
    rc = lib().nxh_bvh2_build(_ptr(tris), len(tris), threads, _ptr(nodes), _ptr(idx))
    if rc != 0:
        raise NexusError(f"nxh_bvh2_build failed ({rc})")
    return nodes, idx


def mat4_from_trs(pos=(0, 0, 0), rot_deg=(0, 0, 0), scale=(1, 1, 1)):
    out = np.zeros(16, np.float32)
    p, r, s = (np.asarray(x, np.float32) for x in (pos, rot_deg, scale))
    lib().nxh_mat4_from_trs(_ptr(p), _ptr(r), _ptr(s), _ptr(out))
    return out


def mat4_invert(m):
    m = np.ascontiguousarray(m, np.float32)
    out = np.zeros(16, np.float32)
    lib().nxh_mat4_invert(_ptr(m), _ptr(out))
    return out


def instance_init(bvh_idx, material_id, transform, blas_root_node):
    inst = np.zeros(1, dtype=pod.INST_DT)
    t = np.ascontiguousarray(transform, np.float32)
    root = np.ascontiguousarray(blas_root_node, dtype=pod.NODE_DT).reshape(1)
    rc = lib().nxh_instance_init(_ptr(inst), int(bvh_idx), int(material_id), _ptr(t), _ptr(root))
    if rc != 0:
        raise NexusError(f"nxh_instance_init failed ({rc})")
    return inst[0]


def camera_init(position, forward, hfov_deg, width, height, focus_dist=5.0, defocus_deg=0.0):
    cam = np.zeros(1, dtype=pod.CAM_DT)
    p = np.asarray(position, np.float32)
    f = np.asarray(forward, np.float32)
    rc = lib().nxh_camera_init(_ptr(cam), _ptr(p), _ptr(f), hfov_deg, width, height, focus_dist, defocus_deg)
    if rc != 0:
        raise NexusError(f"nxh_camera_init failed ({rc})")
    return cam[0]


# ---- device context -------------------------------------------------------------------------------

def tlas_refit(nodes, inst_idx, instances):
    """In-place-style refit of a TLAS (returns the new node array): same topology, bounds from the instances' current boxes."""
    nodes = np.ascontiguousarray(nodes, dtype=pod.NODE_DT).copy()
    inst_idx = np.ascontiguousarray(inst_idx, dtype=np.uint32)
    instances = np.ascontiguousarray(instances, dtype=pod.INST_DT)
    if lib().nxh_tlas_refit(_ptr(nodes), len(nodes), _ptr(inst_idx), _ptr(instances), len(instances)) != 0:
        raise NexusError("nxh_tlas_refit: malformed TLAS")
    return nodes


def load_scene_file(path):
    """nexus::OBJLoader::Parse through the C-ABI: (meshes [TRI_DT arrays], materials MAT_DT array, instances LOADED_INST_DT array)."""
    L = lib()
    h = C.c_void_p()
    if L.nxh_load_scene_file(str(path).encode(), C.byref(h)) != 0:
        raise NexusError("nxh_load_scene_file: " + L.nxs_last_error().decode())
    try:
        meshes = []
        for m in range(L.nxh_loaded_mesh_count(h)):
            t = np.zeros(L.nxh_loaded_mesh_triangle_count(h, m), dtype=pod.TRI_DT)
            if L.nxh_loaded_mesh_triangles(h, m, _ptr(t)) != 0:
                raise NexusError(L.nxs_last_error().decode())
            meshes.append(t)
        mats = np.zeros(L.nxh_loaded_material_count(h), dtype=pod.MAT_DT)
        if L.nxh_loaded_materials(h, _ptr(mats)) != 0:
            raise NexusError(L.nxs_last_error().decode())
        insts = np.zeros(L.nxh_loaded_instance_count(h), dtype=pod.LOADED_INST_DT)
        if L.nxh_loaded_instances(h, _ptr(insts)) != 0:
            raise NexusError(L.nxs_last_error().decode())
    finally:
        L.nxh_loaded_scene_free(h)
    return meshes, mats, insts


def load_scene_textures(path):
    """The images of a scene file as the C++ reader decodes them: (textures [(kind, HxWx4 uint8)], diffuse texture index
    per material, emissive texture index per material, warnings)."""
    L = lib()
    h = C.c_void_p()
    if L.nxh_load_scene_file(str(path).encode(), C.byref(h)) != 0:
        raise NexusError("nxh_load_scene_file: " + L.nxs_last_error().decode())
    try:
        L.nxh_loaded_texture_count.argtypes = [C.c_void_p]
        L.nxh_loaded_texture_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
        L.nxh_loaded_texture_pixels.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.nxh_loaded_material_textures.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.nxh_loaded_warning_count.argtypes = [C.c_void_p]
        L.nxh_loaded_warning.argtypes = [C.c_void_p, C.c_uint32]
        L.nxh_loaded_warning.restype = C.c_char_p
        texs = []
        for i in range(L.nxh_loaded_texture_count(h)):
            w, hh, k = C.c_uint32(0), C.c_uint32(0), C.c_int32(0)
            if L.nxh_loaded_texture_info(h, i, C.byref(w), C.byref(hh), C.byref(k)) != 0:
                raise NexusError(L.nxs_last_error().decode())
            px = np.zeros((hh.value, w.value, 4), dtype=np.uint8)
            if L.nxh_loaded_texture_pixels(h, i, _ptr(px)) != 0:
                raise NexusError(L.nxs_last_error().decode())
            texs.append(("emissive" if k.value == 1 else "diffuse", px))
        n = L.nxh_loaded_material_count(h)
        dt, et = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        if L.nxh_loaded_material_textures(h, _ptr(dt), _ptr(et)) != 0:
            raise NexusError(L.nxs_last_error().decode())
        warns = [L.nxh_loaded_warning(h, i).decode() for i in range(L.nxh_loaded_warning_count(h))]
    finally:
        L.nxh_loaded_scene_free(h)
    return texs, dt, et, warns


def decode_png(data):
    """nexus::IMGLoader::LoadIMG through the C-ABI: (HxWx4 uint8, channels of the file)"""
    L = lib()
    L.nxh_decode_png.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t]
    buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    w, h, ch = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    if L.nxh_decode_png(_ptr(buf), len(buf), C.byref(w), C.byref(h), C.byref(ch), None, 0) != 0:
        raise NexusError("nxh_decode_png: " + L.nxs_last_error().decode())
    out = np.zeros((h.value, w.value, 4), dtype=np.uint8)
    if L.nxh_decode_png(_ptr(buf), len(buf), C.byref(w), C.byref(h), C.byref(ch), _ptr(out), out.size) != 0:
        raise NexusError("nxh_decode_png: " + L.nxs_last_error().decode())
    return out, int(ch.value)


def decode_image(data):
    """IMGLoader::LoadIMG on an image file in memory — PNG, JPEG or Radiance .hdr by its signature: (HxWx4 uint8, channels)."""
    return decode_png(data)


class Context:
    """One ``nxhip_ctx`` (one GPU).  Thin 1:1 wrapper of the C-ABI; raises NexusError on any failure."""

    def __init__(self, width, height, device=0, stream=None):
        self.L = lib()
        self.width, self.height = int(width), int(height)
        h = C.c_void_p()
        check(self.L.nxhip_create(device, self.width, self.height, C.c_void_p(stream) if stream else None, C.byref(h)), "nxhip_create")
        self.h = h
        self.local_count = self.width * self.height
        self.frames_per_pass = 1

    def close(self):
        if getattr(self, "h", None):
            self.L.nxhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # scene
    def upload_blas(self, nodes, tris, tri_idx):
        nodes = np.ascontiguousarray(nodes, dtype=pod.NODE_DT)
        tris = np.ascontiguousarray(tris, dtype=pod.TRI_DT)
        tri_idx = np.ascontiguousarray(tri_idx, dtype=np.uint32)
        bid = C.c_int32(-1)
        check(self.L.nxhip_upload_blas(self.h, _ptr(nodes), len(nodes), _ptr(tris), len(tris), _ptr(tri_idx), C.byref(bid)), "nxhip_upload_blas")
        return bid.value

    def clear_blas(self):
        check(self.L.nxhip_clear_blas(self.h), "nxhip_clear_blas")

    def set_tlas(self, nodes, inst_idx, instances):
        nodes = np.ascontiguousarray(nodes, dtype=pod.NODE_DT)
        inst_idx = np.ascontiguousarray(inst_idx, dtype=np.uint32)
        instances = np.ascontiguousarray(instances, dtype=pod.INST_DT)
        check(self.L.nxhip_set_tlas(self.h, _ptr(nodes), len(nodes), _ptr(inst_idx), _ptr(instances), len(instances)), "nxhip_set_tlas")

    def set_materials(self, materials):
        materials = np.ascontiguousarray(materials, dtype=pod.MAT_DT)
        check(self.L.nxhip_set_materials(self.h, _ptr(materials), len(materials)), "nxhip_set_materials")

    def set_lights(self, lights):
        lights = np.ascontiguousarray(lights, dtype=pod.LIGHT_DT)
        check(self.L.nxhip_set_lights(self.h, _ptr(lights) if len(lights) else None, len(lights)), "nxhip_set_lights")

    def upload_texture(self, kind, rgba8):
        img = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert img.ndim == 3 and img.shape[2] == 4
        tid = C.c_int32(-1)
        check(self.L.nxhip_upload_texture(self.h, {"diffuse": 0, "emissive": 1, "hdr": 2}[kind], _ptr(img), img.shape[1], img.shape[0], C.byref(tid)),
              "nxhip_upload_texture")
        return tid.value

    def clear_textures(self):
        check(self.L.nxhip_clear_textures(self.h), "nxhip_clear_textures")

    def set_camera(self, cam):
        cam = np.ascontiguousarray(cam, dtype=pod.CAM_DT).reshape(1)
        check(self.L.nxhip_set_camera(self.h, _ptr(cam)), "nxhip_set_camera")

    def set_render_settings(self, st):
        st = np.ascontiguousarray(st, dtype=pod.SETTINGS_DT).reshape(1)
        check(self.L.nxhip_set_render_settings(self.h, _ptr(st)), "nxhip_set_render_settings")

    def set_modes(self, rng_mode=pod.RNG_REFERENCE_SLOT, compact_mode=pod.COMPACT_FAST, conductor_mode=pod.CONDUCTOR_REFERENCE):
        check(self.L.nxhip_set_modes(self.h, rng_mode, compact_mode, conductor_mode), "nxhip_set_modes")

    def set_pixel_map(self, pixel_map):
        if pixel_map is None:
            check(self.L.nxhip_set_pixel_map(self.h, None, 0), "nxhip_set_pixel_map")
            self.local_count = self.width * self.height
        else:
            pm = np.ascontiguousarray(pixel_map, dtype=np.uint32)
            check(self.L.nxhip_set_pixel_map(self.h, _ptr(pm), len(pm)), "nxhip_set_pixel_map")
            self.local_count = len(pm)

    def set_pixel_order(self, order):
        """nxhip_set_pixel_order: ORDER_ROWS (0, the reference's) or ORDER_TILES (1: 8 x 8 pixel tiles) over the full frame"""
        self.L.nxhip_set_pixel_order.argtypes = [C.c_void_p, C.c_int]
        check(self.L.nxhip_set_pixel_order(self.h, int(order)), "nxhip_set_pixel_order")
        self.local_count = self.width * self.height

    def build_blas(self, tris):
        """BLAS built on the device (binned SAH + SAH-DP collapse by default, see set_device_builder); returns the BLAS id"""
        t = np.ascontiguousarray(tris, dtype=pod.TRI_DT)
        bid = C.c_int32(-1)
        self.L.nxhip_build_blas.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]
        check(self.L.nxhip_build_blas(self.h, _ptr(t), len(t), C.byref(bid)), "nxhip_build_blas")
        return int(bid.value)

    def build_blas_batch(self, meshes):
        """the BLASes of several meshes in one device build (nxhip_build_blas_batch); returns their ids (consecutive)"""
        arrays = [np.ascontiguousarray(m, dtype=pod.TRI_DT) for m in meshes]
        ptrs = (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
        counts = np.array([len(a) for a in arrays], dtype=np.uint32)
        ids = np.full(len(arrays), -1, dtype=np.int32)
        self.L.nxhip_build_blas_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        check(self.L.nxhip_build_blas_batch(self.h, ptrs, _ptr(counts), len(arrays), _ptr(ids)), "nxhip_build_blas_batch")
        return [int(i) for i in ids]

    def read_blas_batch(self, first_id, tri_counts):
        """nodes and primitive index lists of consecutive BLAS ids: [(nodes, idx), ...]"""
        count = len(tri_counts)
        self.L.nxhip_read_blas_batch.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        node_counts = np.zeros(count, dtype=np.uint32)
        check(self.L.nxhip_read_blas_batch(self.h, first_id, count, None, 0, _ptr(node_counts), None, 0), "nxhip_read_blas_batch")
        nodes = np.zeros(int(node_counts.sum()), dtype=pod.NODE_DT)
        idx = np.zeros(int(np.sum(tri_counts)), dtype=np.uint32)
        check(self.L.nxhip_read_blas_batch(self.h, first_id, count, _ptr(nodes), len(nodes), _ptr(node_counts), _ptr(idx), len(idx)), "nxhip_read_blas_batch")
        out, na, pa = [], 0, 0
        for k in range(count):
            out.append((nodes[na:na + int(node_counts[k])], idx[pa:pa + int(tri_counts[k])]))
            na += int(node_counts[k])
            pa += int(tri_counts[k])
        return out

    def read_blas(self, blas_id, tri_count):
        self.L.nxhip_read_blas.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        n = C.c_uint32(0)
        check(self.L.nxhip_read_blas(self.h, blas_id, None, 0, None, 0, C.byref(n)), "nxhip_read_blas")
        nodes = np.zeros(n.value, dtype=pod.NODE_DT)
        idx = np.zeros(tri_count, dtype=np.uint32)
        check(self.L.nxhip_read_blas(self.h, blas_id, _ptr(nodes), n.value, _ptr(idx), tri_count, C.byref(n)), "nxhip_read_blas")
        return nodes, idx

    def set_device_builder(self, clustering_radius=-1):
        """device BLAS / TLAS builders: -1 (NXHIP_BUILDER_SAH, the default) = top-down binned SAH, 0 = radix tree (LBVH),
        > 0 = clustering (PLOC) with this search radius"""
        check(self.L.nxhip_set_device_builder(self.h, int(clustering_radius)), "nxhip_set_device_builder")

    def release_queues(self):
        """PathTracer::FreeDeviceBuffers: queue / path-state buffers back to the allocator until the next render"""
        check(self.L.nxhip_release_queues(self.h), "nxhip_release_queues")

    def rebuild_tlas(self, instances):
        """build the TLAS on the device (the BLAS builder over the instances' boxes) and install it; returns (nodes, instance index list)"""
        instances = np.ascontiguousarray(instances, dtype=pod.INST_DT)
        self.L.nxhip_rebuild_tlas.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        check(self.L.nxhip_rebuild_tlas(self.h, _ptr(instances), len(instances)), "nxhip_rebuild_tlas")
        self.L.nxhip_read_tlas_index.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        n = C.c_uint32(0)
        idx = np.zeros(len(instances), dtype=np.uint32)
        check(self.L.nxhip_read_tlas_index(self.h, _ptr(idx), len(idx), C.byref(n)), "nxhip_read_tlas_index")
        nodes, _ = self.read_tlas(n.value, len(instances))
        return nodes, idx

    def debug_write_blas_node(self, blas_id, node_idx, node):
        """test hook: overwrite one node of an uploaded BLAS without the upload checks"""
        node = np.ascontiguousarray(node, dtype=pod.NODE_DT).reshape(1)
        self.L.nxhip_debug_write_blas_node.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p]
        check(self.L.nxhip_debug_write_blas_node(self.h, blas_id, node_idx, _ptr(node)), "nxhip_debug_write_blas_node")

    def debug_set_scan_epoch(self, epoch):
        """test hook: passes begun since the ordered compaction's status words were last cleared (wraps at 2^20)"""
        self.L.nxhip_debug_set_scan_epoch.argtypes = [C.c_void_p, C.c_uint32]
        check(self.L.nxhip_debug_set_scan_epoch(self.h, int(epoch)), "nxhip_debug_set_scan_epoch")

    def set_instance_transforms(self, instance_ids, transforms16):
        """move existing instances on the device (inverse, bounds, traversal records, TLAS refit): no scene re-upload"""
        ids = np.ascontiguousarray(instance_ids, dtype=np.uint32)
        m = np.ascontiguousarray(transforms16, dtype=np.float32).reshape(len(ids), 16)
        check(self.L.nxhip_set_instance_transforms(self.h, _ptr(ids), _ptr(m), len(ids)), "nxhip_set_instance_transforms")

    def read_tlas(self, node_count, instance_count):
        nodes = np.zeros(node_count, dtype=pod.NODE_DT)
        insts = np.zeros(instance_count, dtype=pod.INST_DT)
        check(self.L.nxhip_read_tlas(self.h, _ptr(nodes), node_count, _ptr(insts), instance_count), "nxhip_read_tlas")
        return nodes, insts

    # ---- native multi-GPU tile split (RCCL inside the library; bench.py's N > 1 path uses torch.distributed instead)
    def mgpu_init(self, world, rank, unique_id, tile_rows):
        uid = np.frombuffer(bytes(unique_id), dtype=np.uint8).copy()
        assert len(uid) == 128
        check(self.L.nxhip_mgpu_init(self.h, world, rank, _ptr(uid), tile_rows), "nxhip_mgpu_init")
        n = C.c_uint32(0)
        check(self.L.nxhip_tile_pixel_map(self.width, self.height, world, rank, tile_rows, 1, None, C.byref(n)), "nxhip_tile_pixel_map")
        self.local_count = int(n.value)

    def mgpu_gather(self):
        check(self.L.nxhip_mgpu_gather(self.h), "nxhip_mgpu_gather")

    def mgpu_read_rgba8(self):
        out = np.zeros(self.width * self.height, dtype=np.uint32)
        check(self.L.nxhip_mgpu_read_rgba8(self.h, _ptr(out)), "nxhip_mgpu_read_rgba8")
        return out

    def mgpu_read_accumulation(self):
        out = np.zeros((self.width * self.height, 3), dtype=np.float32)
        check(self.L.nxhip_mgpu_read_accumulation(self.h, _ptr(out)), "nxhip_mgpu_read_accumulation")
        return out

    def mgpu_shutdown(self):
        check(self.L.nxhip_mgpu_shutdown(self.h), "nxhip_mgpu_shutdown")

    def resize(self, width, height):
        check(self.L.nxhip_resize(self.h, width, height), "nxhip_resize")
        self.width, self.height = int(width), int(height)
        self.local_count = self.width * self.height

    # rendering
    def reset_frame_number(self):
        check(self.L.nxhip_reset_frame_number(self.h), "nxhip_reset_frame_number")

    def set_frame_number(self, f):
        check(self.L.nxhip_set_frame_number(self.h, f), "nxhip_set_frame_number")

    def frame_number(self):
        return int(self.L.nxhip_frame_number(self.h))

    def render_frame(self):
        check(self.L.nxhip_render_frame(self.h), "nxhip_render_frame")

    def accumulate(self):
        check(self.L.nxhip_accumulate(self.h), "nxhip_accumulate")

    def accumulate_external(self, dev_ptr, count, first_frame, pixel_map_dev_ptr=None, slices=1, slice_stride=None):
        check(self.L.nxhip_accumulate_external(self.h, C.c_void_p(dev_ptr), count, slices, slice_stride if slice_stride is not None else count,
                                               first_frame, C.c_void_p(pixel_map_dev_ptr) if pixel_map_dev_ptr else None), "nxhip_accumulate_external")

    def compose_tiles(self, src_accum_dev_ptr, count, pixel_map_dev_ptr, dst_accum_dev_ptr, dst_rgba8_dev_ptr=None):
        check(self.L.nxhip_compose_tiles(self.h, C.c_void_p(src_accum_dev_ptr), count, C.c_void_p(pixel_map_dev_ptr) if pixel_map_dev_ptr else None,
                                         C.c_void_p(dst_accum_dev_ptr), C.c_void_p(dst_rgba8_dev_ptr) if dst_rgba8_dev_ptr else None), "nxhip_compose_tiles")

    def set_env_sampling(self, on=True):
        self.L.nxhip_set_env_sampling.argtypes = [C.c_void_p, C.c_int]
        check(self.L.nxhip_set_env_sampling(self.h, 1 if on else 0), "nxhip_set_env_sampling")

    def set_entry_points(self, on=True):
        """primary rays start from the state their run's first node steps provably share (include/nexus_hip.h)"""
        self.L.nxhip_set_entry_points.argtypes = [C.c_void_p, C.c_int]
        check(self.L.nxhip_set_entry_points(self.h, 1 if on else 0), "nxhip_set_entry_points")

    def debug_thin_counts_of_pass(self, bounce):
        """rays the trace launches of level `bounce` of the last pass handed to the thin kernel: (closest-hit, any-hit)"""
        c = (C.c_int32 * 2)()
        self.L.nxhip_debug_thin_counts_of_pass.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        check(self.L.nxhip_debug_thin_counts_of_pass(self.h, bounce, c), "nxhip_debug_thin_counts_of_pass")
        return int(c[0]), int(c[1])

    def debug_set_requeue(self, on=True):
        """test hook: the trace kernels hand the same rays out again and again (include/nexus_hip.h)"""
        self.L.nxhip_debug_set_requeue.argtypes = [C.c_void_p, C.c_int]
        check(self.L.nxhip_debug_set_requeue(self.h, 1 if on else 0), "nxhip_debug_set_requeue")

    def sync_timeout(self, timeout_ms):
        """nxhip_sync with a wall-clock limit; raises NexusError (code NXHIP_ERR_TIMEOUT = 6, the context is dead afterwards) when it expires"""
        self.L.nxhip_sync_timeout.argtypes = [C.c_void_p, C.c_uint32]
        check(self.L.nxhip_sync_timeout(self.h, int(timeout_ms)), "nxhip_sync_timeout")

    def debug_set_thin(self, lanes=16, iters=16, in_hooks=False, any_time=False):
        """the thin kernel's hand-over rule, and whether the ray-batch hooks use it too (a test hook: include/nexus_hip.h);
        any_time: hand over after `iters` iterations of every stretch between two refill points, dry queue or not"""
        self.L.nxhip_debug_set_thin.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        check(self.L.nxhip_debug_set_thin(self.h, lanes, iters, (1 if in_hooks else 0) | (2 if any_time else 0)), "nxhip_debug_set_thin")

    def debug_set_thin_pool(self, slots=0):
        """how many items a thin wave's pool may hold before a round puts items back (0: the product's limit; a test hook)"""
        self.L.nxhip_debug_set_thin_pool.argtypes = [C.c_void_p, C.c_uint32]
        check(self.L.nxhip_debug_set_thin_pool(self.h, int(slots)), "nxhip_debug_set_thin_pool")

    def debug_thin_counts(self):
        """rays the last ray-batch hook call handed to the thin kernel: (closest-hit, any-hit)"""
        out = (C.c_int32 * 2)()
        self.L.nxhip_debug_thin_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        check(self.L.nxhip_debug_thin_counts(self.h, out), "nxhip_debug_thin_counts")
        return int(out[0]), int(out[1])

    def read_entry_states(self):
        """(runs, 20) int32: the entry states of the last pass; column 19 = node steps saved, 16 = stack entries, 18 = instance record"""
        self.L.nxhip_read_entry_states.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        n = C.c_uint32(0)
        check(self.L.nxhip_read_entry_states(self.h, None, 0, C.byref(n)), "nxhip_read_entry_states")
        out = np.zeros((n.value, 20), np.int32)
        if n.value:
            check(self.L.nxhip_read_entry_states(self.h, _ptr(out), n.value, C.byref(n)), "nxhip_read_entry_states")
        return out

    TAIL_AUTO = 0xFFFFFFFF

    def set_tail_bounce(self, bounce):
        """0 = off, 2 .. pathLength = from that bounce on, Context.TAIL_AUTO = the default rule (include/nexus_hip.h)"""
        self.L.nxhip_set_tail_bounce.argtypes = [C.c_void_p, C.c_uint32]
        check(self.L.nxhip_set_tail_bounce(self.h, bounce), "nxhip_set_tail_bounce")

    def set_passes_in_flight(self, passes):
        self.L.nxhip_set_passes_in_flight.argtypes = [C.c_void_p, C.c_uint32]
        check(self.L.nxhip_set_passes_in_flight(self.h, passes), "nxhip_set_passes_in_flight")

    def set_frames_per_pass(self, frames):
        check(self.L.nxhip_set_frames_per_pass(self.h, frames), "nxhip_set_frames_per_pass")
        self.frames_per_pass = int(frames)

    def write_accumulation(self, accumulation, frame_number):
        a = np.ascontiguousarray(accumulation, dtype=np.float32).reshape(-1, 3)
        assert len(a) == self.local_count
        check(self.L.nxhip_write_accumulation(self.h, _ptr(a), int(frame_number)), "nxhip_write_accumulation")

    def bind_radiance(self, dev_ptr, capacity):
        check(self.L.nxhip_bind_radiance(self.h, C.c_void_p(dev_ptr) if dev_ptr else None, capacity), "nxhip_bind_radiance")

    def read_full_accumulation(self):
        out = np.zeros((self.width * self.height, 3), np.float32)
        check(self.L.nxhip_read_full_accumulation(self.h, _ptr(out)), "nxhip_read_full_accumulation")
        return out

    def read_full_rgba8(self):
        out = np.zeros(self.width * self.height, np.uint32)
        check(self.L.nxhip_read_full_rgba8(self.h, _ptr(out)), "nxhip_read_full_rgba8")
        return out

    def render(self, frames):
        check(self.L.nxhip_render(self.h, frames), "nxhip_render")

    def sync(self):
        check(self.L.nxhip_sync(self.h), "nxhip_sync")

    def read_radiance(self):
        out = np.zeros((self.local_count * self.frames_per_pass, 3), np.float32)
        check(self.L.nxhip_read_radiance(self.h, _ptr(out)), "nxhip_read_radiance")
        return out

    def read_accumulation(self):
        out = np.zeros((self.local_count, 3), np.float32)
        check(self.L.nxhip_read_accumulation(self.h, _ptr(out)), "nxhip_read_accumulation")
        return out

    def read_rgba8(self):
        out = np.zeros(self.local_count, np.uint32)
        check(self.L.nxhip_read_rgba8(self.h, _ptr(out)), "nxhip_read_rgba8")
        return out

    def radiance_device_ptr(self):
        return self.L.nxhip_radiance_device_ptr(self.h)

    def accumulation_device_ptr(self):
        return self.L.nxhip_accumulation_device_ptr(self.h)

    def read_queue_sizes(self):
        q = QueueSizes()
        check(self.L.nxhip_read_queue_sizes(self.h, C.byref(q)), "nxhip_read_queue_sizes")
        return {n: np.array(getattr(q, n)[:], dtype=np.int32) for n, _ in QueueSizes._fields_}

    def set_pixel_query(self, x, y):
        check(self.L.nxhip_set_pixel_query(self.h, x, y), "nxhip_set_pixel_query")

    def get_selected_instance(self):
        v = C.c_int32(0)
        check(self.L.nxhip_get_selected_instance(self.h, C.byref(v)), "nxhip_get_selected_instance")
        return v.value

    # hooks
    def trace_batch(self, rays):
        rays = np.ascontiguousarray(rays, dtype=pod.RAY_DT)
        hits = np.zeros(len(rays), dtype=pod.HIT_DT)
        check(self.L.nxhip_trace_batch(self.h, _ptr(rays), len(rays), _ptr(hits)), "nxhip_trace_batch")
        return hits

    def trace_shadow_batch(self, rays, tmax):
        rays = np.ascontiguousarray(rays, dtype=pod.RAY_DT)
        tmax = np.ascontiguousarray(tmax, dtype=np.float32)
        occ = np.zeros(len(rays), dtype=np.uint8)
        check(self.L.nxhip_trace_shadow_batch(self.h, _ptr(rays), _ptr(tmax), len(rays), _ptr(occ)), "nxhip_trace_shadow_batch")
        return occ

    def _bsdf_batch(self, fn, name, material, queries):
        mat = np.ascontiguousarray(material, dtype=pod.MAT_DT).reshape(1)
        q = np.ascontiguousarray(queries, dtype=pod.BSDF_QUERY_DT)
        out = np.zeros(len(q), dtype=pod.BSDF_RESULT_DT)
        check(fn(self.h, _ptr(mat), _ptr(q), len(q), _ptr(out)), name)
        return out

    def bsdf_sample_batch(self, material, queries):
        return self._bsdf_batch(self.L.nxhip_bsdf_sample_batch, "nxhip_bsdf_sample_batch", material, queries)

    def bsdf_eval_batch(self, material, queries):
        return self._bsdf_batch(self.L.nxhip_bsdf_eval_batch, "nxhip_bsdf_eval_batch", material, queries)

    def tex2d_batch(self, kind, texture_id, uv):
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((len(uv), 4), dtype=np.float32)
        check(self.L.nxhip_tex2d_batch(self.h, {"diffuse": 0, "emissive": 1, "hdr": 2}[kind], int(texture_id), _ptr(uv), len(uv), _ptr(out)), "nxhip_tex2d_batch")
        return out

    def fmath_batch(self, op, a, b=None):
        """include/nexus_fmath.h on the device: out[i] = nxf_apply(op, a[i], b[i]) (op: pod.NXF_OP_*)"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        bb = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
        out = np.zeros(len(a), dtype=np.float64)
        check(self.L.nxhip_fmath_batch(self.h, int(op), _ptr(a), None if bb is None else _ptr(bb), len(a), _ptr(out)), "nxhip_fmath_batch")
        return out

    def enable_trace_stats(self, on=True):
        check(self.L.nxhip_enable_trace_stats(self.h, 1 if on else 0), "nxhip_enable_trace_stats")

    def read_trace_stats(self, reset=False):
        a, b = TraceStats(), TraceStats()
        check(self.L.nxhip_read_trace_stats(self.h, C.byref(a), C.byref(b), 1 if reset else 0), "nxhip_read_trace_stats")
        return a.as_dict(), b.as_dict()

    def enable_kernel_timing(self, on=True, in_graph=False, last_replay_only=False):
        """on: hipEvent pair per kernel launch; in_graph: keep the hipGraph (and its trace || shadow overlap) and time with
        event-record nodes inside it, otherwise launch kernel by kernel; last_replay_only (with in_graph): no sync between
        replays, read_kernel_times returns the last replay of the series"""
        mode = 0 if not on else (3 if (in_graph and last_replay_only) else 2 if in_graph else 1)
        check(self.L.nxhip_enable_kernel_timing(self.h, mode), "nxhip_enable_kernel_timing")

    def read_graph_timeline(self):
        """the kernels of the last timed replay (enable_kernel_timing(in_graph=True)) in graph order:
        [(class name, start ms since the replay's first kernel, duration ms), ...]"""
        n = C.c_uint32(0)
        check(self.L.nxhip_read_graph_timeline(self.h, None, None, None, 0, C.byref(n)), "nxhip_read_graph_timeline")
        k = np.zeros(n.value, np.int32)
        s = np.zeros(n.value, np.float32)
        d = np.zeros(n.value, np.float32)
        if n.value:
            check(self.L.nxhip_read_graph_timeline(self.h, _ptr(k), _ptr(s), _ptr(d), n.value, C.byref(n)), "nxhip_read_graph_timeline")
        return [(KERNEL_CLASSES[int(k[i])], float(s[i]), float(d[i])) for i in range(n.value)]

    def read_kernel_times(self, reset=False):
        t = KernelTimes()
        check(self.L.nxhip_read_kernel_times(self.h, C.byref(t), 1 if reset else 0), "nxhip_read_kernel_times")
        return {k: {"ms": t.ms[i], "launches": int(t.launches[i])} for i, k in enumerate(KERNEL_CLASSES)}


def tile_pixel_map(width, height, world, rank, tile_rows, tiled=True):
    """the library's own tile split (nxhip_tile_pixel_map); no GPU needed"""
    n = C.c_uint32(0)
    check(lib().nxhip_tile_pixel_map(width, height, world, rank, tile_rows, 1 if tiled else 0, None, C.byref(n)), "nxhip_tile_pixel_map")
    out = np.zeros(n.value, dtype=np.uint32)
    check(lib().nxhip_tile_pixel_map(width, height, world, rank, tile_rows, 1 if tiled else 0, _ptr(out), C.byref(n)), "nxhip_tile_pixel_map")
    return out


def mgpu_unique_id():
    uid = np.zeros(128, dtype=np.uint8)
    check(lib().nxhip_mgpu_unique_id(_ptr(uid)), "nxhip_mgpu_unique_id")
    return uid.tobytes()


def device_count():
    return int(lib().nxhip_device_count())


# ---- C++ Scene / PathTracer facade (include/nexus/Scene.h, PathTracer.h) -------------------------------

def _scheck(rc, what):
    if rc != 0:
        msg = lib().nxs_last_error()
        raise NexusError(f"{what} failed: {msg.decode() if msg else ''}")


class Scene:
    """nexus::Scene + its AssetManager, driven with the reference's call sequence."""

    def __init__(self, width, height):
        self.L = lib()
        h = C.c_void_p()
        _scheck(self.L.nxs_scene_create(width, height, C.byref(h)), "nxs_scene_create")
        self.h = h
        self.width, self.height = width, height

    def close(self):
        if getattr(self, "h", None):
            self.L.nxs_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_material(self, mat):
        m = np.ascontiguousarray(mat, dtype=pod.MAT_DT).reshape(1)
        i = C.c_int32(-1)
        _scheck(self.L.nxs_scene_add_material(self.h, _ptr(m), C.byref(i)), "nxs_scene_add_material")
        return i.value

    def add_texture(self, kind, rgba8):
        img = np.ascontiguousarray(rgba8, dtype=np.uint8)
        i = C.c_int32(-1)
        _scheck(self.L.nxs_scene_add_texture(self.h, {"diffuse": 0, "emissive": 1}[kind], _ptr(img), img.shape[1], img.shape[0], C.byref(i)), "nxs_scene_add_texture")
        return i.value

    def set_hdr_map(self, rgba8):
        img = np.ascontiguousarray(rgba8, dtype=np.uint8)
        _scheck(self.L.nxs_scene_set_hdr_map(self.h, _ptr(img), img.shape[1], img.shape[0]), "nxs_scene_set_hdr_map")

    def add_mesh(self, tris, material_id=-1):
        t = np.ascontiguousarray(tris, dtype=pod.TRI_DT)
        i = C.c_int32(-1)
        _scheck(self.L.nxs_scene_add_mesh(self.h, _ptr(t), len(t), material_id, C.byref(i)), "nxs_scene_add_mesh")
        return i.value

    def set_instance_transform(self, instance_id, position, rotation_deg, scale):
        p, r, sc = (np.asarray(x, np.float32) for x in (position, rotation_deg, scale))
        _scheck(self.L.nxs_scene_set_instance_transform(self.h, instance_id, _ptr(p), _ptr(r), _ptr(sc)), "nxs_scene_set_instance_transform")

    def assign_material(self, instance_id, material_id):
        self.L.nxs_scene_assign_material.argtypes = [C.c_void_p, C.c_uint32, C.c_int32]
        _scheck(self.L.nxs_scene_assign_material(self.h, instance_id, material_id), "nxs_scene_assign_material")

    def set_tlas_refit(self, enable=True):
        _scheck(self.L.nxs_scene_set_tlas_refit(self.h, 1 if enable else 0), "nxs_scene_set_tlas_refit")

    def set_device_tlas(self, enable=True):
        """the TLAS is built (nxhip_rebuild_tlas) and refitted on the device; update() runs no host TLAS builder"""
        self.L.nxs_scene_set_device_tlas.argtypes = [C.c_void_p, C.c_int]
        _scheck(self.L.nxs_scene_set_device_tlas(self.h, 1 if enable else 0), "nxs_scene_set_device_tlas")

    def load_file(self, path, file_name):
        """Scene::CreateMeshInstanceFromFile: materials, meshes (one BVH8 each) and instances of a .glb / .obj"""
        _scheck(self.L.nxs_scene_load_file(self.h, str(path).encode(), str(file_name).encode()), "nxs_scene_load_file")

    def create_instance(self, mesh_id, material_id, position=(0, 0, 0), rotation_deg=(0, 0, 0), scale=(1, 1, 1)):
        p, r, s = (np.asarray(x, np.float32) for x in (position, rotation_deg, scale))
        i = C.c_int32(-1)
        _scheck(self.L.nxs_scene_create_instance(self.h, mesh_id, material_id, _ptr(p), _ptr(r), _ptr(s), C.byref(i)), "nxs_scene_create_instance")
        return i.value

    def set_camera(self, position, forward, hfov_deg, focus_dist=5.0, defocus_deg=0.0):
        p, f = np.asarray(position, np.float32), np.asarray(forward, np.float32)
        _scheck(self.L.nxs_scene_set_camera(self.h, _ptr(p), _ptr(f), hfov_deg, focus_dist, defocus_deg), "nxs_scene_set_camera")

    def set_render_settings(self, st):
        st = np.ascontiguousarray(st, dtype=pod.SETTINGS_DT).reshape(1)
        _scheck(self.L.nxs_scene_set_render_settings(self.h, _ptr(st)), "nxs_scene_set_render_settings")

    def update(self):
        _scheck(self.L.nxs_scene_update(self.h), "nxs_scene_update")

    def light_count(self):
        return int(self.L.nxs_scene_light_count(self.h))

    def instance_count(self):
        return int(self.L.nxs_scene_instance_count(self.h))


def write_png(path, rgba8, width, height, flip=True):
    px = np.ascontiguousarray(rgba8, dtype=np.uint32)
    L = lib()
    L.nxh_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
    _scheck(L.nxh_write_png(str(path).encode(), _ptr(px), width, height, 1 if flip else 0), "nxh_write_png")


def write_exr(path, rgb, width, height, flip=True):
    px = np.ascontiguousarray(rgb, dtype=np.float32)
    L = lib()
    L.nxh_write_exr.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
    _scheck(L.nxh_write_exr(str(path).encode(), _ptr(px), width, height, 1 if flip else 0), "nxh_write_exr")


class Renderer:
    """nexus::Renderer: the reference's frame driver (Renderer::Render / SaveScreenshot) without its window."""

    def __init__(self, width, height, scene, device=0):
        self.L = lib()
        L = self.L
        L.nxs_renderer_create.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.nxs_renderer_destroy.argtypes = [C.c_void_p]
        L.nxs_renderer_render.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        L.nxs_renderer_reset.argtypes = [C.c_void_p]
        L.nxs_renderer_on_resize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.nxs_renderer_save_screenshot.argtypes = [C.c_void_p, C.c_char_p]
        L.nxs_renderer_save_exr.argtypes = [C.c_void_p, C.c_char_p]
        L.nxs_renderer_frame_number.argtypes = [C.c_void_p]
        L.nxs_renderer_frame_number.restype = C.c_uint32
        L.nxs_renderer_megasamples_per_second.argtypes = [C.c_void_p]
        L.nxs_renderer_megasamples_per_second.restype = C.c_double
        L.nxs_renderer_device_context.argtypes = [C.c_void_p]
        L.nxs_renderer_device_context.restype = C.c_void_p
        L.nxs_renderer_set_modes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        h = C.c_void_p()
        _scheck(L.nxs_renderer_create(width, height, scene.h, device, C.byref(h)), "nxs_renderer_create")
        self.h = h
        self.width, self.height = width, height

    def close(self):
        if getattr(self, "h", None):
            self.L.nxs_renderer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_modes(self, rng_mode, compact_mode, conductor_mode):
        _scheck(self.L.nxs_renderer_set_modes(self.h, rng_mode, compact_mode, conductor_mode), "nxs_renderer_set_modes")

    def render(self, scene, delta_time=0.0):
        _scheck(self.L.nxs_renderer_render(self.h, scene.h, delta_time), "nxs_renderer_render")

    def reset(self):
        _scheck(self.L.nxs_renderer_reset(self.h), "nxs_renderer_reset")

    def save_screenshot(self, path):
        _scheck(self.L.nxs_renderer_save_screenshot(self.h, str(path).encode()), "nxs_renderer_save_screenshot")

    def save_exr(self, path):
        _scheck(self.L.nxs_renderer_save_exr(self.h, str(path).encode()), "nxs_renderer_save_exr")

    def frame_number(self):
        return int(self.L.nxs_renderer_frame_number(self.h))

    def megasamples_per_second(self):
        return float(self.L.nxs_renderer_megasamples_per_second(self.h))

    def read_accumulation(self):
        out = np.zeros((self.width * self.height, 3), dtype=np.float32)
        lib().nxhip_read_accumulation.argtypes = [C.c_void_p, C.c_void_p]
        check(lib().nxhip_read_accumulation(C.c_void_p(self.L.nxs_renderer_device_context(self.h)), _ptr(out)), "nxhip_read_accumulation")
        return out

    def read_pixels(self):
        out = np.zeros(self.width * self.height, dtype=np.uint32)
        check(lib().nxhip_read_rgba8(C.c_void_p(self.L.nxs_renderer_device_context(self.h)), _ptr(out)), "nxhip_read_rgba8")
        return out


class PathTracer:
    """nexus::PathTracer: Render(scene) = one frame through the device layer + accumulate."""

    def __init__(self, width, height, device=0):
        self.L = lib()
        h = C.c_void_p()
        _scheck(self.L.nxs_pathtracer_create(width, height, device, C.byref(h)), "nxs_pathtracer_create")
        self.h = h
        self.width, self.height = width, height

    def close(self):
        if getattr(self, "h", None):
            self.L.nxs_pathtracer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_modes(self, rng_mode, compact_mode, conductor_mode):
        _scheck(self.L.nxs_pathtracer_set_modes(self.h, rng_mode, compact_mode, conductor_mode), "nxs_pathtracer_set_modes")

    def set_frames_per_pass(self, frames):
        _scheck(self.L.nxs_pathtracer_set_frames_per_pass(self.h, frames), "nxs_pathtracer_set_frames_per_pass")

    def set_passes_in_flight(self, passes):
        _scheck(self.L.nxs_pathtracer_set_passes_in_flight(self.h, passes), "nxs_pathtracer_set_passes_in_flight")

    def set_pixel_order(self, order):
        self.L.nxs_pathtracer_set_pixel_order.argtypes = [C.c_void_p, C.c_int]
        _scheck(self.L.nxs_pathtracer_set_pixel_order(self.h, int(order)), "nxs_pathtracer_set_pixel_order")

    def set_entry_points(self, on=True):
        self.L.nxs_pathtracer_set_entry_points.argtypes = [C.c_void_p, C.c_int]
        _scheck(self.L.nxs_pathtracer_set_entry_points(self.h, 1 if on else 0), "nxs_pathtracer_set_entry_points")

    def set_device_blas_build(self, scene, enable=True):
        """PathTracer::SetDeviceBlasBuild: meshes added to `scene` from now on are built into BVH8s on the GPU"""
        self.L.nxs_pathtracer_set_device_blas_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _scheck(self.L.nxs_pathtracer_set_device_blas_build(self.h, scene.h, 1 if enable else 0), "nxs_pathtracer_set_device_blas_build")

    def update_device_scene(self, scene):
        _scheck(self.L.nxs_pathtracer_update_device_scene(self.h, scene.h), "nxs_pathtracer_update_device_scene")

    def render(self, scene):
        _scheck(self.L.nxs_pathtracer_render(self.h, scene.h), "nxs_pathtracer_render")

    def reset_frame_number(self):
        _scheck(self.L.nxs_pathtracer_reset_frame_number(self.h), "nxs_pathtracer_reset_frame_number")

    def frame_number(self):
        return int(self.L.nxs_pathtracer_frame_number(self.h))

    def read_pixels(self):
        out = np.zeros(self.width * self.height, np.uint32)
        _scheck(self.L.nxs_pathtracer_read_pixels(self.h, _ptr(out)), "nxs_pathtracer_read_pixels")
        return out

    def read_radiance(self):
        out = np.zeros((self.width * self.height, 3), np.float32)
        check(self.L.nxhip_read_radiance(self.L.nxs_pathtracer_device_context(self.h), _ptr(out)), "nxhip_read_radiance")
        return out

    def read_blas(self, blas_id, tri_count):
        """nodes and primitive index list of BLAS `blas_id` of this path tracer's device context"""
        ctx = self.L.nxs_pathtracer_device_context(self.h)
        self.L.nxhip_read_blas.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        n = C.c_uint32(0)
        check(self.L.nxhip_read_blas(ctx, blas_id, None, 0, None, 0, C.byref(n)), "nxhip_read_blas")
        nodes = np.zeros(n.value, dtype=pod.NODE_DT)
        idx = np.zeros(tri_count, dtype=np.uint32)
        check(self.L.nxhip_read_blas(ctx, blas_id, _ptr(nodes), n.value, _ptr(idx), tri_count, C.byref(n)), "nxhip_read_blas")
        return nodes, idx
