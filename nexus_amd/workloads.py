"""Scene assembly and the BASELINE.json configurations as seeded procedural workloads.

`Workload` builds BLAS / TLAS with the product's host builders (nexus_amd.capi) and uploads them to a device context;
bench.py renders them, and the tests hand the very same bytes to their CPU checker (tests.scene_helpers.BuiltScene adds
that side; nothing here depends on it)."""
import numpy as np

from . import capi, pod, scenegen

IDENTITY = np.eye(4, dtype=np.float32).reshape(16)


def make_settings(use_mis=True, path_length=4, background=(1, 1, 1), background_intensity=0.0):
    """RenderSettings defaults of the reference: Renderer/RenderSettings.h:4-10"""
    s = np.zeros((), dtype=pod.SETTINGS_DT)
    s["useMIS"] = 1 if use_mis else 0
    s["pathLength"] = path_length
    s["backgroundColor"] = background
    s["backgroundIntensity"] = background_intensity
    return s


def mesh_lights(instances, materials):
    """Scene::UpdateInstanceLighting (/root/reference/Nexus/src/Scene/Scene.cpp:142-176): an instance is a light iff its
    material has an emissive map or intensity * max(emissive) > 0; meshId = index of the instance."""
    out = []
    for i, inst in enumerate(instances):
        m = materials[inst["materialId"]]
        if m["emissiveMapId"] != -1 or float(m["intensity"]) * float(np.max(m["emissive"])) > 0.0:
            l = np.zeros((), dtype=pod.LIGHT_DT)
            l["meshId"] = i
            l["type"] = pod.LIGHT_MESH
            out.append(l)
    return np.array(out, dtype=pod.LIGHT_DT) if out else np.zeros(0, pod.LIGHT_DT)


def checker_texture(w=64, h=32, seed=0, alpha=False):
    rng = np.random.RandomState(seed)
    img = rng.randint(0, 256, size=(h, w, 4)).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    chk = ((xx // 8 + yy // 8) % 2).astype(bool)
    img[chk, :3] = img[chk, :3] // 3
    img[..., 3] = rng.randint(128, 256, size=(h, w)) if alpha else 255
    return img


class Workload:
    def __init__(self, meshes, placements, materials=None, lights=None, camera=None, settings=None, diffuse_maps=(), emissive_maps=(),
                 hdr_map=None, build_threads=4):
        """meshes: list of TRI_DT arrays; placements: list of (meshIdx, materialId, transform16); build_threads 0 = all cores."""
        self.meshes = [np.ascontiguousarray(m, dtype=pod.TRI_DT) for m in meshes]
        self.blas = []
        for m in self.meshes:
            nodes, idx = capi.bvh8_build(m, threads=build_threads)
            self.blas.append((nodes, m, idx))
        insts = []
        for mesh_idx, mat_id, xf in placements:
            insts.append(capi.instance_init(mesh_idx, mat_id, xf, self.blas[mesh_idx][0][0]))
        self.instances = np.array(insts, dtype=pod.INST_DT)
        self.tlas_nodes, self.tlas_idx = capi.tlas_build(self.instances)
        self.materials = np.ascontiguousarray(materials if materials is not None else np.array([pod.make_material()], dtype=pod.MAT_DT), dtype=pod.MAT_DT)
        self.lights = np.ascontiguousarray(lights if lights is not None else np.zeros(0, pod.LIGHT_DT), dtype=pod.LIGHT_DT)
        self.camera = camera
        self.settings = settings if settings is not None else make_settings()
        self.diffuse_maps, self.emissive_maps, self.hdr_map = list(diffuse_maps), list(emissive_maps), hdr_map
        self.env_sampling = False  # extension: importance-sample the environment map in the NEE (nxhip_set_env_sampling)

    @property
    def triangles(self):
        """triangles reachable through the TLAS (instanced meshes count once per instance)"""
        return int(sum(len(self.meshes[int(i["bvhIdx"])]) for i in self.instances))

    @property
    def unique_triangles(self):
        return int(sum(len(m) for m in self.meshes))

    @property
    def bvh8_nodes(self):
        return int(sum(len(b[0]) for b in self.blas))

    def scene_bytes(self):
        """device bytes the trace kernels read from: 80-byte nodes + 48-byte intersection records + instance records"""
        return int(sum(80 * len(b[0]) + 48 * len(b[1]) for b in self.blas) + 80 * len(self.tlas_nodes) + 80 * len(self.instances))

    def upload(self, ctx, device_bvh=False, device_tlas=False):
        """device_bvh: build every BLAS on the GPU (nxhip_build_blas: binned SAH + SAH-DP collapse) instead of uploading the host
        builder's; the instances' world bounds follow the BLAS root frame, so instances and TLAS are rebuilt for those roots."""
        ctx.clear_blas()
        ctx.clear_textures()
        if device_bvh:
            roots = []
            for _nodes, tris, _idx in self.blas:
                bid = ctx.build_blas(tris)
                roots.append(ctx.read_blas(bid, len(tris))[0][0])
            insts = np.array([capi.instance_init(int(i["bvhIdx"]), int(i["materialId"]), i["transform"], roots[int(i["bvhIdx"])]) for i in self.instances],
                             dtype=pod.INST_DT)
            if device_tlas:
                ctx.rebuild_tlas(insts)
            else:
                tlas_nodes, tlas_idx = capi.tlas_build(insts)
                ctx.set_tlas(tlas_nodes, tlas_idx, insts)
        else:
            for nodes, tris, idx in self.blas:
                ctx.upload_blas(nodes, tris, idx)
            ctx.set_tlas(self.tlas_nodes, self.tlas_idx, self.instances)
        ctx.set_materials(self.materials)
        ctx.set_lights(self.lights)
        for img in self.diffuse_maps:
            ctx.upload_texture("diffuse", img)
        for img in self.emissive_maps:
            ctx.upload_texture("emissive", img)
        if self.hdr_map is not None:
            ctx.upload_texture("hdr", self.hdr_map)
            ctx.set_env_sampling(self.env_sampling)
        if self.camera is not None:
            ctx.set_camera(self.camera)
        ctx.set_render_settings(self.settings)


def _look(eye, target, hfov, width, height):
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    return capi.camera_init(eye, fwd, hfov, width, height, 5.0, 0.0)


def config1(glb_path, width=512, height=512, path_length=4, force_diffuse=True, use_mis=True, cls=Workload):
    """BASELINE.json configs[0]: the reference's cornell_box.glb (8 primitives -> 8 BLAS / instances, node rotated +90 deg
    about X), all materials DIFFUSE with the glb base colours, light emissive (1,1,1) x 35, camera at (0,1,3.9) looking
    down -z with a 40 degree horizontal FOV, focus 5, no defocus (the file carries no camera; SURVEY.md section 8d fixes
    these numbers), pathLength 4, MIS on, black background."""
    from . import loaders

    ls = loaders.load_glb(glb_path)
    mats = ls.materials.copy()
    if force_diffuse:
        mats["type"] = pod.MAT_DIFFUSE
    placements = [(inst["mesh"], inst["material"], capi.mat4_from_trs(inst["position"], inst["rotation"], inst["scale"])) for inst in ls.instances]
    cam = capi.camera_init((0.0, 1.0, 3.9), (0.0, 0.0, -1.0), 40.0, width, height, 5.0, 0.0)
    sc = cls(ls.meshes, placements, materials=mats, camera=cam,
             settings=make_settings(use_mis=use_mis, path_length=path_length, background=(1, 1, 1), background_intensity=0.0))
    sc.lights = mesh_lights(sc.instances, sc.materials)
    return sc


def world_triangles(sc):
    """every instance's triangles in world space, concatenated: what a single-level BVH2 over the whole scene is built from
    (configs[0]'s "CPU BVH2 intersect reference path")"""
    out = []
    for inst in sc.instances:
        tris = sc.blas[int(inst["bvhIdx"])][1].copy()
        M = inst["transform"].reshape(4, 4).astype(np.float64)
        for k in ("pos0", "pos1", "pos2"):
            tris[k] = (tris[k].astype(np.float64) @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
        out.append(tris)
    return np.concatenate(out)


def pixel_centre_rays(camera, width, height):
    """the camera's rays through the pixel centres (no lens, no jitter), in row-major pixel order"""
    jj, ii = np.mgrid[0:height, 0:width]
    x = ((ii + 0.5) / width).reshape(-1, 1)
    y = ((jj + 0.5) / height).reshape(-1, 1)
    target = camera["lowerLeftCorner"].astype(np.float64) + camera["viewportX"].astype(np.float64) * x + camera["viewportY"].astype(np.float64) * y
    d = target - camera["position"].astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(width * height, dtype=pod.RAY_DT)
    rays["origin"] = camera["position"]
    rays["direction"] = d.astype(np.float32)
    return rays


def config2(width=1920, height=1080, nu=1024, nv=512, path_length=8, cls=Workload):
    """configs[1]: seeded displaced torus (2*nu*nv triangles) resting on a 2-triangle floor under a 2-triangle emissive
    quad; mesh = CONDUCTOR (ior (0.2,0.9,1.1), k (3.9,2.4,2.2), roughness 0.3), floor = DIFFUSE 0.7, light intensity 20."""
    torus = scenegen.displaced_torus(nu, nv, seed=1, major=1.0, minor=0.45, amp=0.06, center=(0.0, 0.56, 0.0))
    floor = scenegen.quad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6))
    light = scenegen.quad((-1.2, 4.0, -1.2), (1.2, 4.0, -1.2), (1.2, 4.0, 1.2), (-1.2, 4.0, 1.2))
    mats = np.array([
        pod.make_material(pod.MAT_CONDUCTOR, roughness=0.3, conductor_ior=(0.2, 0.9, 1.1), conductor_k=(3.9, 2.4, 2.2)),
        pod.make_material(pod.MAT_DIFFUSE, albedo=(0.7, 0.7, 0.7)),
        pod.make_material(pod.MAT_DIFFUSE, albedo=(0.8, 0.8, 0.8), emissive=(1.0, 1.0, 1.0), intensity=20.0),
    ], dtype=pod.MAT_DT)
    sc = cls([torus, floor, light], [(i, i, IDENTITY) for i in range(3)], materials=mats,
             camera=_look((0.0, 3.3, 4.9), (0.0, 0.35, 0.0), 52.0, width, height),
             settings=make_settings(use_mis=True, path_length=path_length, background=(1, 1, 1), background_intensity=0.0), build_threads=0)
    sc.lights = mesh_lights(sc.instances, sc.materials)
    return sc


def procedural_sky(width=2048, height=1024):
    """Equirectangular RGBA8 sky: vertical gradient, a bright sun disc and seeded bands (config 4's environment)."""
    v = np.linspace(0.0, 1.0, height, dtype=np.float32)[:, None]
    u = np.linspace(0.0, 1.0, width, dtype=np.float32)[None, :]
    top = np.array([0.25, 0.45, 0.9], np.float32)
    hor = np.array([0.9, 0.85, 0.8], np.float32)
    gnd = np.array([0.25, 0.22, 0.2], np.float32)
    t = np.clip(v * 2.0, 0.0, 1.0)[..., None]
    b = np.clip(v * 2.0 - 1.0, 0.0, 1.0)[..., None]
    img = (top * (1 - t) + hor * t) * (1 - b) + gnd * b
    img = np.broadcast_to(img, (height, width, 3)).copy()
    sun = np.exp(-(((u - 0.3) * 2.0) ** 2 + ((v - 0.25) * 1.0) ** 2) * 400.0)[..., None]
    img = np.clip(img + sun * np.array([1.0, 0.95, 0.8], np.float32), 0.0, 1.0)
    img *= (0.9 + 0.1 * np.sin(u * 40.0))[..., None]
    out = np.zeros((height, width, 4), np.uint8)
    out[..., :3] = (img * 255.0 + 0.5).astype(np.uint8)
    out[..., 3] = 255
    return out


def split_by_octants(mesh, levels):
    """PROBE (instance opening): the mesh cut into 8^levels parts by the octant of each triangle's centroid in the
    bounding box of its part — what a TLAS whose leaves enter an instance's BLAS below its root would see, built from existing pieces
    (every part a BLAS of its own, every placement repeated per part)."""
    parts = [mesh]
    for _ in range(levels):
        nxt = []
        for m in parts:
            c = (m["pos0"].astype(np.float64) + m["pos1"] + m["pos2"]) / 3.0
            mid = 0.5 * (c.min(0) + c.max(0))
            key = (c[:, 0] > mid[0]) * 4 + (c[:, 1] > mid[1]) * 2 + (c[:, 2] > mid[2]) * 1
            nxt += [m[key == k] for k in range(8) if np.any(key == k)]
        parts = nxt
    return parts


def config4(width=1920, height=1080, path_length=8, n_side=10, nu=250, nv=200, cls=Workload, split_levels=0):
    """configs[3]: one 2*nu*nv-triangle BLAS (seed 2) instanced n_side^3 times on a jittered lattice with random rotations
    and scales (seed 3), DIELECTRIC roughness 0.2 ior 1.45, procedural 2048x1024 equirectangular environment.  The
    reference adds the environment on a miss only (PathTracer.cu:152-164); no environment NEE."""
    mesh = scenegen.displaced_torus(nu, nv, seed=2, major=0.5, minor=0.2, amp=0.03)
    rng = np.random.RandomState(3)
    placements = []
    for ix in range(n_side):
        for iy in range(n_side):
            for iz in range(n_side):
                pos = (np.array([ix, iy, iz], np.float64) - (n_side - 1) / 2.0) * 1.6 + rng.uniform(-0.3, 0.3, 3)
                placements.append((0, 0, capi.mat4_from_trs(pos, rng.uniform(0, 360, 3), rng.uniform(0.6, 1.3, 3))))
    mats = np.array([pod.make_material(pod.MAT_DIELECTRIC, albedo=(0.95, 0.97, 1.0), roughness=0.2, ior=1.45)], dtype=pod.MAT_DT)
    ext = n_side * 1.6
    meshes = [mesh]
    if split_levels:
        meshes = split_by_octants(mesh, split_levels)
        placements = [(k, 0, xf) for (_, _, xf) in placements for k in range(len(meshes))]
    return cls(meshes, placements, materials=mats, camera=_look((ext * 0.9, ext * 0.55, ext * 1.25), (0, 0, 0), 45.0, width, height),
               settings=make_settings(use_mis=True, path_length=path_length, background=(1, 1, 1), background_intensity=1.0),
               hdr_map=procedural_sky(), build_threads=0)


def config5(width=3840, height=2160, path_length=16, field=2200, prop_nu=512, prop_nv=256, n_props=16, cls=Workload):
    """configs[4]: ~10 M triangles — a 2*field^2-triangle displaced room shell (floor / back wall / ceiling from one
    height-field BLAS, rotated) plus instanced props, all four material types, emissive-textured light panels,
    3840x2160, pathLength 16.  About 0.6 GB of nodes + intersection records: larger than the 256 MiB Infinity Cache, so
    this is the configuration whose traversal streams from HBM."""
    shell = scenegen.height_field(field, seed=5, amp=0.08)
    prop = scenegen.displaced_torus(prop_nu, prop_nv, seed=6, major=0.5, minor=0.2, amp=0.04)
    panel = scenegen.quad((-0.8, 0, -0.8), (0.8, 0, -0.8), (0.8, 0, 0.8), (-0.8, 0, 0.8))
    mats = np.array([
        pod.make_material(pod.MAT_DIFFUSE, albedo=(0.75, 0.72, 0.7), diffuse_map=0),
        pod.make_material(pod.MAT_PLASTIC, albedo=(0.8, 0.3, 0.2), roughness=0.35, ior=1.5),
        pod.make_material(pod.MAT_DIELECTRIC, albedo=(0.95, 0.97, 1.0), roughness=0.15, ior=1.45),
        pod.make_material(pod.MAT_CONDUCTOR, roughness=0.25),
        pod.make_material(pod.MAT_DIFFUSE, albedo=(0.8, 0.8, 0.8), emissive=(1.0, 0.92, 0.85), intensity=18.0, emissive_map=0),
    ], dtype=pod.MAT_DT)
    placements = [
        (0, 0, capi.mat4_from_trs((0, 0, 0), (0, 0, 0), (6, 1, 6))),            # floor
        (0, 0, capi.mat4_from_trs((0, 3, -6), (90, 0, 0), (6, 1, 3))),          # back wall
        (0, 0, capi.mat4_from_trs((0, 6, 0), (180, 0, 0), (6, 1, 6))),          # ceiling
    ]
    rng = np.random.RandomState(7)
    for k in range(n_props):
        pos = (rng.uniform(-4.5, 4.5), rng.uniform(0.5, 2.5), rng.uniform(-4.5, 3.0))
        placements.append((1, 1 + k % 3, capi.mat4_from_trs(pos, rng.uniform(0, 360, 3), rng.uniform(0.7, 1.4, 3))))
    for x in (-3.0, 0.0, 3.0):
        placements.append((2, 4, capi.mat4_from_trs((x, 5.6, -1.0), (180, 0, 0))))
    sc = cls([shell, prop, panel], placements, materials=mats, camera=_look((0.0, 2.6, 9.5), (0.0, 1.8, 0.0), 55.0, width, height),
             settings=make_settings(use_mis=True, path_length=path_length, background=(0.5, 0.6, 0.8), background_intensity=0.3),
             diffuse_maps=[checker_texture(256, 256, 11)], emissive_maps=[checker_texture(64, 64, 12)], build_threads=0)
    sc.lights = mesh_lights(sc.instances, sc.materials)
    return sc


def check_obj_round_trip(sc, mesh_index=0):
    """configs[1] speaks of a "single 1M-triangle .obj mesh": write the workload's mesh as a Wavefront .obj, read it back with
    the product's reader (nexus::OBJLoader through nxh_load_scene_file: the route of the reference's OBJLoader::LoadOBJ,
    Assets/OBJLoader.cpp:213-239) and require the triangles it yields to equal the in-memory ones the BVH was built from —
    positions and normals bit for bit, texture coordinates too when the 1 - v flip is exact in float32.  Untimed."""
    import os
    import tempfile
    import time

    from . import loaders

    mesh = sc.meshes[mesh_index]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "mesh.obj")
        t0 = time.time()
        n_verts = loaders.write_obj(path, mesh)
        t_write = time.time() - t0
        size = os.path.getsize(path)
        t0 = time.time()
        meshes, _mats, _insts = capi.load_scene_file(path)
        t_read = time.time() - t0
    if len(meshes) != 1 or len(meshes[0]) != len(mesh):
        raise RuntimeError("obj round trip: %d meshes / %d triangles read, %d written" % (len(meshes), len(meshes[0]) if meshes else 0, len(mesh)))
    got = meshes[0]
    for f in ("pos0", "pos1", "pos2", "normal0", "normal1", "normal2"):
        if not np.array_equal(got[f].view(np.uint32), mesh[f].view(np.uint32)):
            raise RuntimeError("obj round trip: field %s differs from the in-memory mesh" % f)
    uv_exact = all(np.array_equal(got[f].view(np.uint32), mesh[f].view(np.uint32)) for f in ("texCoord0", "texCoord1", "texCoord2"))
    if not uv_exact:
        worst = max(float(np.abs(got[f] - mesh[f]).max()) for f in ("texCoord0", "texCoord1", "texCoord2"))
        if worst > 1e-6:
            raise RuntimeError("obj round trip: texture coordinates differ by %g" % worst)
    return {"triangles": int(len(mesh)), "vertices": int(n_verts), "file_MB": round(size / 1e6, 1), "write_s": round(t_write, 2), "read_s": round(t_read, 2),
            "positions_normals_bit_exact": True, "texcoords_bit_exact": bool(uv_exact), "reader": "nexus::OBJLoader (nxh_load_scene_file)"}
