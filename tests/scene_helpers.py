"""Shared scene assembly for the tests: builds BLAS/TLAS with the PRODUCT host builders (nexus_amd.capi) and hands
the same bytes to both the oracle (tests.oracle_lib.OracleScene) and a device context."""
import numpy as np

from nexus_amd import capi, pod, scenegen
from tests import oracle_lib as O


class BuiltScene:
    def __init__(self, meshes, placements, materials=None, lights=None, camera=None, settings=None, diffuse_maps=(), emissive_maps=(),
                 hdr_map=None):
        """meshes: list of TRI_DT arrays; placements: list of (meshIdx, materialId, transform16)."""
        self.meshes = [np.ascontiguousarray(m, dtype=pod.TRI_DT) for m in meshes]
        self.blas = []
        for m in self.meshes:
            nodes, idx = capi.bvh8_build(m, threads=4)
            self.blas.append((nodes, m, idx))
        insts = []
        for mesh_idx, mat_id, xf in placements:
            insts.append(capi.instance_init(mesh_idx, mat_id, xf, self.blas[mesh_idx][0][0]))
        self.instances = np.array(insts, dtype=pod.INST_DT)
        self.tlas_nodes, self.tlas_idx = capi.tlas_build(self.instances)
        self.materials = np.ascontiguousarray(materials if materials is not None else np.array([pod.make_material()], dtype=pod.MAT_DT), dtype=pod.MAT_DT)
        self.lights = np.ascontiguousarray(lights if lights is not None else np.zeros(0, pod.LIGHT_DT), dtype=pod.LIGHT_DT)
        self.camera = camera
        self.settings = settings if settings is not None else O.make_settings()
        self.diffuse_maps, self.emissive_maps, self.hdr_map = list(diffuse_maps), list(emissive_maps), hdr_map

    def oracle(self):
        return O.OracleScene(self.blas, self.instances, self.tlas_nodes, self.tlas_idx, self.materials, self.lights, self.camera, self.settings,
                             self.diffuse_maps, self.emissive_maps, self.hdr_map)

    def upload(self, ctx):
        ctx.clear_blas()
        ctx.clear_textures()
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
        if self.camera is not None:
            ctx.set_camera(self.camera)
        ctx.set_render_settings(self.settings)


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


IDENTITY = np.eye(4, dtype=np.float32).reshape(16)


def soup_scene(n=3000, seed=1):
    return BuiltScene([scenegen.random_soup(n, seed=seed)], [(0, 0, IDENTITY)])


def instanced_scene(seed=3, n_inst=20):
    rng = np.random.RandomState(seed)
    meshes = [scenegen.random_soup(1500, seed=seed, extent=0.5, size=0.08), scenegen.displaced_torus(48, 24, seed=seed, major=0.5, minor=0.2)]
    placements = []
    for i in range(n_inst):
        xf = capi.mat4_from_trs(rng.uniform(-2, 2, 3), rng.uniform(0, 360, 3), rng.uniform(0.5, 1.5, 3))
        placements.append((i % 2, 0, xf))
    return BuiltScene(meshes, placements)


def hit_records_equal(a, b):
    """Bitwise equality of hit records (floats compared as bit patterns)."""
    return (np.array_equal(a["hitDistance"].view(np.uint32), b["hitDistance"].view(np.uint32)) and np.array_equal(a["u"].view(np.uint32), b["u"].view(np.uint32))
            and np.array_equal(a["v"].view(np.uint32), b["v"].view(np.uint32)) and np.array_equal(a["triIdx"], b["triIdx"])
            and np.array_equal(a["instanceIdx"], b["instanceIdx"]))
