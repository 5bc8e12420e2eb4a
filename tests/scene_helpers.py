"""Shared scene assembly for the tests: builds BLAS/TLAS with the PRODUCT host builders (nexus_amd.capi) and hands
the same bytes to both the oracle (tests.oracle_lib.OracleScene) and a device context."""
import numpy as np

from nexus_amd import capi, pod, scenegen, workloads
from tests import oracle_lib as O


class BuiltScene(workloads.Workload):
    """A product workload plus the oracle's view of the very same bytes."""

    def oracle(self):
        return O.OracleScene(self.blas, self.instances, self.tlas_nodes, self.tlas_idx, self.materials, self.lights, self.camera, self.settings,
                             self.diffuse_maps, self.emissive_maps, self.hdr_map, env_sampling=self.env_sampling)


mesh_lights = workloads.mesh_lights
checker_texture = workloads.checker_texture
IDENTITY = workloads.IDENTITY


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


import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cornell_scene(width=512, height=512, path_length=4, force_diffuse=True, use_mis=True):
    """BASELINE.json configs[0]: the reference's cornell_box.glb (8 primitives -> 8 BLAS/instances, node rotated +90 deg
    about X), all materials DIFFUSE with the glb base colours, light emissive (1,1,1) x 35, camera at (0,1,3.9) looking
    down -z with a 40 degree horizontal FOV (the file carries no camera; SURVEY.md section 8d fixes these numbers)."""
    return workloads.config1(os.path.join(GOLDEN, "cornell_box.glb"), width, height, path_length, force_diffuse, use_mis, cls=BuiltScene)


def glb_scene(path, width, height, path_length=4, eye=(0.0, 1.0, 3.9), forward=(0.0, 0.0, -1.0), hfov=40.0, use_mis=True):
    """A .glb read by the Python reader (nexus_amd.loaders), materials and textures as the file gives them: what
    Scene::CreateMeshInstanceFromFile builds (Scene.cpp:83-91), as a BuiltScene."""
    from nexus_amd import loaders

    ls = loaders.load_glb(path)
    mats = ls.materials.copy()
    kinds = {"diffuse": [], "emissive": []}
    tex_id = []
    for kind, px in ls.textures:  # ids are positions in the per-kind lists, as AssetManager::AddTexture hands them out
        tex_id.append(len(kinds[kind]))
        kinds[kind].append(px)
    for i in range(len(mats)):
        if ls.material_diffuse_texture[i] >= 0:
            mats["diffuseMapId"][i] = tex_id[ls.material_diffuse_texture[i]]
        if ls.material_emissive_texture[i] >= 0:
            mats["emissiveMapId"][i] = tex_id[ls.material_emissive_texture[i]]
    placements = [(inst["mesh"], inst["material"], capi.mat4_from_trs(inst["position"], inst["rotation"], inst["scale"])) for inst in ls.instances]
    cam = capi.camera_init(eye, forward, hfov, width, height, 5.0, 0.0)
    settings = O.make_settings(use_mis=use_mis, path_length=path_length, background=(1, 1, 1), background_intensity=0.0)
    sc = BuiltScene(ls.meshes, placements, materials=mats, camera=cam, settings=settings, diffuse_maps=kinds["diffuse"], emissive_maps=kinds["emissive"])
    sc.lights = mesh_lights(sc.instances, sc.materials)
    return sc


def material_zoo_scene(width=96, height=64, path_length=5, hdr=True, textures=True):
    """Every material type, an emissive-textured light, diffuse texture with alpha, opacity < 1, instanced + rotated BLAS,
    equirectangular background."""
    torus = scenegen.displaced_torus(40, 20, seed=4, major=0.5, minor=0.22)
    floor = scenegen.quad((-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4))
    light = scenegen.quad((-0.8, 0, -0.8), (0.8, 0, -0.8), (0.8, 0, 0.8), (-0.8, 0, 0.8))
    mats = np.array([
        pod.make_material(pod.MAT_DIFFUSE, albedo=(0.7, 0.7, 0.7), diffuse_map=0 if textures else -1),
        pod.make_material(pod.MAT_PLASTIC, albedo=(0.8, 0.3, 0.2), roughness=0.4, ior=1.5),
        pod.make_material(pod.MAT_DIELECTRIC, albedo=(0.95, 0.97, 1.0), roughness=0.2, ior=1.45),
        pod.make_material(pod.MAT_CONDUCTOR, roughness=0.3),
        pod.make_material(pod.MAT_DIFFUSE, albedo=(0.8, 0.8, 0.8), emissive=(1.0, 0.9, 0.8), intensity=12.0, emissive_map=0 if textures else -1),
        pod.make_material(pod.MAT_DIFFUSE, albedo=(0.3, 0.6, 0.9), opacity=0.6),
    ], dtype=pod.MAT_DT)
    placements = [
        (1, 0, capi.mat4_from_trs((0, 0, 0))),
        (0, 1, capi.mat4_from_trs((-1.4, 0.55, 0.0), (20, 30, 0))),
        (0, 2, capi.mat4_from_trs((0.0, 0.55, 0.3), (90, 0, 15), (1.1, 1.1, 1.1))),
        (0, 3, capi.mat4_from_trs((1.4, 0.6, -0.2), (0, 60, 40), (1.0, 1.3, 1.0))),
        (0, 5, capi.mat4_from_trs((0.2, 0.5, 1.6), (45, 0, 0), (0.7, 0.7, 0.7))),
        (2, 4, capi.mat4_from_trs((0, 3.0, 0), (180, 0, 0))),
    ]
    cam = capi.camera_init((0.0, 1.6, 5.0), (0.0, -0.2, -0.98), 50.0, width, height, 5.0, 1.5)
    settings = O.make_settings(use_mis=True, path_length=path_length, background=(0.6, 0.7, 0.9), background_intensity=0.5)
    sc = BuiltScene([torus, floor, light], placements, materials=mats, camera=cam, settings=settings,
                    diffuse_maps=[checker_texture(64, 32, 1, alpha=True)] if textures else (), emissive_maps=[checker_texture(32, 32, 2)] if textures else (),
                    hdr_map=checker_texture(128, 64, 3) if hdr else None)
    sc.lights = mesh_lights(sc.instances, sc.materials)
    return sc


def frames_identical(got, want, what=""):
    """Device frame against oracle frame, BIT FOR BIT (since round 4 both sides compile the same text for the transcendental
    functions, include/nexus_fmath.h, so nothing is left to tolerate).  Always prints what was measured — pixels identical,
    pixels within 1e-3, the first pixels that differ — so that a failing comparison says how far off it was."""
    got = np.ascontiguousarray(got, dtype=np.float32).reshape(-1, 3)
    want = np.ascontiguousarray(want, dtype=np.float32).reshape(-1, 3)
    if got.shape != want.shape:
        print("%s: shapes differ %r / %r" % (what, got.shape, want.shape))
        return False
    same = np.all(got.view(np.uint32) == want.view(np.uint32), axis=1)
    n_same = int(same.sum())
    if n_same != len(same):
        bad = np.flatnonzero(~same)
        print("%s: %d of %d pixels identical (%.6f), %.6f within 1e-3; first differing pixels %s: device %s, oracle %s" % (
            what, n_same, len(same), n_same / len(same), image_agreement(got, want), bad[:4].tolist(), got[bad[:4]].tolist(), want[bad[:4]].tolist()))
    return n_same == len(same)


QUEUE_KEYS = ("traceSize", "traceShadowSize", "diffuseSize", "plasticSize", "dielectricSize", "conductorSize")


def queue_sizes_identical(got, want, slots=None):
    """Every queue size of every bounce equal (device read-back dict / oracle dict)."""
    ok = True
    for k in QUEUE_KEYS:
        a, b = np.asarray(got[k])[:slots], np.asarray(want[k])[:slots]
        if not np.array_equal(a, b):
            first = int(np.flatnonzero(a != b)[0])
            print("queue %s differs first at bounce %d: device %d, oracle %d" % (k, first, a[first], b[first]))
            ok = False
    return ok


def image_agreement(got, want, rel=1e-3):
    """Fraction of pixels whose RGB all satisfy |got - want| <= rel * max(1, |want|)."""
    tol = rel * np.maximum(1.0, np.abs(want))
    ok = np.all(np.abs(got - want) <= tol, axis=1)
    return float(ok.mean())
