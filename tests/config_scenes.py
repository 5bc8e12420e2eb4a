"""BASELINE.json configs[1..4] as seeded procedural scenes at their full sizes.  Built with the
product host builders; the same bytes go to the oracle and to the device context (tests.scene_helpers.BuiltScene)."""
import numpy as np

from nexus_amd import capi, pod, scenegen
from tests import oracle_lib as O
from tests import scene_helpers as SH


def _look(eye, target, hfov, width, height):
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    return capi.camera_init(eye, fwd, hfov, width, height, 5.0, 0.0)


def config2(width=1920, height=1080, nu=1024, nv=512, path_length=8):
    """configs[1]: the bench workload (bench.build_config2 builds the identical scene)."""
    torus = scenegen.displaced_torus(nu, nv, seed=1, major=1.0, minor=0.45, amp=0.06, center=(0.0, 0.56, 0.0))
    floor = scenegen.quad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6))
    light = scenegen.quad((-1.2, 4.0, -1.2), (1.2, 4.0, -1.2), (1.2, 4.0, 1.2), (-1.2, 4.0, 1.2))
    mats = np.array([
        pod.make_material(pod.MAT_CONDUCTOR, roughness=0.3, conductor_ior=(0.2, 0.9, 1.1), conductor_k=(3.9, 2.4, 2.2)),
        pod.make_material(pod.MAT_DIFFUSE, albedo=(0.7, 0.7, 0.7)),
        pod.make_material(pod.MAT_DIFFUSE, albedo=(0.8, 0.8, 0.8), emissive=(1.0, 1.0, 1.0), intensity=20.0),
    ], dtype=pod.MAT_DT)
    sc = SH.BuiltScene([torus, floor, light], [(i, i, SH.IDENTITY) for i in range(3)], materials=mats,
                       camera=_look((0.0, 3.3, 4.9), (0.0, 0.35, 0.0), 52.0, width, height),
                       settings=O.make_settings(use_mis=True, path_length=path_length, background=(1, 1, 1), background_intensity=0.0), build_threads=0)
    sc.lights = SH.mesh_lights(sc.instances, sc.materials)
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


def config4(width=1920, height=1080, path_length=8, n_side=10, nu=250, nv=200):
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
    sc = SH.BuiltScene([mesh], placements, materials=mats, camera=_look((ext * 0.9, ext * 0.55, ext * 1.25), (0, 0, 0), 45.0, width, height),
                       settings=O.make_settings(use_mis=True, path_length=path_length, background=(1, 1, 1), background_intensity=1.0),
                       hdr_map=procedural_sky(), build_threads=0)
    return sc


def config5(width=3840, height=2160, path_length=16, field=2200, prop_nu=512, prop_nv=256, n_props=16):
    """configs[4]: ~10 M triangles — a 2*field^2-triangle displaced room shell (floor / back wall / ceiling from one
    height-field BLAS, rotated) plus instanced props, all four material types, emissive-textured light panels,
    3840x2160, pathLength 16."""
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
    sc = SH.BuiltScene([shell, prop, panel], placements, materials=mats, camera=_look((0.0, 2.6, 9.5), (0.0, 1.8, 0.0), 55.0, width, height),
                       settings=O.make_settings(use_mis=True, path_length=path_length, background=(0.5, 0.6, 0.8), background_intensity=0.3),
                       diffuse_maps=[SH.checker_texture(256, 256, 11)], emissive_maps=[SH.checker_texture(64, 64, 12)], build_threads=0)
    sc.lights = SH.mesh_lights(sc.instances, sc.materials)
    return sc
