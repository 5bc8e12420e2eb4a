#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_pins.npz.

What these vectors are: outputs of THIS repository's CPU oracle (oracle/*.c) and host builders on seeded inputs, frozen so
that a later change to either is noticed, and so that the GPU tests can compare against committed data as well as against
a live oracle run.  They are NOT outputs of the reference program (it cannot be built in this image: CUDA, glm, assimp);
the reference ships no test vectors for this path, so the oracle itself stays "parity unpinned" (oracle/nexus_oracle.h).

    python tests/golden/make_golden.py          # rewrites oracle_pins.npz next to this file

Everything is compared exactly by the tests (tests/test_golden.py): BVH bytes, hit records, RNG streams, and — since the
transcendental functions became the shared text of include/nexus_fmath.h in round 4 — radiance, accumulation, RGBA8 and queue
sizes too, on the CPU (oracle) and on the GPU (HIP path).
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from nexus_amd import pod, scenegen  # noqa: E402
from tests import oracle_lib as O  # noqa: E402
from tests import scene_helpers as SH  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_pins.npz")


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def ray_batch(n, seed, extent, radius):
    a = scenegen.random_rays(n // 2, seed=seed, radius=radius, target_extent=extent)
    b = scenegen.interior_rays(n - n // 2, seed=seed + 1, extent=extent)
    return np.concatenate([a, b])


def hits_to_u32(h):
    """hit records as a (n, 5) uint32 table: t, u, v bit patterns, triIdx, instanceIdx"""
    return np.stack([h["hitDistance"].view(np.uint32), h["u"].view(np.uint32), h["v"].view(np.uint32), h["triIdx"].astype(np.uint32),
                     h["instanceIdx"].astype(np.uint32)], axis=1)


def generate():
    g = {}
    # ---- Cornell box (BASELINE.json configs[0]): builder bytes, primary hits, frames ------------------------------
    W = H = 64
    cb = SH.cornell_scene(W, H, path_length=4)
    g["cornell_blas_node_sha"] = np.stack([sha(b[0]) for b in cb.blas])
    g["cornell_blas_idx_sha"] = np.stack([sha(b[2]) for b in cb.blas])
    g["cornell_tlas_nodes"] = np.frombuffer(np.ascontiguousarray(cb.tlas_nodes).tobytes(), dtype=np.uint8)
    g["cornell_tlas_idx"] = np.asarray(cb.tlas_idx, dtype=np.uint32)
    g["cornell_instances_sha"] = sha(cb.instances)
    rays = ray_batch(4096, 101, extent=1.0, radius=4.0)
    rays["origin"][:, 1] += 1.0  # the box is centred at height 1
    g["cornell_rays"] = np.frombuffer(rays.tobytes(), dtype=np.uint8)
    orc = cb.oracle()
    g["cornell_hits"] = hits_to_u32(orc.trace_closest(rays))
    for mode_name, rng_mode in (("slot", pod.RNG_REFERENCE_SLOT), ("keyed", pod.RNG_PIXEL_KEYED)):
        w = O.Wavefront(orc, W * H, None, rng_mode, pod.CONDUCTOR_REFERENCE)
        frames, queues = [], []
        for f in range(1, 5):
            w.render(f, threads=1)
            w.accumulate(f)
            frames.append(w.radiance().copy())
            q = w.queue_sizes()
            queues.append(np.stack([np.asarray(q[k][:6]) for k in ("traceSize", "traceShadowSize", "diffuseSize", "plasticSize", "dielectricSize", "conductorSize")]))
        g["cornell_radiance_" + mode_name] = np.stack(frames).astype(np.float32)
        g["cornell_queues_" + mode_name] = np.stack(queues).astype(np.int32)
        g["cornell_accumulation_" + mode_name] = w.accumulation().astype(np.float32)
        g["cornell_rgba8_" + mode_name] = w.rgba8().astype(np.uint32)
    # ---- synthetic meshes: 2 k and 50 k triangles, an instanced set with rotations / scales ------------------------
    for name, scene, ext, rad in (("soup2k", SH.soup_scene(n=2000, seed=1), 1.0, 3.0), ("torus50k", SH.BuiltScene([scenegen.displaced_torus(250, 100, seed=3)], [(0, 0, SH.IDENTITY)]), 1.5, 4.0),
                                  ("instanced", SH.instanced_scene(seed=3, n_inst=20), 2.0, 5.0)):
        rays = ray_batch(1024, 7, extent=ext, radius=rad)
        g[name + "_rays"] = np.frombuffer(rays.tobytes(), dtype=np.uint8)
        g[name + "_hits"] = hits_to_u32(scene.oracle().trace_closest(rays))
        g[name + "_node_sha"] = np.stack([sha(b[0]) for b in scene.blas])
        g[name + "_tlas_sha"] = sha(scene.tlas_nodes)
    # ---- material zoo (all four BSDFs, textures, environment): frames ---------------------------------------------
    zoo = SH.material_zoo_scene(48, 32, path_length=4)
    w = O.Wavefront(zoo.oracle(), 48 * 32, None, pod.RNG_PIXEL_KEYED, pod.CONDUCTOR_EXTENDED)
    frames = []
    for f in range(1, 3):
        w.render(f, threads=1)
        w.accumulate(f)
        frames.append(w.radiance().copy())
    g["zoo_radiance_keyed"] = np.stack(frames).astype(np.float32)
    # ---- RNG stream prefixes ---------------------------------------------------------------------------------------
    L = O.lib()
    import ctypes as C

    seeds = []
    for px, py, resx, frame in ((0, 0, 64, 1), (5, 9, 64, 1), (63, 63, 64, 4), (1, 77, 1920, 12)):
        s = C.c_uint32(L.orc_rng_init_pixel(px, py, resx, frame))
        row = [s.value]
        vals = []
        for _ in range(8):
            vals.append(np.float32(L.orc_rand(C.byref(s))).view(np.uint32))
        seeds.append(row + [int(v) for v in vals])
    g["rng_prefixes"] = np.array(seeds, dtype=np.uint32)
    return g


if __name__ == "__main__":
    data = generate()
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(data), "arrays")
