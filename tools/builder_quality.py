"""Tree quality of the three BLAS builders on the GPU box: the host's binned-SAH + SAH-DP collapse (the reference's algorithm),
the device's clustering builder (PLOC, nxhip_set_device_builder radius 16, the default) and its radix-tree builder (LBVH,
radius 0) — build time, node count, nodes / triangles visited per ray (the trace kernel's counting variant) and closest-hit
rays per second, on a regular mesh (the bench's displaced torus) and on irregular ones (triangle soup; the 10 M-triangle
scene's height-field shell and props; the reference's cornell_box_sphere.glb replicated on a lattice).
    python tools/builder_quality.py > profiles/r03_builder_quality.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexus_amd import capi, loaders, pod, scenegen  # noqa: E402

IDENT = np.eye(4, dtype=np.float32).reshape(16)
# device builders: (clustering radius, 0 = radix tree; collapse: the SAH cost table, or round 2's greedy rule)
DEVICE = {"SAH device": (-1, "sah"), "PLOC r16": (16, "sah"), "PLOC r8": (8, "sah"), "LBVH": (0, "sah"), "PLOC r16 greedy": (16, "greedy"), "LBVH greedy": (0, "greedy")}
BUILDERS = tuple(os.environ["BQ_BUILDERS"].split(",")) if os.environ.get("BQ_BUILDERS") else ("host SAH", "SAH device", "PLOC r16", "PLOC r8", "LBVH", "PLOC r16 greedy", "LBVH greedy")


def meshes():
    yield "torus 1 M (regular grid)", scenegen.displaced_torus(1024, 512, seed=1, major=1.0, minor=0.45, amp=0.06)
    yield "triangle soup 300 k", scenegen.random_soup(300000, seed=3, extent=1.0, size=0.02)
    yield "height field 2 M (config 5 shell)", scenegen.height_field(1000, seed=5, amp=0.08)
    ls = loaders.load_glb(os.path.join(ROOT, "tests", "golden", "cornell_box_sphere.glb"))
    parts = []
    rng = np.random.RandomState(4)
    for ix in range(6):
        for iy in range(6):
            for iz in range(6):
                off = np.array([ix, iy, iz], np.float32) * 2.4 + rng.uniform(-0.2, 0.2, 3).astype(np.float32)
                for inst in ls.instances:
                    m = ls.meshes[inst["mesh"]].copy()
                    xf = capi.mat4_from_trs(inst["position"], inst["rotation"], inst["scale"]).reshape(4, 4)
                    for f in ("pos0", "pos1", "pos2"):
                        m[f] = (m[f] @ xf[:3, :3].T + xf[:3, 3] + off).astype(np.float32)
                    parts.append(m)
    yield "cornell_box_sphere.glb x 216, flattened (473 k)", np.concatenate(parts)


def rays_for(tris, n, seed):
    lo = np.minimum(np.minimum(tris["pos0"].min(0), tris["pos1"].min(0)), tris["pos2"].min(0))
    hi = np.maximum(np.maximum(tris["pos0"].max(0), tris["pos1"].max(0)), tris["pos2"].max(0))
    c, ext = (lo + hi) / 2, float(np.max(hi - lo)) / 2
    a = scenegen.random_rays(n // 2, seed=seed, radius=2.5 * ext, target_extent=ext)
    b = scenegen.interior_rays(n - n // 2, seed=seed + 1, extent=ext)
    r = np.concatenate([a, b])
    r["origin"] += c.astype(np.float32)
    return r


def main():
    print("%-48s %-16s %9s %9s %10s %10s %9s" % ("mesh", "builder", "build s", "nodes", "nodes/ray", "tris/ray", "Grays/s"))
    for name, tris in meshes():
        tris = np.ascontiguousarray(tris, dtype=pod.TRI_DT)
        rays = rays_for(tris, 2000000, 7)
        ref = None
        for builder in BUILDERS:
            ctx = capi.Context(1920, 1080)
            ctx.set_frames_per_pass(1)
            t0 = time.time()
            if builder == "host SAH":
                nodes, idx = capi.bvh8_build(tris, threads=0)
                t_build = time.time() - t0
                bid = ctx.upload_blas(nodes, tris, idx)
            else:
                radius, collapse = DEVICE[builder]
                os.environ["NX_TUNING_KNOBS"] = "1"
                os.environ["NX_DEVICE_COLLAPSE"] = collapse  # read by every build
                ctx.set_device_builder(radius)
                ctx.build_blas(tris[:64])  # first use: code objects
                ctx.clear_blas()
                ctx.sync()
                t0 = time.time()
                bid = ctx.build_blas(tris)
                ctx.sync()
                t_build = time.time() - t0
                nodes, idx = ctx.read_blas(bid, len(tris))
            inst = np.array([capi.instance_init(bid, 0, IDENT, nodes[0])], dtype=pod.INST_DT)
            tn, ti = capi.tlas_build(inst)
            ctx.set_tlas(tn, ti, inst)
            ctx.set_materials(np.array([pod.make_material()], dtype=pod.MAT_DT))
            ctx.trace_batch(rays[:1000])
            ctx.enable_trace_stats(True)
            ctx.read_trace_stats(reset=True)
            hits = ctx.trace_batch(rays)
            st, _ = ctx.read_trace_stats(reset=True)
            ctx.enable_trace_stats(False)
            ctx.enable_kernel_timing(True)
            ctx.read_kernel_times(reset=True)
            for _ in range(3):
                ctx.trace_batch(rays)
            kt = ctx.read_kernel_times(reset=True)
            ctx.enable_kernel_timing(False)
            grays = 3 * len(rays) / (kt["trace"]["ms"] * 1e-3) / 1e9
            if ref is None:
                ref = hits
            same = np.array_equal(ref["hitDistance"].view(np.uint32), hits["hitDistance"].view(np.uint32))
            print("%-48s %-16s %9.3f %9d %10.2f %10.2f %9.2f %s" % (name, builder, t_build, len(nodes), st["nodes"] / st["rays"], st["tris"] / st["rays"], grays,
                                                                   "" if same else "HIT DISTANCES DIFFER FROM THE SAH BUILD"))
            ctx.close()


if __name__ == "__main__":
    main()
