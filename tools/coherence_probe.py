"""How much does the order of the rays in the queue matter to the closest-hit kernel?  The same 4 M secondary-like rays (origins
on the bench mesh's surface, cosine-distributed directions) traced in random order, grouped by direction octant (what a
workgroup-local octant binning in the material kernels could deliver at best), and sorted by octant + Morton code of the origin
(an upper bound no cheap binning reaches).  Prints rays per second and the counting variant's lane statistics."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexus_amd import capi, pod, scenegen  # noqa: E402

IDENT = np.eye(4, dtype=np.float32).reshape(16)


def morton(q):
    def spread(v):
        v = v.astype(np.uint64) & 0x3ff
        v = (v | (v << 16)) & 0x30000ff
        v = (v | (v << 8)) & 0x300f00f
        v = (v | (v << 4)) & 0x30c30c3
        v = (v | (v << 2)) & 0x9249249
        return v
    return (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])


def main():
    tris = scenegen.displaced_torus(1024, 512, seed=1, major=1.0, minor=0.45, amp=0.06)
    nodes, idx = capi.bvh8_build(tris, threads=0)
    ctx = capi.Context(1920, 1080)
    ctx.set_frames_per_pass(2)
    bid = ctx.upload_blas(nodes, tris, idx)
    inst = np.array([capi.instance_init(bid, 0, IDENT, nodes[0])], dtype=pod.INST_DT)
    tn, ti = capi.tlas_build(inst)
    ctx.set_tlas(tn, ti, inst)
    ctx.set_materials(np.array([pod.make_material()], dtype=pod.MAT_DT))
    rng = np.random.RandomState(3)
    n = 4000000
    t = tris[rng.randint(0, len(tris), n)]
    u, v = rng.rand(n).astype(np.float32), rng.rand(n).astype(np.float32)
    flip = u + v > 1
    u[flip], v[flip] = 1 - u[flip], 1 - v[flip]
    p = t["pos0"] + (t["pos1"] - t["pos0"]) * u[:, None] + (t["pos2"] - t["pos0"]) * v[:, None]
    nrm = t["normal0"] / np.maximum(np.linalg.norm(t["normal0"], axis=1, keepdims=True), 1e-20)
    r1, r2 = rng.rand(n), rng.rand(n)
    phi = 2 * np.pi * r1
    local = np.stack([np.sqrt(r2) * np.cos(phi), np.sqrt(r2) * np.sin(phi), np.sqrt(1 - r2)], 1)
    a = np.where(np.abs(nrm[:, :1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    tx = np.cross(nrm, a)
    tx /= np.linalg.norm(tx, axis=1, keepdims=True)
    ty = np.cross(nrm, tx)
    d = (tx * local[:, :1] + ty * local[:, 1:2] + nrm * local[:, 2:3]).astype(np.float32)
    rays = np.zeros(n, dtype=pod.RAY_DT)
    rays["origin"] = (p + nrm * 1e-3).astype(np.float32)
    rays["direction"] = d
    octant = ((d[:, 0] < 0).astype(np.int64) << 2) | ((d[:, 1] < 0).astype(np.int64) << 1) | (d[:, 2] < 0).astype(np.int64)
    lo, hi = rays["origin"].min(0), rays["origin"].max(0)
    q = np.clip((rays["origin"] - lo) / (hi - lo) * 1023, 0, 1023).astype(np.uint32)
    orders = {"random order": rng.permutation(n), "surface order (as generated per triangle pick)": np.arange(n),
              "grouped by octant within blocks of 256": np.concatenate([b + np.argsort(octant[b:b + 256], kind="stable") for b in range(0, n, 256)]),
              "grouped by octant within blocks of 4096": np.concatenate([b + np.argsort(octant[b:b + 4096], kind="stable") for b in range(0, n, 4096)]),
              "sorted by octant": np.argsort(octant, kind="stable"),
              "sorted by octant, then Morton code of the origin": np.lexsort((morton(q), octant))}
    # "blocks" only make sense on an order that is already spatially local: apply them to the random order too
    perm = orders["random order"]
    orders["random, then grouped by octant within blocks of 256"] = np.concatenate([perm[b:b + 256][np.argsort(octant[perm[b:b + 256]], kind="stable")] for b in range(0, n, 256)])
    print("%-56s %9s %10s %9s %9s %9s" % ("order of the 4 M rays in the queue", "Grays/s", "nodes/ray", "active", "node", "prim"))
    for name, order in orders.items():
        r = np.ascontiguousarray(rays[order])
        ctx.trace_batch(r[:100000])
        ctx.enable_trace_stats(True)
        ctx.read_trace_stats(reset=True)
        ctx.trace_batch(r)
        st, _ = ctx.read_trace_stats(reset=True)
        ctx.enable_trace_stats(False)
        ctx.enable_kernel_timing(True)
        ctx.read_kernel_times(reset=True)
        for _ in range(3):
            ctx.trace_batch(r)
        kt = ctx.read_kernel_times(reset=True)
        ctx.enable_kernel_timing(False)
        it = max(1, 64 * st["waveIters"])
        print("%-56s %9.2f %10.2f %9.3f %9.3f %9.3f" % (name, 3 * n / (kt["trace"]["ms"] * 1e-3) / 1e9, st["nodes"] / st["rays"], st["lanesActive"] / it, st["lanesNode"] / it, st["lanesPrim"] / it))
    ctx.close()


if __name__ == "__main__":
    main()
