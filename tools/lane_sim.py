#!/usr/bin/env python3
"""What would a wave's 64 lanes be doing under another loop policy?  A CPU-side model of the trace kernel's scheduling, fed with the
REAL per-ray record sequences of the bench scene (the oracle's traversal logs the kinds of records every ray visits, in visiting
order: node, triangle, instance entry; oracle/orc_trace.c orc_trace_set_step_log).  No GPU needed.

The kernel is bound by VALU issue, i.e. by wave-instructions per ray, and an instruction costs the same
whether 5 or 64 lanes are enabled.  Today's loop ("both": every iteration runs the node block and the triangle block for
whichever lanes want them) enables 34 of 64 lanes per instruction.  Policies modelled here:
  both      every iteration: every busy lane advances by one record; cost = overhead + node block (if any lane wants a node or an
            instance) + instance block (if any enters one) + triangle block (if any wants a triangle)
  vote      every iteration runs ONE block kind, the one more lanes are waiting for (weighted by `bias`); the other lanes wait
  postpone  node block every iteration for the lanes that want it; the triangle block only when at least `thresh` lanes are
            waiting for a triangle or no lane wants a node (the reference's triangle postponing, BVH8Traversal.cuh:269-290,
            restated for a wave)
Costs are VALU instructions per block as counted in the kernel's ISA (tools/kernel_resources.py --segments).
Output: wave-instructions per ray and lanes per instruction for each policy, on primary and on bounce rays.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexus_amd import pod  # noqa: E402
from tests import config_scenes as CS  # noqa: E402
from tests import oracle_lib as O  # noqa: E402


def log_sequences(orc, rays):
    """list of uint8 arrays (1 node, 2 triangle, 3 instance) per ray, in the oracle's visiting order"""
    L = O.lib()
    L.orc_trace_set_step_log.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_trace_step_log_length.restype = C.c_uint64
    cap = 400 * len(rays) + 1024
    buf = np.zeros(cap, np.uint8)
    L.orc_trace_set_step_log(O._ptr(buf), cap)
    hits = orc.trace_closest(rays, threads=1)
    n = int(L.orc_trace_step_log_length())
    L.orc_trace_set_step_log(None, 0)
    assert n < cap
    ends = np.flatnonzero(buf[:n] == 0)
    assert len(ends) == len(rays)
    seqs, start = [], 0
    for e in ends:
        s = buf[start:e]
        # the device tests the BLAS root in the iteration that enters the instance: drop the node step that follows an entry
        keep = np.ones(len(s), bool)
        keep[1:][(s[:-1] == 3) & (s[1:] == 1)] = False
        seqs.append(s[keep].copy())
        start = e + 1
    return seqs, hits


def camera_rays(cam, tiles, seed):
    """primary rays of whole 8x8 pixel tiles (the order bench.py's paths have), jittered"""
    rng = np.random.RandomState(seed)
    W, H = int(cam["resolution"][0]), int(cam["resolution"][1])
    tx = rng.randint(0, W // 8, tiles)
    ty = rng.randint(0, H // 8, tiles)
    order = np.lexsort((tx, ty))
    px = (tx[order, None] * 8 + np.tile(np.arange(8), 8)[None, :]).ravel()
    py = (ty[order, None] * 8 + np.repeat(np.arange(8), 8)[None, :]).ravel()
    x = ((px + rng.rand(len(px))) / W)[:, None]
    y = ((py + rng.rand(len(px))) / H)[:, None]
    target = cam["lowerLeftCorner"].astype(np.float64) + cam["viewportX"].astype(np.float64) * x + cam["viewportY"].astype(np.float64) * y
    d = target - cam["position"].astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(len(px), dtype=pod.RAY_DT)
    rays["origin"] = cam["position"]
    rays["direction"] = d.astype(np.float32)
    return rays


def bounce_rays(scene, rays, hits, seed):
    """cosine-distributed continuation rays from the hit points (what bounce 1 traces, by and large)"""
    rng = np.random.RandomState(seed)
    ok = hits["hitDistance"] < 1e29
    r, h = rays[ok], hits[ok]
    p = r["origin"] + r["direction"] * h["hitDistance"][:, None]
    nrm = np.zeros_like(p)
    for inst_id in np.unique(h["instanceIdx"]):
        m = h["instanceIdx"] == inst_id
        tris = scene.blas[int(scene.instances[int(inst_id)]["bvhIdx"])][1]
        t = tris[h["triIdx"][m]]
        n = np.cross(t["pos1"] - t["pos0"], t["pos2"] - t["pos0"])
        nrm[m] = n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
    flip = np.sum(nrm * r["direction"], axis=1) > 0
    nrm[flip] = -nrm[flip]
    n = len(p)
    r1, r2 = rng.rand(n), rng.rand(n)
    phi = 2 * np.pi * r1
    local = np.stack([np.sqrt(r2) * np.cos(phi), np.sqrt(r2) * np.sin(phi), np.sqrt(1 - r2)], 1)
    a = np.where(np.abs(nrm[:, :1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    tx = np.cross(nrm, a)
    tx /= np.linalg.norm(tx, axis=1, keepdims=True)
    ty = np.cross(nrm, tx)
    out = np.zeros(n, dtype=pod.RAY_DT)
    out["origin"] = (p + nrm * 1e-3).astype(np.float32)
    out["direction"] = (tx * local[:, :1] + ty * local[:, 1:2] + nrm * local[:, 2:3]).astype(np.float32)
    return out


def simulate(seqs, policy, cost, refill_below=40, thresh=16, bias=1.0, rays_per_wave=256):
    """Waves of 64 lanes, each with a backlog of `rays_per_wave` consecutive rays; returns (instructions per ray, lanes per instruction,
    iterations per ray-lane)."""
    total_instr = 0.0
    lane_instr = 0.0
    iters = 0
    n_rays = len(seqs)
    for w0 in range(0, n_rays, rays_per_wave):
        queue = list(range(w0, min(n_rays, w0 + rays_per_wave)))
        qpos = 0
        seq = [None] * 64
        pos = [0] * 64
        active = [False] * 64
        while True:
            n_active = sum(active)
            if qpos < len(queue) and n_active < refill_below:
                for l in range(64):
                    if not active[l] and qpos < len(queue):
                        s = seqs[queue[qpos]]
                        qpos += 1
                        if len(s) == 0:
                            continue
                        seq[l], pos[l], active[l] = s, 0, True
                total_instr += cost["refill"]
                lane_instr += cost["refill"] * 32
                n_active = sum(active)
            if n_active == 0:
                if qpos >= len(queue):
                    break
                continue
            kinds = [seq[l][pos[l]] if active[l] else 0 for l in range(64)]
            nN = sum(1 for k in kinds if k == 1 or k == 3)
            nI = sum(1 for k in kinds if k == 3)
            nT = sum(1 for k in kinds if k == 2)
            run_node = run_tri = False
            if policy == "both":
                run_node, run_tri = nN > 0, nT > 0
            elif policy == "vote":
                if nN * cost["tri"] * bias >= nT * cost["node"] or nT == 0:
                    run_node = nN > 0
                    run_tri = nN == 0 and nT > 0
                else:
                    run_tri = True
            elif policy == "postpone":
                run_node = nN > 0
                run_tri = nT > 0 and (nT >= thresh or nN == 0)
            elif policy == "postpone_inst":
                # round 6: instance entries wait until `thresh` lanes want one, or nothing else is left to do (node / triangle lanes go on)
                run_node, run_tri = nN > 0, nT > 0
                run_inst = nI > 0 and (nI >= thresh or (nN - nI == 0 and nT == 0))
            instr = cost["over"]
            lanes = cost["over"] * n_active
            if policy == "postpone_inst":
                nNode = nN - (0 if run_inst else nI)  # lanes in the node block: pure nodes, and the BLAS root test of the entries that run
                if nNode > 0:
                    instr += cost["node"]
                    lanes += cost["node"] * nNode
                if run_inst:
                    instr += cost["inst"]
                    lanes += cost["inst"] * nI
            elif run_node:
                instr += cost["node"]
                lanes += cost["node"] * nN
                if nI:
                    instr += cost["inst"]
                    lanes += cost["inst"] * nI
            if run_tri:
                instr += cost["tri"]
                lanes += cost["tri"] * nT
            total_instr += instr
            lane_instr += lanes
            iters += 1
            for l in range(64):
                k = kinds[l]
                if policy == "postpone_inst":
                    go = (k == 1) or (k == 2) or (k == 3 and run_inst)
                else:
                    go = (run_node and (k == 1 or k == 3)) or (run_tri and k == 2)
                if go:
                    pos[l] += 1
                    if pos[l] >= len(seq[l]):
                        active[l] = False
    return total_instr / n_rays, lane_instr / max(1.0, total_instr), 64.0 * iters / n_rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=400)
    ap.add_argument("--node", type=float, default=213.0)
    ap.add_argument("--tri", type=float, default=62.0)
    ap.add_argument("--inst", type=float, default=45.0)
    ap.add_argument("--over", type=float, default=48.0)
    ap.add_argument("--refill", type=float, default=90.0)
    ap.add_argument("--scene", type=int, default=2, help="2 = configs[1], 4 = configs[3] (transformed instances: --inst 150)")
    args = ap.parse_args()
    cost = {"node": args.node, "tri": args.tri, "inst": args.inst, "over": args.over, "refill": args.refill}
    scene = CS.config4(1920, 1080) if args.scene == 4 else CS.config2(1920, 1080)
    orc = scene.oracle()
    prim = camera_rays(scene.camera, args.tiles, seed=1)
    pseq, phits = log_sequences(orc, prim)
    brays = bounce_rays(scene, prim, phits, seed=2)
    bseq, _ = log_sequences(orc, brays)
    for name, seqs in (("primary rays (8x8 tiles)", pseq), ("bounce rays (cosine, from the primary hits)", bseq)):
        kinds = np.concatenate(seqs) if len(seqs) else np.zeros(0, np.uint8)
        print("%s: %d rays, per ray %.2f nodes %.2f triangles %.2f instance entries" % (name, len(seqs), (kinds == 1).sum() / len(seqs), (kinds == 2).sum() / len(seqs), (kinds == 3).sum() / len(seqs)))
        rows = [("both (today)", dict(policy="both"))]
        for b in (0.5, 1.0, 2.0):
            rows.append(("vote, bias %.1f" % b, dict(policy="vote", bias=b)))
        for t in (4, 8, 16, 24, 32):
            rows.append(("postpone triangles below %d lanes" % t, dict(policy="postpone", thresh=t)))
        for t in (2, 4, 6, 8, 12, 16):
            rows.append(("postpone instance entries below %d lanes" % t, dict(policy="postpone_inst", thresh=t)))
        base = None
        for label, kw in rows:
            ipr, lanes, it = simulate(seqs, cost=cost, **kw)
            base = base or ipr
            print("    %-38s %7.1f wave-instructions per ray (%+5.1f %%)  %5.1f lanes per instruction  %6.2f iterations per ray-lane" % (label, ipr, 100 * (ipr / base - 1), lanes, it))


if __name__ == "__main__":
    main()
