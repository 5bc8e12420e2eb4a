#!/usr/bin/env python3
"""Could the primary rays of an 8x8 pixel tile start their traversal below the root?  

For a ray to be started at a node N instead of the root with its result unchanged, its traversal up to N must have left nothing
behind: at every ancestor its hit masks must contain the path child only (anything else would be on its stack).  For a TILE to be
started there, that must hold for all 64 rays — i.e. the 64 rays' traversals agree, step for step, on the node visited and on both
hit masks ChildTrace produces (inner children, leaf primitives), and every agreed step leaves exactly one inner child and no leaf.
This tool measures, on the bench scene (configs[1]) and with the oracle's real traversal (oracle/orc_trace.c, node log), for
seeded tiles of the 1080p view:
  agree      the number of leading node steps on which all 64 rays of a tile agree (node and masks): what a tile could do ONCE
  skippable  the number of leading node steps that agree AND leave a single inner child and no leaf: what an entry point skips
against the node steps a ray takes in total.  No GPU needed:  python tools/entry_point_probe.py [--tiles 600]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import config_scenes as CS  # noqa: E402
from tests import oracle_lib as O  # noqa: E402
from tools.lane_sim import camera_rays  # noqa: E402


def node_logs(orc, rays):
    L = O.lib()
    L.orc_trace_set_node_log.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_trace_node_log_length.restype = C.c_uint64
    cap = 3 * (200 * len(rays) + 1024)
    buf = np.zeros(cap, np.uint64)
    L.orc_trace_set_node_log(O._ptr(buf), cap)
    orc.trace_closest(rays, threads=1)
    n = int(L.orc_trace_node_log_length())
    L.orc_trace_set_node_log(None, 0)
    assert n < cap
    t = buf[:n].reshape(-1, 3)
    ends = np.flatnonzero((t[:, 0] == 0) & (t[:, 1] == 0) & (t[:, 2] == 0) & np.r_[True, np.ones(len(t) - 1, bool)])
    # (a TLAS root step is id 0 as well, but its inner mask is never 0 AND leaf mask 0 at once unless the ray misses everything:
    #  such a step is (0, 0, 0) too; tell the two apart by position: a terminator follows every ray)
    seqs, start = [], 0
    k = 0
    while k < len(t):
        # the first triple of a ray is its TLAS root step (id 0); the terminator is the next all-zero triple after it
        e = k + 1
        while not (t[e, 0] == 0 and t[e, 1] == 0 and t[e, 2] == 0):
            e += 1
        seqs.append(t[k:e])
        k = e + 1
    assert len(seqs) == len(rays), (len(seqs), len(rays))
    return seqs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=600)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    scene = CS.config2()
    orc = scene.oracle()
    rays = camera_rays(scene.camera, a.tiles, a.seed)
    seqs = node_logs(orc, rays)
    agree, skip, total = [], [], []
    for tile in range(a.tiles):
        s = seqs[tile * 64:(tile + 1) * 64]
        n = min(len(x) for x in s)
        k = 0
        while k < n and all(np.array_equal(x[k], s[0][k]) for x in s[1:]):
            k += 1
        agree.append(k)
        j = 0
        while j < k and bin(int(s[0][j][1]) >> 24).count("1") == 1 and (int(s[0][j][2]) & 0xffffff) == 0:
            j += 1
        skip.append(j)
        total.append(float(np.mean([len(x) for x in s])))
    agree, skip, total = np.array(agree), np.array(skip), np.array(total)
    print("configs[1], %d seeded 8x8 tiles of the 1080p view (jittered primary rays), node steps per ray: mean %.2f" % (a.tiles, total.mean()))
    print("leading node steps on which all 64 rays of a tile agree (same node, same hit masks): mean %.2f" % agree.mean())
    print("  histogram " + "  ".join("%d: %.1f %%" % (v, 100.0 * np.mean(agree == v)) for v in range(0, int(agree.max()) + 1)))
    print("... of which leave a single inner child and no leaf (what an entry point could skip): mean %.2f" % skip.mean())
    print("  histogram " + "  ".join("%d: %.1f %%" % (v, 100.0 * np.mean(skip == v)) for v in range(0, int(skip.max()) + 1)))
    print("upper bound of the node steps an exact per-tile entry point removes: %.1f %% of the primary level's (%.1f %% of a ray's records incl. triangles are not touched)"
          % (100.0 * skip.sum() / total.sum(), 0.0))
    print("upper bound if the agreed steps were done once per tile instead of per ray (scalar prologue): %.1f %% of the node steps" % (100.0 * agree.sum() / total.sum()))


if __name__ == "__main__":
    main()
