"""Experiment: do two independent passes in flight (two contexts, two streams) overlap usefully on one GPU?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nexus_amd import capi, multigpu

W, H = 1920, 1080
sc = bench.workloads.config2(W, H, 1024, 512, 8)
pm = multigpu.tiled_order(np.arange(W * H, dtype=np.uint32), W)


def make(S):
    c = capi.Context(W, H, device=0)
    bench.upload(c, sc)
    c.set_pixel_map(pm)
    c.set_frames_per_pass(S)
    return c


def run(ctxs, passes, S):
    for c in ctxs:
        c.render_frame(); c.accumulate()
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    for _ in range(passes):
        for c in ctxs:
            c.render_frame(); c.accumulate()
    for c in ctxs:
        c.sync()
    dt = time.perf_counter() - t0
    frames = passes * len(ctxs) * S
    return W * H * frames / dt / 1e6, dt / frames * 1e3


for S in (32, 16):
    for blocks in ("", "4", "3"):
        if blocks:
            os.environ["NX_TUNING_KNOBS"] = "1"
            os.environ["NX_TRACE_BLOCKS_PER_CU"] = blocks
        else:
            os.environ.pop("NX_TRACE_BLOCKS_PER_CU", None)
        one = [make(S)]
        print("S", S, "blocks", blocks or "auto", "one ctx  ", run(one, 4, S), flush=True)
        two = one + [make(S)]
        print("S", S, "blocks", blocks or "auto", "two ctxs ", run(two, 2, S), flush=True)
        for c in two:
            c.close()
