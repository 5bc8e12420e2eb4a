"""Experiment: aggregate Msamples/s of K independent contexts (own queues, own stream each) that render small passes
concurrently on one GPU: does the drain phase of one pass overlap with the bulk of another?"""
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
    return round(W * H * frames / dt / 1e6, 1), round(dt / frames * 1e3, 3)


for S, passes in ((10, 6), (7, 6), (20, 4)):
    ctxs = []
    for K in (1, 2, 3):
        while len(ctxs) < K:
            ctxs.append(make(S))
        print("S", S, "contexts", K, run(ctxs, passes, S), flush=True)
    for c in ctxs:
        c.close()
