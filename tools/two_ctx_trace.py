"""Experiment: two contexts (two streams), S frames per pass each, passes issued alternately; run under
rocprofv3 --kernel-trace and analyse with tools/overlap.py: do kernels of the two streams overlap in time?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nexus_amd import capi, multigpu
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W, H = 1920, 1080
sc = bench.workloads.config2(W, H, 1024, 512, 8)
pm = multigpu.tiled_order(np.arange(W * H, dtype=np.uint32), W)
ctxs = []
for _ in range(K):
    c = capi.Context(W, H, device=0)
    bench.upload(c, sc); c.set_pixel_map(pm); c.set_frames_per_pass(S)
    ctxs.append(c)
for _ in range(6):
    for c in ctxs:
        c.render_frame(); c.accumulate()
for c in ctxs:
    c.sync()
for c in ctxs:
    c.close()
