#!/usr/bin/env python3
"""Experiment: how long is every PRIMARY ray of configs[1] — records visited, from the oracle's step log —
and does starting the slow 8x8 tiles first shorten the primary trace launch?  Writes the per-pixel record counts (row-major uint16) to
tools/_pixel_cost.npy for bench.py's NX_BENCH_PIXEL_COST / NX_BENCH_SLOW_FIRST experiment switches.  CPU only, a few minutes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import lane_sim as LS  # noqa: E402
from nexus_amd import pod  # noqa: E402
from tests import config_scenes as CS  # noqa: E402

W, H = 1920, 1080
scene = CS.config2(W, H)
orc = scene.oracle()
cam = scene.camera
cost = np.zeros(W * H, np.uint16)
rows_per_chunk = 40
for y0 in range(0, H, rows_per_chunk):
    ys = np.arange(y0, min(H, y0 + rows_per_chunk))
    px = np.tile(np.arange(W), len(ys))
    py = np.repeat(ys, W)
    x = ((px + 0.5) / W)[:, None]
    y = ((py + 0.5) / H)[:, None]
    target = cam["lowerLeftCorner"].astype(np.float64) + cam["viewportX"].astype(np.float64) * x + cam["viewportY"].astype(np.float64) * y
    d = target - cam["position"].astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(len(px), dtype=pod.RAY_DT)
    rays["origin"] = cam["position"]
    rays["direction"] = d.astype(np.float32)
    seqs, _ = LS.log_sequences(orc, rays)
    cost[y0 * W:(y0 + len(ys)) * W] = np.minimum(65535, [len(s) for s in seqs])
    print("rows %d-%d: mean %.1f max %d" % (y0, ys[-1], cost[y0 * W:(y0 + len(ys)) * W].mean(), cost[y0 * W:(y0 + len(ys)) * W].max()), flush=True)
np.save(os.path.join(ROOT, "tools", "_pixel_cost.npy"), cost)
t = cost.reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3)).ravel()
print("pixels: mean %.2f  p99 %d  p99.9 %d  max %d" % (cost.mean(), np.percentile(cost, 99), np.percentile(cost, 99.9), cost.max()))
print("tiles (max over 64 pixels): p50 %d p90 %d p99 %d max %d; tiles above 100: %d of %d" % (np.percentile(t, 50), np.percentile(t, 90), np.percentile(t, 99), t.max(), (t > 100).sum(), len(t)))
