"""How long does a closest-hit launch take as a function of the number of rays (incoherent rays, bench scene)?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nexus_amd import capi, scenegen

W, H = 1920, 1080
sc = bench.workloads.config2(W, H, 1024, 512, 8)
ctx = capi.Context(W, H, device=0)
bench.upload(ctx, sc)
rays_all = scenegen.interior_rays(1 << 21, seed=5, extent=1.6)
rays_all["origin"][:, 1] = np.abs(rays_all["origin"][:, 1]) * 0.6 + 0.05
for n in (64, 1024, 16384, 65536, 262144, 524288, 1048576, 2097152):
    rays = rays_all[:n]
    ctx.trace_batch(rays)
    ctx.enable_trace_stats(True); ctx.read_trace_stats(reset=True)
    ctx.trace_batch(rays)
    st, _ = ctx.read_trace_stats(reset=True)
    ctx.enable_trace_stats(False)
    ctx.enable_kernel_timing(True)
    ctx.read_kernel_times(reset=True)
    for _ in range(5):
        ctx.trace_batch(rays)
    kt = ctx.read_kernel_times(reset=True)
    ctx.enable_kernel_timing(False)
    ms = kt["trace"]["ms"] / kt["trace"]["launches"]
    it = 64.0 * st["waveIters"] / max(1, st["rays"])
    print("rays %8d  %.3f ms  %.1f Mrays/s  iters/ray %.1f  waveIters %d  nodes/ray %.1f  cycles/waveIter %.0f" % (
        n, ms, n / ms / 1e3, it, st["waveIters"], st["nodes"] / n, sum(st["cycles"]) / max(1, st["waveIters"])), flush=True)
