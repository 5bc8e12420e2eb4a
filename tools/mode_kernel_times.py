"""Kernel times per class of one pass of configs[1] in the three compaction / RNG configurations bench.py reports (fast; ordered = slot
RNG + grid-wide ordered compaction, same work; reference = ordered without the conductor kernel): where the ordered mode's time goes.
    python tools/mode_kernel_times.py [frames_per_pass]  > profiles/r04_mode_kernel_times.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexus_amd import capi, multigpu, pod, workloads  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H = 1920, 1080
sc = workloads.config2(W, H, 1024, 512, 8)
ctx = capi.Context(W, H, device=0)
sc.upload(ctx, device_bvh=True, device_tlas=True)
ctx.set_pixel_map(multigpu.tiled_order(np.arange(W * H, dtype=np.uint32), W))
ctx.set_frames_per_pass(S)
print("configs[1], one pass of %d frames; per-class kernel time in ms per frame (hipEvent nodes around every kernel of the pass graph, last of 3 replays)" % S)
print("%-34s %8s %8s %8s %8s %8s %8s %9s" % ("mode", "generate", "trace", "shadow", "logic", "shade", "accum", "sum"))
for name, modes in (("fast (pixel RNG, racing slots)", (pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)),
                    ("ordered (slot RNG, serial slots)", (pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED, pod.CONDUCTOR_EXTENDED)),
                    ("pixel RNG, serial slots", (pod.RNG_PIXEL_KEYED, pod.COMPACT_ORDERED, pod.CONDUCTOR_EXTENDED)),
                    ("reference (no conductor kernel)", (pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED, pod.CONDUCTOR_REFERENCE))):
    ctx.set_modes(*modes)
    ctx.reset_frame_number()
    for _ in range(2):
        ctx.render_frame()
        ctx.accumulate()
    ctx.sync()
    ctx.enable_kernel_timing(True, in_graph=True, last_replay_only=True)
    ctx.read_kernel_times(reset=True)
    for _ in range(3):
        ctx.render_frame()
        ctx.accumulate()
    kt = ctx.read_kernel_times(reset=True)
    ctx.enable_kernel_timing(False)
    v = [kt[k]["ms"] / (S * 3 if k == "accumulate" else S) for k in ("generate", "trace", "shadow", "logic", "shade", "accumulate")]
    # (shadow runs beside trace: the sum counts the longer of the two)
    print("%-34s %8.4f %8.4f %8.4f %8.4f %8.4f %8.4f %9.4f" % (name, v[0], v[1], v[2], v[3], v[4], v[5], v[0] + max(v[1], v[2]) + v[3] + v[4] + v[5]))
ctx.close()
