#!/usr/bin/env python3
"""Where a trace launch's time goes, wave by wave: reads the array a measurement build (tools/build_variant.sh wavetl
"-DNX_WAVE_TIMELINE=1") keeps — every wave's start, the moment it found the queue dry, its end (100 MHz chip clock), rays taken,
loop iterations, rays handed to the thin kernel — as bench.py saves it (NX_WAVE_TIMELINE_OUT=file.npy), for the LAST launch of
every level.

    NEXUS_AMD_LIB=nexus_amd/lib/variants/lib_wavetl.so NX_WAVE_TIMELINE_OUT=gpurun_out/x.npy python bench.py --steps 20 --warmup 5 --reps 1 ...
    python tools/wave_timeline.py gpurun_out/x.npy
"""
import sys

import numpy as np


def q(a, p):
    return float(np.quantile(a, p)) if len(a) else float("nan")


def main():
    t = np.load(sys.argv[1])
    for bounce in range(t.shape[1]):
        for kind in (0, 1):
            rec = t[kind, bounce]
            live = rec[:, 2] != 0
            if not live.any():
                continue
            r = rec[live]
            t0 = r[:, 0].min()
            start = (r[:, 0] - t0) / 100.0
            end = (r[:, 2] - t0) / 100.0
            dry = np.where(r[:, 1] != 0, (r[:, 1].astype(np.int64) - np.int64(t0)) / 100.0, np.nan)
            taken = (r[:, 3] >> np.uint64(32)).astype(np.int64)
            iters = (r[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
            handed = r[:, 4].astype(np.int64)
            span = end.max()
            # chip occupancy over time: waves alive in each tenth of the launch
            edges = np.linspace(0, span, 11)
            alive = [int(((start <= e) & (end > e)).sum()) for e in edges[:-1] + span / 20]
            print("%s bounce %d: %d waves, %d rays, launch %.0f us | wave start max %.0f | queue dry p1 %.0f p50 %.0f p99 %.0f | wave end p1 %.0f p10 %.0f p50 %.0f "
                  "p90 %.0f p99 %.0f | after dry: p50 %.0f p90 %.0f max %.0f us | iterations per wave p50 %d max %d, per ray %.3f | us per iteration %.2f | handed %d"
                  % ("any-hit" if kind else "closest", bounce, len(r), taken.sum(), span, start.max(), q(dry[~np.isnan(dry)], .01), q(dry[~np.isnan(dry)], .5),
                     q(dry[~np.isnan(dry)], .99), q(end, .01), q(end, .1), q(end, .5), q(end, .9), q(end, .99),
                     q((end - dry)[~np.isnan(dry)], .5), q((end - dry)[~np.isnan(dry)], .9), np.nanmax(end - dry) if (~np.isnan(dry)).any() else float("nan"),
                     int(np.median(iters)), iters.max(), iters.sum() / max(1, taken.sum()), float(np.median((end - start) / np.maximum(iters, 1))), handed.sum()))
            print("    waves alive at 5 %% ... 95 %% of the launch: %s" % alive)


if __name__ == "__main__":
    main()
