"""Copy what tools/profile_round.sh left under gpurun_out/prof_<tag>/ into profiles/ (summaries only) and merge the per-unit
counter constants of the given (config, tag) pairs into profiles/trace_counters.json:
    python tools/collect_profiles.py 2:r03_cfg2_s20 5:r03_cfg5"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for spec in sys.argv[1:]:
    cfg, tag = spec.split(":")
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    out["config" + cfg] = json.load(open(os.path.join(src, "trace_counters.json")))["config" + cfg]
    for f in ("bench_line.json", "kernel_stats.csv", "bench_under_rocprof.json", "pass_timeline.txt", "per_ray.txt"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(ROOT, "profiles", "%s_%s" % (tag, f)))
    with open(os.path.join(ROOT, "profiles", tag + "_pmc_summary.txt"), "w") as o:
        for i in range(1, 7):
            p = os.path.join(src, "pmc%d.summary.txt" % i)
            if os.path.exists(p):
                o.write(open(p).read())
json.dump(out, open(os.path.join(ROOT, "profiles", "trace_counters.json"), "w"), indent=1, sort_keys=True)
print("merged", sorted(out))
