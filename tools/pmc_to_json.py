"""Per-ray counter constants of the two trace kernels from rocprofv3 --pmc pass directories, merged into
profiles/trace_counters.json (read by bench.py for roofline.traffic and the VALU-issue ceiling).

    tools/pmc_to_json.py --config 2 --frames F --bench bench_line.json --tag r02 --out profiles/trace_counters.json pass_dir...

`frames` = frames the profiled command rendered (warm-up + steps x reps); the bench line supplies rays per frame of each
kernel (counted by the kernels' counting variant).  gfx950 corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE
(KiB) counts half of the bytes of wide reads -> x2; WRITE_SIZE (KiB) as is."""
import argparse, collections, csv, glob, json, os, re

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, required=True)
ap.add_argument("--frames", type=int, required=True)
ap.add_argument("--bench", required=True)
ap.add_argument("--tag", default="")
ap.add_argument("--command", default="")
ap.add_argument("--out", required=True)
ap.add_argument("passes", nargs="+")
a = ap.parse_args()

# kernel classes: name pattern -> the unit its counters are divided by (rays traced; queue items processed)
# (shade: the one material launch of the SCAN pipeline, or the classic per-type kernels; logic: the classic logic kernel, or what the SCAN pipeline keeps of it
#  under NX_SCAN_SEPARATE; thin: the launch behind a level's trace launches that finishes the rays their dry waves handed over)
KERNELS = {"trace_closest": ("trace_kernel<false, false",), "trace_shadow": ("trace_kernel<true, false",), "logic": ("logic_kernel<", "miss_scan_kernel"),
           "shade": ("shade_kernel<", "shade_scan_kernel"), "thin": ("thin_kernel",)}
bench = json.load(open(a.bench))
rf = bench["roofline"]
rays = {"trace_closest": rf["rays_per_frame"] * a.frames, "trace_shadow": rf["shadow"]["rays_per_frame"] * a.frames,
        "logic": rf.get("items_per_frame", {}).get("logic", 0) * a.frames, "shade": rf.get("items_per_frame", {}).get("shade", 0) * a.frames,
        "thin": rf["rays_per_frame"] * a.frames}  # (thin: per closest-hit ray of the frame, so that its cost adds to the trace kernel's)


import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402  (the one definition of "the code these counters were measured on")

CLASS_OF = {"trace_closest": "trace", "trace_shadow": "trace", "thin": "trace", "logic": "wavefront", "shade": "wavefront"}

sums = {k: collections.defaultdict(float) for k in KERNELS}
frames_of = {}  # counter name -> frames the run that collected it rendered
for d in a.passes:
    frames_d = a.frames
    line = os.path.normpath(d) + ".json"  # tools/profile_round.sh keeps the bench line of every counter run beside its directory
    if os.path.exists(line):
        try:
            frames_d = json.load(open(line))["config"].get("frames_rendered_by_the_timed_loop", a.frames)
        except Exception:
            pass
    dur = {}
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    seen = {k: set() for k in KERNELS}
    names = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for k, pats in KERNELS.items():
                if any(pat in r["Kernel_Name"] for pat in pats):
                    sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    names.add(r["Counter_Name"])
                    if r["Dispatch_Id"] not in seen[k]:
                        seen[k].add(r["Dispatch_Id"])
    # kernel time of exactly the dispatches this pass counted, keyed by the pass's counter set
    for k in KERNELS:
        t = sum(dur[i][1] for i in seen[k] if i in dur)
        for n in names:
            sums[k]["_ns_with_" + n] = t
            frames_of[n] = frames_d

out = json.load(open(a.out)) if os.path.exists(a.out) else {}
cfg = out.setdefault("config%d" % a.config, {})
cfg["profile"] = {"tag": a.tag, "command": a.command, "frames": sorted(set(frames_of.values()))[0] if frames_of else a.frames,
                  "pass_dirs": [os.path.basename(os.path.normpath(p)) for p in a.passes]}
per_frame = {"trace_closest": rf["rays_per_frame"], "trace_shadow": rf["shadow"]["rays_per_frame"], "thin": rf["rays_per_frame"],
             "logic": rf.get("items_per_frame", {}).get("logic", 0) or rf["rays_per_frame"], "shade": rf.get("items_per_frame", {}).get("shade", 0)}
for k in KERNELS:
    s, n = dict(sums[k]), max(1, rays[k])
    if not s:
        continue
    # the units are those of the runs that collected the counters: frames each rendered x units per frame (every counter run
    # of one profile_round is the same command, so they agree; the bench line beside each pass says how many it rendered)
    seen_frames = sorted(set(frames_of.values())) or [a.frames]
    if len(seen_frames) > 1:
        raise SystemExit("counter passes rendered different numbers of frames: %s" % seen_frames)
    n = max(1, per_frame[k] * seen_frames[0])
    rays[k] = n
    e = {"rays_profiled": int(rays[k]), "unit": "ray" if k.startswith("trace") else "queue item", "source_sha16": kernel_source_hash(CLASS_OF[k])}
    if "FETCH_SIZE" in s:
        e["hbm_read_bytes_per_ray"] = round(2.0 * s["FETCH_SIZE"] * 1024.0 / n, 2)
    if "WRITE_SIZE" in s:
        e["hbm_write_bytes_per_ray"] = round(s["WRITE_SIZE"] * 1024.0 / n, 2)
    if "SQ_INSTS_VALU" in s:
        e["valu_insts_per_ray"] = round(s["SQ_INSTS_VALU"] / n, 2)
    if "SQ_INSTS_SALU" in s:
        e["salu_insts_per_ray"] = round(s["SQ_INSTS_SALU"] / n, 2)
    if "GRBM_GUI_ACTIVE" in s and s.get("_ns_with_GRBM_GUI_ACTIVE"):
        e["clock_GHz"] = round(s["GRBM_GUI_ACTIVE"] / 8.0 / s["_ns_with_GRBM_GUI_ACTIVE"], 3)  # the counter sums the 8 XCDs
    if "SQ_WAVE_CYCLES" in s:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in s:
                e[c.lower() + "_frac_of_wave_cycles"] = round(s[c] / s["SQ_WAVE_CYCLES"], 4)
    if "SQ_THREAD_CYCLES_VALU" in s and "SQ_INSTS_VALU" in s:
        e["valu_active_lanes_per_inst"] = round(s["SQ_THREAD_CYCLES_VALU"] / s["SQ_INSTS_VALU"], 2)
    if "SQ_BUSY_CU_CYCLES" in s and s.get("_ns_with_SQ_BUSY_CU_CYCLES"):
        e["sq_busy_cu_cycles_per_ns"] = round(s["SQ_BUSY_CU_CYCLES"] / s["_ns_with_SQ_BUSY_CU_CYCLES"], 3)
    if "TCC_HIT_sum" in s:
        e["l2_hit_rate"] = round(s["TCC_HIT_sum"] / max(1.0, s["TCC_HIT_sum"] + s["TCC_MISS_sum"]), 4)
        e["l2_requests_per_ray"] = round(s.get("TCC_REQ_sum", 0.0) / n, 2)
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in s:
        e["l1_accesses_per_ray"] = round(s["TCP_TOTAL_CACHE_ACCESSES_sum"] / n, 2)
        e["l1_hit_rate"] = round(1.0 - s["TCP_TCC_READ_REQ_sum"] / max(1.0, s["TCP_TOTAL_CACHE_ACCESSES_sum"]), 4)
        if s.get("_ns_with_TCP_TOTAL_CACHE_ACCESSES_sum"):
            e["l1_accesses_per_cu_per_ns"] = round(s["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / s["_ns_with_TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
    if "SQ_INSTS_VMEM_RD" in s:
        e["vmem_rd_insts_per_ray"] = round(s["SQ_INSTS_VMEM_RD"] / n, 3)
    e["raw_sums"] = {c: v for c, v in sorted(s.items()) if not c.startswith("_")}
    cfg[k] = e
json.dump(out, open(a.out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {x: y for x, y in cfg[k].items() if x != "raw_sums"} for k in KERNELS if k in cfg}, indent=1))
