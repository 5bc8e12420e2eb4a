#!/usr/bin/env python3
"""configs[1] through the kept C++ boundary: writes bench.py's mesh as a Wavefront .obj, runs
nexus_amd/lib/nexus_bench (examples/nexus_bench.cpp: nexus::Scene / OBJLoader / PathTracer::Render only) in the reference's own
settings and in bench.py's, then bench.py itself on the same box, and prints the three lines + the ratio.

This process never touches the GPU (the three programs are its children), so it may start them freely.
    python tools/facade_bench.py [--steps 20 --warmup 5 --out gpurun_out/facade]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--nu", type=int, default=1024)
    ap.add_argument("--nv", type=int, default=512)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "facade"))
    args = ap.parse_args()
    from nexus_amd import loaders, scenegen  # numpy only

    os.makedirs(args.out, exist_ok=True)
    exe = os.path.join(ROOT, "nexus_amd", "lib", "nexus_bench")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "bench_example"], check=True)
    lines = {}
    with tempfile.TemporaryDirectory() as d:
        t0 = time.time()
        torus = scenegen.displaced_torus(args.nu, args.nv, seed=1, major=1.0, minor=0.45, amp=0.06, center=(0.0, 0.56, 0.0))  # workloads.config2's mesh
        loaders.write_obj(os.path.join(d, "mesh.obj"), torus)
        print("[facade %6.1f s] %d triangles written as %s (%.0f MB)" % (time.time() - t0, len(torus), os.path.join(d, "mesh.obj"), os.path.getsize(os.path.join(d, "mesh.obj")) / 1e6), file=sys.stderr)
        for mode in ("headline", "reference", "reference+3"):
            extra = ["--passes-in-flight", "3"] if mode.endswith("+3") else []  # (what a viewer that does not wait for every frame gets)
            r = subprocess.run([exe, d + "/", "mesh.obj", "--mode", mode.split("+")[0], "--frames", str(args.steps), "--warmup", str(args.warmup), "--reps", str(args.reps)] + extra,
                               capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                print(r.stderr, file=sys.stderr)
                raise SystemExit("nexus_bench --mode %s failed (%d)" % (mode, r.returncode))
            lines[mode] = json.loads(r.stdout.strip().splitlines()[-1])
            open(os.path.join(args.out, "facade_%s.json" % mode), "w").write(json.dumps(lines[mode]) + "\n")
            print("[facade %6.1f s] %s: %.1f Msamples/s" % (time.time() - t0, mode, lines[mode]["value"]), file=sys.stderr)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-obj-check"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    if r.returncode != 0:
        print(r.stderr[-2000:], file=sys.stderr)
        raise SystemExit("bench.py failed")
    b = json.loads(r.stdout.strip().splitlines()[-1])
    open(os.path.join(args.out, "bench_py.json"), "w").write(json.dumps(b) + "\n")
    ref_mode = b.get("config", {}).get("reference_mode", {})
    out = {"through_the_cpp_api_headline_settings": lines["headline"]["value"], "bench_py_value": b["value"],
           "ratio": round(lines["headline"]["value"] / b["value"], 4),
           "through_the_cpp_api_reference_defaults": lines["reference"]["value"],
           "through_the_cpp_api_reference_defaults_3_calls_in_flight": lines["reference+3"]["value"],
           "bench_py_reference_mode": ref_mode, "same_image_as_bench_py": None}
    print(json.dumps(out))
    open(os.path.join(args.out, "summary.json"), "w").write(json.dumps(out) + "\n")


if __name__ == "__main__":
    main()
