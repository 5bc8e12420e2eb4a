"""Register / scratch / LDS footprint of every device kernel, from the compiler's own metadata (hipcc -S of the device files
with the Makefile's flags): VGPRs, AGPRs, SGPRs, spilled VGPRs, scratch bytes per lane, LDS bytes, occupancy (waves per SIMD).
    python tools/kernel_resources.py > profiles/r03_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Iinclude", "-Inexus_amd/csrc/device", "-Inexus_amd/csrc/host", "-fno-slp-vectorize", "-mllvm", "-amdgpu-use-amdgpu-trackers=1", "-mllvm", "-amdgpu-sched-strategy=max-memory-clause",
         "--offload-arch=gfx950", "-DNX_BUILT_FOR_GFX950=1", "-S", "--cuda-device-only"] + [a for a in sys.argv[1:] if a != "--all"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names) + "\n", capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\((?:[^()]|\([^()]*\))*\)\s*$", "", o).replace("void ", "").replace("nxd::", "").replace("(anonymous namespace)::", "") for o in out]


print("%-58s %5s %5s %5s %7s %8s %6s %5s" % ("kernel", "VGPR", "AGPR", "SGPR", "spilled", "scratch", "LDS", "occ"))
for src in sorted(glob.glob(os.path.join(ROOT, "nexus_amd/csrc/device/*.hip"))):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", tmp.name, src], cwd=ROOT, capture_output=True, text=True)
        if r.returncode != 0:
            print("# %s: %s" % (os.path.basename(src), r.stderr.strip().split("\n")[-1]))
            continue
        text = open(tmp.name).read()
    rows = []
    for m in re.finditer(r"- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size", text, re.S):
        blk = m.group(0)
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "0"])[1]
        rows.append((g("name"), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    occ = dict(re.findall(r"; Occupancy: (\d+)\n(?:.*\n){0,12}?\s*\.section\s+\.AMDGPU\.csdata.*\n(?:.*\n){0,3}?\s*\.text\n\s*\.protected\s+(\S+)", text))
    occ_by_kernel = {}
    for m in re.finditer(r"^(\S+):\s+; @\1\n(?:.*\n)*?; Occupancy: (\d+)", text, re.M):
        occ_by_kernel[m.group(1)] = m.group(2)
    names = demangle([r[0] for r in rows])
    for (raw, v, a, s, sp, scr, lds), n in zip(rows, names):
        if "rocprim" in n and "--all" not in sys.argv:
            continue  # the sort library's kernels (device LBVH build)
        print("%-58s %5s %5s %5s %7s %8s %6s %5s" % (n[:58], v, a, s, sp, scr, lds, occ_by_kernel.get(raw, "?")))
