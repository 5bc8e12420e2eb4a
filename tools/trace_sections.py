"""Static instruction counts of the trace kernel's loop sections, from the ISA of its counting variant (trace_kernel<closest, STATS>):
the variant reads the shader clock at the section boundaries (NX_STAMP in nx_trace.hip -> s_memtime), so the instructions between two
consecutive clock reads in the code layout are one section's.  Beside the measured cycle shares (bench.py roofline.simd.cycle_share)
this says whether a section's share of the wave's time is instructions or waiting (VERDICT r3 item 2 i).
    python tools/trace_sections.py [extra -D flags] > profiles/r04_trace_sections.txt
Sections, in source order: refill | pop / retire | record select + fetch issue | instance entry | node decode (child_trace) |
triangle test | rest of the iteration (retire test, stall guard, loop condition).  Basic blocks the compiler moved out of line (the
reservation path of the refill, scratch-stack spills) are attributed to the section whose clock read precedes them in the layout, so
the figures are approximate for `refill` and exact enough for the others."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Iinclude", "-Inexus_amd/csrc/device", "-Inexus_amd/csrc/host", "-fno-slp-vectorize", "-mllvm", "-amdgpu-use-amdgpu-trackers=1", "-mllvm", "-amdgpu-sched-strategy=max-memory-clause",
         "--offload-arch=gfx950", "-DNX_BUILT_FOR_GFX950=1", "-S", "--cuda-device-only"] + sys.argv[1:]

with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", tmp.name, os.path.join(ROOT, "nexus_amd/csrc/device/nx_trace.hip")], cwd=ROOT, check=True, capture_output=True)
    text = open(tmp.name).read()


def body(mangled_part):
    m = re.search(r"^(_ZN3nxd12trace_kernel%s\S*):\s*; @" % mangled_part, text, re.M)
    start = m.end()
    end = text.index(".Lfunc_end", start)
    return [l.strip() for l in text[start:end].split("\n")]


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    return "other"


def count(lines):
    sections, cur = [], {}
    for l in lines:
        if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
            continue
        op = l.split()[0]
        if op.startswith("s_memtime") or op.startswith("s_memrealtime"):
            sections.append(cur)
            cur = {}
            continue
        k = classify(op)
        cur[k] = cur.get(k, 0) + 1
    sections.append(cur)
    return sections


# the two (and a half) classes of wave64 VALU instructions on gfx950 (tools/micro/valu_classes.hip -> profiles/r04_valu_opcode_classes.txt:
# 2.2 cycles of issue for every instruction; the second class also occupies a unit for 4.1 cycles, the third for 8.1, beside the issue slot)
FAST = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_mul_legacy_f32", "v_not_b32", "v_add_co_u32", "v_addc_co_u32")
VERY_SLOW = ("v_rcp_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_pk_fma_f32", "v_mov_b64", "v_lshl_add_u64", "v_mul_lo_u32", "v_mul_hi_u32")


def valu_classes(lines):
    f = s_ = v = 0
    for l in lines:
        if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
            continue
        op = l.split()[0]
        if not op.startswith("v_"):
            continue
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        if base in FAST:
            f += 1
        elif base in VERY_SLOW:
            v += 1
        else:
            s_ += 1
    return f, s_, v


KEYS = ["valu", "salu", "lds", "vmem", "scratch", "wait", "branch"]
for label, part in (("closest hit, counting variant (STATS): sections between clock reads, in code layout order", "ILb0ELb1EEE"), ("closest hit, product kernel: whole kernel", "ILb0ELb0EEE"),
                    ("any hit, product kernel: whole kernel", "ILb1ELb0EEE")):
    secs = count(body(part))
    print(label)
    print("  %-9s " % "section" + " ".join("%8s" % k for k in KEYS) + "   total")
    for i, s in enumerate(secs):
        if len(secs) == 1:
            name = "all"
        else:
            name = "pre" if i == 0 else "s%d" % i
        print("  %-9s " % name + " ".join("%8d" % s.get(k, 0) for k in KEYS) + "   %5d" % sum(s.values()))
    f, sl, vs = valu_classes(body(part))
    print("  VALU classes: %d full-rate, %d second-unit (4.1 cycles), %d reciprocal-class (8.1) -> issue %d cycles (2.2 per instruction), second unit %d cycles per pass through this code"
          % (f, sl, vs, round(2.2 * (f + sl + vs)), round(4.1 * sl + 8.1 * vs)))
    print()
