#!/bin/bash
# A/B/C... comparison of kernel variants under drifting clocks: every variant is built into its own library, then the
# variants are benched round-robin ROUNDS times and the per-variant medians are printed.
# usage: ROUNDS=3 ARGS="--steps 64 --warmup 32" tools/ab_bench.sh tagA="-DX=1" tagB="" ...
ROUNDS=${ROUNDS:-3}
ARGS=${ARGS:---steps 64 --warmup 32}
BASEFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Inexus_amd/csrc/device -Inexus_amd/csrc/host -Wall -Wno-unused-function"
mkdir -p gpurun_out/ab build/variants
tags=()
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  tags+=($tag)
  devextra=""   # empty: the Makefile's own default for device files (it probes the -mllvm scheduler options before using them)
  if [ "${flags:0:1}" = "@" ]; then devextra="${flags:1}"; flags=""; fi   # "@..." = flags for the device files only (replaces the default)
  make -j16 OUT=build/variants/lib_$tag.so OBJDIR=build/obj_$tag COMMON="$BASEFLAGS $flags" ${devextra:+DEVEXTRA="$devextra"} >/dev/null 2>gpurun_out/ab/build_$tag.err || { echo "build failed: $tag"; tail -5 gpurun_out/ab/build_$tag.err; exit 1; }
  NEXUS_AMD_LIB=build/variants/lib_$tag.so timeout -k 10 120 python -m pytest tests/test_gpu_trace.py -m gpu -x -q 2>&1 | tail -1
done
for r in $(seq 1 $ROUNDS); do
  for tag in "${tags[@]}"; do
    NEXUS_AMD_LIB=build/variants/lib_$tag.so timeout -k 10 200 python bench.py $ARGS --no-cpu-baseline --no-obj-check > gpurun_out/ab/${tag}_$r.json 2>gpurun_out/ab/${tag}_$r.err || { echo "bench failed: $tag"; tail -3 gpurun_out/ab/${tag}_$r.err; exit 1; }
  done
done
python - "$ROUNDS" "${tags[@]}" <<'PY'
import json, sys, statistics as st
rounds = int(sys.argv[1])
for tag in sys.argv[2:]:
    v, tr, sh = [], [], []
    for r in range(1, rounds + 1):
        d = json.load(open("gpurun_out/ab/%s_%d.json" % (tag, r)))
        v.append(d["value"]); k = d["roofline"]["kernel_ms_per_frame"]; tr.append(k["trace"]); sh.append(k["shadow"])
    print("%-14s median %8.1f  all %s  trace %.4f shadow %.4f" % (tag, st.median(v), [round(x) for x in v], st.median(tr), st.median(sh)))
PY
