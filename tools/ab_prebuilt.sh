#!/bin/bash
# Round-robin A/B of library variants built beforehand (nexus_amd/lib/variants/lib_<tag>.so, or "main" = the product library):
#   ROUNDS=3 ARGS="--steps 256 --warmup 64" OUT=gpurun_out/ab1 tools/ab_prebuilt.sh main r2 nostall ...
# Prints per-variant medians of Msamples/s and of the closest-hit / any-hit trace time per frame.
ROUNDS=${ROUNDS:-3}
ARGS=${ARGS:---steps 256 --warmup 64}
OUT=${OUT:-gpurun_out/ab}
mkdir -p $OUT
# A tag may carry environment settings for the run: "main+NX_ANY_FIRST=1" = the product library with that variable (and
# NX_TUNING_KNOBS=1) set; files are named after the whole tag.
lib_of() { local t=${1%%+*}; if [ "$t" = main ]; then echo nexus_amd/lib/libnexus_amd.so; else echo nexus_amd/lib/variants/lib_$t.so; fi; }
env_of() { case "$1" in *+*) echo "NX_TUNING_KNOBS=1 ${1#*+}" | tr '+' ' ';; esac; }
for r in $(seq 1 $ROUNDS); do
  for tag in "$@"; do
    env $(env_of $tag) NEXUS_AMD_LIB=$(lib_of $tag) timeout -k 10 200 python bench.py $ARGS --no-cpu-baseline --no-obj-check > $OUT/${tag}_$r.json 2> $OUT/${tag}_$r.err || { echo "bench failed: $tag"; tail -3 $OUT/${tag}_$r.err; exit 1; }
  done
done
python - "$OUT" "$ROUNDS" "$@" <<'PY'
import json, sys, statistics as st
out, rounds = sys.argv[1], int(sys.argv[2])
for tag in sys.argv[3:]:
    v, tr, sh, sd, lg = [], [], [], [], []
    for r in range(1, rounds + 1):
        d = json.load(open("%s/%s_%d.json" % (out, tag, r)))
        v.append(d["value"])
        k = d.get("roofline", {}).get("kernel_ms_per_frame")
        if k:
            tr.append(k["trace"]); sh.append(k["shadow"]); sd.append(k["shade"]); lg.append(k["logic"])
    extra = " trace %.4f shadow %.4f shade %.4f logic %.4f" % (st.median(tr), st.median(sh), st.median(sd), st.median(lg)) if tr else ""
    print("%-14s median %8.1f  all %s%s" % (tag, st.median(v), [round(x) for x in v], extra))
PY
