#!/bin/bash
# Round-robin A/B of prebuilt library variants on the bench's reference-mode figures (ordered / reference beside the headline):
#   ROUNDS=2 OUT=gpurun_out/x tools/ab_ordered.sh main sb512 ...
ROUNDS=${ROUNDS:-2}
ARGS=${ARGS:---steps 20 --warmup 5}
OUT=${OUT:-gpurun_out/abo}
mkdir -p $OUT
lib_of() { if [ "$1" = main ]; then echo nexus_amd/lib/libnexus_amd.so; else echo nexus_amd/lib/variants/lib_$1.so; fi; }
for r in $(seq 1 $ROUNDS); do
  for tag in "$@"; do
    NEXUS_AMD_LIB=$(lib_of $tag) timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --no-obj-check --no-roofline > $OUT/${tag}_$r.json 2> $OUT/${tag}_$r.err || { echo "bench failed: $tag"; tail -3 $OUT/${tag}_$r.err; exit 1; }
  done
done
python - "$OUT" "$ROUNDS" "$@" <<'PY'
import json, sys, statistics as st
out, rounds = sys.argv[1], int(sys.argv[2])
for tag in sys.argv[3:]:
    v, o, rf = [], [], []
    for r in range(1, rounds + 1):
        d = json.load(open("%s/%s_%d.json" % (out, tag, r)))
        v.append(d["value"]); m = d["config"]["reference_mode"]
        o.append(m["ordered"]["value"]); rf.append(m["reference"]["value"])
    print("%-10s headline %7.1f  ordered %7.1f (%.3f of headline)  reference %7.1f   all ordered %s" % (tag, st.median(v), st.median(o), st.median(o) / st.median(v), st.median(rf), [round(x) for x in o]))
PY
