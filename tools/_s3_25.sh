#!/bin/bash
O=gpurun_out/s3_25; mkdir -p $O
for rep in 1 2; do
for n in 8 4; do
for b in -1 3 4 5 0; do
  env NX_TUNING_KNOBS=1 NX_TAIL_BOUNCE=$b timeout -k 10 200 python bench.py --steps 20 --warmup 5 --reps 5 --emulate-rank-of $n --no-cpu-baseline --no-obj-check --no-reference-mode --no-roofline > $O/n${n}_b${b}_$rep.json 2> $O/n${n}_b${b}_$rep.err || { tail -3 $O/n${n}_b${b}_$rep.err; exit 1; }
  python - $O/n${n}_b${b}_$rep.json $n $b <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); e=d["emulated_rank"]
print("rank of %s, tail from bounce %s: %.3f ms (reps %s); full frame %.1f" % (sys.argv[2], sys.argv[3], e["per_rank_ms"], e["rep_ms"], d["value"]))
PY
done; done; done
