#!/bin/bash
# The committed bench lines of a round's final build (profiles/<prefix>_*.json): run on the GPU box, copy from gpurun_out/final/.
O=gpurun_out/final; mkdir -p $O
b() { out=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$out.json 2> $O/$out.err || { echo "FAILED $out"; tail -3 $O/$out.err; }; python -c "import json,sys; d=json.load(open('$O/$out.json')); print('$out', d['value'], d['config'].get('rep_ms'), (d.get('emulated_rank') or {}).get('per_rank_ms'))"; }
b driver_1 --gpus 1 --steps 20 --warmup 5
b default
b default_r1 --passes-in-flight 1 --no-cpu-baseline
b cfg5 --config 5 --no-cpu-baseline
b cfg4 --config 4 --no-cpu-baseline
b s1 --frames-per-pass 1 --no-cpu-baseline --no-reference-mode
for n in 2 4 8; do b emulate_$n --steps 20 --warmup 5 --reps 7 --scaling strong --emulate-rank-of $n --no-cpu-baseline --no-reference-mode --no-roofline; done
for n in 2 4 8; do b emulate_weak_$n --steps 20 --warmup 5 --reps 5 --scaling weak --emulate-rank-of $n --no-cpu-baseline --no-reference-mode --no-roofline; done
