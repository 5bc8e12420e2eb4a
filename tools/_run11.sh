timeout -k 10 300 python tools/latency_probe.py || exit 1
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
ROUNDS=3 ARGS="--steps 64 --warmup 32" tools/ab_bench.sh den8="" den4="-DNX_POOL_DEN=4" den16="-DNX_POOL_DEN=16" den8r128="-DNX_RESERVE=128" den8r512="-DNX_RESERVE=512"
for S in 1 4; do python bench.py --no-cpu-baseline --no-roofline --frames-per-pass $S --steps 64 --warmup 8 > gpurun_out/b_s$S.json; python -c "
import json;d=json.load(open('gpurun_out/b_s$S.json'));print('S=$S', d['value'], d['ms_per_step'])"; done
