set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/bench_20_5.json 2> gpurun_out/bench_20_5.err
python bench.py --steps 37 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/bench_37_3.json 2> gpurun_out/bench_37_3.err
export NX_BENCH_BACKEND=gloo NX_BENCH_SHARE_GPU=1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 40 --warmup 8 --png gpurun_out/bench_2rank.png > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
unset NX_BENCH_BACKEND NX_BENCH_SHARE_GPU
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --png gpurun_out/bench_1rank.png > gpurun_out/bench_1rank.json 2> gpurun_out/bench_1rank.err
python tools/cmp_png.py gpurun_out/bench_1rank.png gpurun_out/bench_2rank.png
for f in default 20_5 37_3 2rank 1rank; do python -c "
import json;d=json.loads(open('gpurun_out/bench_$f.json').read().strip().splitlines()[-1]);print('$f', d['value'], d['ms_per_step'], d['steps'], d['n_gpus'], d.get('cpu_baseline'))"; done
