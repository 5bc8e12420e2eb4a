cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for S in 1 2 4 8 16 32 64; do python bench.py --no-cpu-baseline --no-roofline --frames-per-pass $S --steps 128 --warmup 64 > gpurun_out/b_s$S.json; python -c "
import json;d=json.load(open('gpurun_out/b_s$S.json'));print('S=$S', d['value'], d['ms_per_step'])"; done
rm -rf gpurun_out/tl/cur; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/cur -- python bench.py --steps 96 --warmup 32 --no-cpu-baseline --no-roofline > gpurun_out/tl/cur.json 2>gpurun_out/tl/cur.err
python tools/pass_timeline.py gpurun_out/tl/cur 2
rm -rf gpurun_out/tl/cur
