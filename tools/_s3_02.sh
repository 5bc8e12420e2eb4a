#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s3_02; mkdir -p $O
for k in 20 64; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$k -- python bench.py --steps $k --warmup $k --passes-in-flight 1 --reps 2 --no-cpu-baseline --no-obj-check --no-reference-mode --no-roofline > $O/b$k.json 2> $O/b$k.err || { tail -3 $O/b$k.err; exit 1; }
  python tools/level_timeline.py $O/t$k 2 > $O/timeline$k.txt 2>&1
done
find $O -name "*.csv" -size +3M -delete
cat $O/timeline20.txt | tail -70
echo ======
cat $O/timeline64.txt | tail -70
