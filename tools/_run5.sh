set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01d/stats -- python bench.py > gpurun_out/r01d/bench_under_rocprof.json 2> gpurun_out/r01d/stats.err
find gpurun_out/r01d/stats -name "*kernel_trace.csv" -size +8M -delete
python bench.py > gpurun_out/r01d/bench.json 2> gpurun_out/r01d/bench.err
