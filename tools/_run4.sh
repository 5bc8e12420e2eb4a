cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01c
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01c/stats -- python bench.py > gpurun_out/r01c/bench_under_rocprof.json 2> gpurun_out/r01c/stats.err || exit 1
find gpurun_out/r01c/stats -name "*kernel_trace.csv" -size +8M -delete
printf 'FETCH_SIZE\nWRITE_SIZE\n' > /tmp/l.list
bash tools/pmc_list.sh gpurun_out/r01c/pmc /tmp/l.list --steps 32 --warmup 32
python bench.py > gpurun_out/r01c/bench.json 2> gpurun_out/r01c/bench.err
