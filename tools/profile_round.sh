#!/bin/bash
# One round's committed measurement evidence for a bench command, on the GPU box:
#   tools/profile_round.sh <tag> <config> <frames = warmup + steps> <bench args...>
# writes gpurun_out/prof_<tag>/: the bench line, the rocprofv3 --kernel-trace --stats summary, and separate --pmc passes
# (never combined with other trace domains; each counter set fits one hardware pass: SQ <= 8, TCC <= 4 slots with
# FETCH_SIZE = 3 and WRITE_SIZE = 2, GRBM independent), then the per-ray constants (tools/pmc_to_json.py).
tag=$1; cfg=$2; frames=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -d "$root" ] || { echo "no repo root"; exit 2; }
cd /tmp && export TMPDIR=/tmp && cd "$root"
# bench.py asks for 24 hardware queues with os.environ.setdefault "before the process first touches HIP" — under rocprofv3 the
# profiler's preloaded library has initialised the GPU before Python starts, so the setting has to be in the environment already
# (otherwise profiled runs with passes in flight get the runtime's 4 queues and do not show the schedule the plain bench measures)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24}
O=gpurun_out/prof_$tag; mkdir -p $O
timeout -k 10 600 python bench.py --config $cfg "$@" > $O/bench_line.json 2> $O/bench_line.err || { tail -5 $O/bench_line.err; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --config $cfg "$@" --reps 1 --no-cpu-baseline --no-obj-check --no-reference-mode > $O/bench_under_rocprof.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python tools/level_timeline.py $O/stats 2 > $O/pass_timeline.txt 2>/dev/null
i=0
dirs=""
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc$i -- python bench.py --config $cfg "$@" --reps 1 --no-cpu-baseline --no-obj-check --no-roofline --no-reference-mode > $O/pmc$i.json 2> $O/pmc$i.err
  rc=$?; echo "pmc pass $i ($ctrs) rc=$rc"
  [ $rc -ne 0 ] && { tail -5 $O/pmc$i.err; exit 1; }
  python tools/pmc_sum.py $O/pmc$i > $O/pmc$i.summary.txt
  dirs="$dirs $O/pmc$i"
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
LIST
python tools/pmc_to_json.py --config $cfg --frames $frames --bench $O/bench_line.json --tag $tag --command "python bench.py --config $cfg $* --reps 1" --out $O/trace_counters.json $dirs > $O/per_ray.txt
cat $O/pmc*.summary.txt > $O/pmc_summary.txt
find $O -name "*.csv" -size +3M -delete
cat $O/per_ray.txt | head -60
