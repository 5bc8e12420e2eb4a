#!/bin/bash
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24}  # (see tools/profile_round.sh)
# rocprofv3 --pmc passes from a list file (one pass per line); usage: tools/pmc_list.sh <outdir> <listfile> [bench args]
out=$1; list=$2; shift; shift
mkdir -p $out
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp && export TMPDIR=/tmp && cd "$root"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $line --output-format csv -d $out/pass$i -- python bench.py "$@" --no-cpu-baseline --no-roofline --no-reference-mode > $out/pass$i.json 2> $out/pass$i.err
  rc=$?
  echo "pass$i ($line) exit=$rc"
  [ $rc -ne 0 ] && exit $rc
  python tools/pmc_sum.py $out/pass$i > $out/pass$i.summary.txt
  find $out/pass$i -name "*.csv" -size +2M -delete
done < $list
exit 0
