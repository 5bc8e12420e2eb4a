#!/bin/bash
# one rocprofv3 --pmc pass over a short bench run; usage: tools/pmc_one.sh <outdir> "<counters>" [bench args]
out=$1; ctrs=$2; shift; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python bench.py "$@" --no-cpu-baseline --no-roofline > $out/bench.json 2> $out/bench.err
echo "exit=$?"
