#!/bin/bash
# Build the library with extra -D flags into a scratch copy and bench it: tools/variant_bench.sh "<flags>" <tag> [bench args]
flags=$1; tag=$2; shift; shift
make clean >/dev/null 2>&1
make -j16 COMMON="-O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Inexus_amd/csrc/device -Inexus_amd/csrc/host -Wall -Wno-unused-function $flags" >/dev/null 2>gpurun_out/build_$tag.err || { echo "build failed $tag"; tail -5 gpurun_out/build_$tag.err; exit 1; }
timeout -k 10 120 python -m pytest tests/test_gpu_trace.py -m gpu -x -q 2>&1 | tail -1
timeout -k 10 200 python bench.py "$@" --no-cpu-baseline > gpurun_out/v_$tag.json 2>gpurun_out/v_$tag.err
python -c "
import json;d=json.load(open('gpurun_out/v_$tag.json'));r=d['roofline'];print('$tag', d['value'], d['ms_per_step'], r['mrays_per_s'], r['kernel_ms_per_frame'])"
