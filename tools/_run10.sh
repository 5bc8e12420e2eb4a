cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BASEFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Inexus_amd/csrc/device -Inexus_amd/csrc/host -Wall -Wno-unused-function"
mkdir -p build/variants gpurun_out/tl
for spec in r64="-DNX_RESERVE=64" r128="-DNX_RESERVE=128" r256="-DNX_RESERVE=256"; do
  tag=${spec%%=*}; flags=${spec#*=}
  make -j16 OUT=build/variants/lib_$tag.so OBJDIR=build/obj_$tag COMMON="$BASEFLAGS $flags" >/dev/null 2>&1 || exit 1
  rm -rf gpurun_out/tl/$tag
  NEXUS_AMD_LIB=build/variants/lib_$tag.so timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/$tag -- python bench.py --steps 96 --warmup 32 --no-cpu-baseline --no-roofline > gpurun_out/tl/$tag.json 2>gpurun_out/tl/$tag.err || exit 1
  echo $tag $(python -c "import json;print(json.load(open('gpurun_out/tl/$tag.json'))['value'])")
  python tools/pass_timeline.py gpurun_out/tl/$tag 2
  python tools/pass_timeline.py gpurun_out/tl/$tag 3
  rm -rf gpurun_out/tl/$tag
done
