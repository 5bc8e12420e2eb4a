#!/bin/bash
# Build a variant of the library with extra -D flags into nexus_amd/lib/variants/lib_<tag>.so (own object directory, so that no
# object of another flag set can be linked in; nxhip_create would refuse such a mix anyway: nx_device.h layout_stamp).
#   tools/build_variant.sh <tag> "<extra flags>"        then on the GPU box:  tools/ab_prebuilt.sh main <tag> ...
set -e
tag=$1; flags=$2
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Inexus_amd/csrc/device -Inexus_amd/csrc/host -Wall -Wno-unused-function"
mkdir -p nexus_amd/lib/variants
rm -rf build/v_$tag
make -j8 OUT=nexus_amd/lib/variants/lib_$tag.so OBJDIR=build/v_$tag COMMON="$BASE $flags" 2>&1 | grep -E "error|Error" && exit 1
ls -la nexus_amd/lib/variants/lib_$tag.so
