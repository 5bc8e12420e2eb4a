#!/bin/bash
# CPU test suite under AddressSanitizer + UBSan: host classes / C-ABI host side of libnexus_amd.so and the oracle, both
# built with clang's sanitizers (GPU ASan is not available on this pool: device code is compiled as usual).
# Restores the normal oracle build afterwards.
set -e
cd "$(dirname "$0")/.."
make asan -j8 >/dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
rc=0
NEXUS_AMD_LIB=$PWD/build/asan/libnexus_amd.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@" || rc=$?
make -C oracle clean >/dev/null && make -C oracle liboracle.so >/dev/null
exit $rc
