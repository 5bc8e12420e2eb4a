# Builds libnexus_amd.so (HIP device layer for gfx950 + C++ host classes + flat C API) in-tree.
# Plain make + hipcc: no cmake needed.  `make oracle` builds the CPU oracle (test infrastructure only).
HIPCC     ?= /opt/rocm/bin/hipcc
ARCH      ?= gfx950
OUT       := nexus_amd/lib/libnexus_amd.so
OBJDIR    := build/obj
# DEVEXTRA -fno-slp-vectorize (device files only): the SLP vectorizer pairs fp32 operations into v_pk_fma_f32 / v_pk_mul_f32; on gfx950 those are not
# double-rate and the register shuffling they need costs more than they save (trace kernel: +9 %, 10 fewer VGPRs)
COMMON    := -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Inexus_amd/csrc/device -Inexus_amd/csrc/host -Wall -Wno-unused-function
# -mllvm -amdgpu-use-amdgpu-trackers=1: the machine scheduler follows register pressure with the AMDGPU-specific trackers instead of
# the generic ones — the material kernels spill 12-14 VGPRs instead of 18-21 at their 96-register budget (-10 % on them), the trace
# kernels 3 / 0 instead of 7 / 1; -amdgpu-sched-strategy=max-memory-clause on top of it: another 1 % (round 4: driver command 2 021 ->
# 2 091, default 2 308 -> 2 397, configs[4] 1 263 -> 1 293; images bit-identical: scheduling reorders no arithmetic)
# The two -mllvm options are internal LLVM cl::opts (present in ROCm 7.2's hipcc): a toolchain without them rejects the whole
# command line ("Unknown command line argument"), so they are probed once — an empty translation unit compiled with them — and
# left out when the probe fails (the build then still succeeds, the kernels spill a little more: see above).
SCHEDFLAGS := -mllvm -amdgpu-use-amdgpu-trackers=1 -mllvm -amdgpu-sched-strategy=max-memory-clause
# The probe runs only when device code is going to be compiled (not for `make clean` / `make oracle`) and only when the caller has
# not given DEVEXTRA itself (tools/ab_bench.sh does); whether the flags went in is compiled into the library (NX_SCHED_FLAGS ->
# nxhip_build_info() bit 1, printed by bench.py as config.build), so a run built without them cannot be mistaken for one with them.
NX_DEVICE_GOALS := $(filter-out clean oracle,$(or $(MAKECMDGOALS),all))
SCHED_DEF :=
ifeq ($(origin DEVEXTRA),undefined)
ifneq ($(NX_DEVICE_GOALS),)
SCHED_OK   := $(shell echo '__global__ void k(){}' | $(HIPCC) -x hip --offload-arch=$(ARCH) $(SCHEDFLAGS) -c -o /dev/null - >/dev/null 2>&1 && echo yes)
ifneq ($(SCHED_OK),yes)
$(warning hipcc does not accept "$(SCHEDFLAGS)": building the device code without them (material kernels spill 18-21 VGPRs instead of 12-14))
SCHEDFLAGS :=
else
SCHED_DEF := -DNX_SCHED_FLAGS=1
endif
endif
DEVEXTRA  := -fno-slp-vectorize $(SCHEDFLAGS)
else
SCHED_DEF := $(if $(findstring amdgpu-sched-strategy,$(DEVEXTRA)),-DNX_SCHED_FLAGS=1,)
endif
# `make release`: the library without the nxhip_debug_* test hooks (NX_NO_DEBUG_HOOKS; the tests need the default build)
HOOKS_DEF ?=
DEVFLAGS  := $(COMMON) $(DEVEXTRA) --offload-arch=$(ARCH) -DNX_BUILT_FOR_GFX950=1 $(SCHED_DEF) $(HOOKS_DEF)
HOSTFLAGS := $(COMMON)

DEV_SRCS  := $(wildcard nexus_amd/csrc/device/*.hip)
HOST_SRCS := $(wildcard nexus_amd/csrc/host/*.cpp) $(wildcard nexus_amd/csrc/capi/*.cpp)
DEV_OBJS  := $(patsubst %.hip,$(OBJDIR)/%.o,$(DEV_SRCS))
HOST_OBJS := $(patsubst %.cpp,$(OBJDIR)/%.o,$(HOST_SRCS))
HDRS      := $(wildcard include/*.h include/nexus/*.h nexus_amd/csrc/device/*.h nexus_amd/csrc/host/*.h)

all: $(OUT)

$(OBJDIR)/%.o: %.hip $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(DEVFLAGS) -c $< -o $@

$(OBJDIR)/%.o: %.cpp $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HOSTFLAGS) -c $< -o $@

$(OUT): $(DEV_OBJS) $(HOST_OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(LDEXTRA) -o $@ $^ -lpthread -lz

oracle:
	$(MAKE) -C oracle liboracle.so

# The library a product links: the same code with the nxhip_debug_* test hooks refusing (nexus_amd/lib/release/; objects of its own)
release:
	$(MAKE) OUT=nexus_amd/lib/release/libnexus_amd.so OBJDIR=build/release/obj HOOKS_DEF=-DNX_NO_DEBUG_HOOKS=1 all

clean:
	rm -rf build $(OUT)

.PHONY: all oracle clean release

# Host-side AddressSanitizer + UBSan build (device code is compiled as usual; GPU ASan is not available on this pool).
# Used by tools/run_sanitized_cpu_tests.sh; output goes to build/asan/ and never replaces the product library.
ASAN_FLAGS := -fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -g -fno-omit-frame-pointer
asan:
	$(MAKE) OUT=build/asan/libnexus_amd.so OBJDIR=build/asan/obj COMMON="$(COMMON) $(ASAN_FLAGS)" LDEXTRA="-fsanitize=address,undefined -shared-libsan"
	$(MAKE) -C oracle clean
	$(MAKE) -C oracle liboracle.so CC=/opt/rocm/lib/llvm/bin/clang SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libsan -g"

.PHONY: asan

# The reference's render loop through the kept C++ API, headless (examples/nexus_render.cpp)
example: $(OUT)
	@mkdir -p build
	$(HIPCC) -O2 -std=c++17 -Iinclude examples/nexus_render.cpp -o build/nexus_render -Lnexus_amd/lib -lnexus_amd -Wl,-rpath,'$$ORIGIN/../nexus_amd/lib'

.PHONY: example

# configs[1] through the kept C++ API (examples/nexus_bench.cpp): beside the library, so that it travels to the GPU box with it
bench_example: $(OUT)
	$(HIPCC) -O2 -std=c++17 -Iinclude examples/nexus_bench.cpp -o nexus_amd/lib/nexus_bench -Lnexus_amd/lib -lnexus_amd -Wl,-rpath,'$$ORIGIN'

.PHONY: bench_example
