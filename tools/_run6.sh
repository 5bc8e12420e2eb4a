A="--steps 64 --warmup 32"
bash tools/variant_bench.sh "-DNX_RESERVE=64" r64 $A &&
bash tools/variant_bench.sh "-DNX_RESERVE=256" r256 $A &&
bash tools/variant_bench.sh "-DNX_RESERVE=64 -DNX_REFILL_BELOW=48" r64f48 $A &&
bash tools/variant_bench.sh "-DNX_REFILL_BELOW=32" f32 $A &&
bash tools/variant_bench.sh "-DNX_REFILL_BELOW=48" f48 $A
