A="--steps 64 --warmup 32"
bash tools/variant_bench.sh "-DNX_RESERVE=128" a128 $A &&
bash tools/variant_bench.sh "-DNX_RESERVE=256" a256 $A &&
bash tools/variant_bench.sh "-DNX_RESERVE=512" a512 $A &&
bash tools/variant_bench.sh "-DNX_RESERVE=1024" a1024 $A &&
bash tools/variant_bench.sh "-DNX_RESERVE=512 -DNX_REFILL_BELOW=48" a512f48 $A &&
bash tools/variant_bench.sh "-DNX_RESERVE=512 -DNX_REFILL_BELOW=56" a512f56 $A &&
bash tools/variant_bench.sh "-DNX_RESERVE=512 -DNX_REFILL_BELOW=48 -DNX_RESERVE_SHARE=2" a512f48s2 $A
