A="--steps 32 --warmup 32 --frames-per-pass 32"
bash tools/variant_bench.sh "" base $A &&
bash tools/variant_bench.sh "-DNX_POSTPONE_TRI=8" pt8 $A &&
bash tools/variant_bench.sh "-DNX_POSTPONE_TRI=16" pt16 $A &&
bash tools/variant_bench.sh "-DNX_POSTPONE_TRI=24" pt24 $A &&
bash tools/variant_bench.sh "-DNX_POSTPONE_TRI=16 -DNX_POSTPONE_NODE=16" pt16n16 $A &&
bash tools/variant_bench.sh "-DNX_POSTPONE_TRI=64 -DNX_POSTPONE_NODE=64" major $A
