ROUNDS=2 ARGS="--steps 64 --warmup 32" tools/ab_bench.sh dyn="-DNX_POOL_DEN=0" dynnoexit="-DNX_POOL_DEN=0 -DNX_NO_EARLY_EXIT=1" den2="-DNX_POOL_DEN=2"
