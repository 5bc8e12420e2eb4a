#!/bin/bash
# which switch makes two pass schedules of the same frames differ?  (scratch tool)
O=gpurun_out/bisect; mkdir -p $O
common="--steps 12 --warmup 4 --reps 1 --width 320 --height 200 --no-cpu-baseline --no-roofline --no-obj-check --no-reference-mode"
run() { # tag, env, extra
  env $2 python bench.py $common $3 --png $O/$1.png > $O/$1.json 2> $O/$1.err || echo "FAILED $1"
}
for sw in "plain:NX_TUNING_KNOBS=1 NX_NO_THIN=1:--no-entry-points" "thin:X=1:--no-entry-points" "entry:NX_TUNING_KNOBS=1 NX_NO_THIN=1:" "both:X=1:"; do
  tag=${sw%%:*}; rest=${sw#*:}; e=${rest%%:*}; x=${rest#*:}
  run ${tag}_fpp2 "$e" "--frames-per-pass 2 $x"
  run ${tag}_fpp8 "$e" "--frames-per-pass 8 $x"
  run ${tag}_fpp8b "$e" "--frames-per-pass 8 $x"
  cmp -s $O/${tag}_fpp2.png $O/${tag}_fpp8.png && echo "$tag: fpp2 == fpp8" || echo "$tag: fpp2 != fpp8"
  cmp -s $O/${tag}_fpp8.png $O/${tag}_fpp8b.png && echo "$tag: fpp8 repeatable" || echo "$tag: fpp8 NOT repeatable"
done
cmp -s $O/plain_fpp2.png $O/both_fpp2.png && echo "plain == both (fpp2)" || echo "plain != both (fpp2)"
