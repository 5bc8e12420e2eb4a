set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for b in auto 4 5 6 8; do
  if [ $b = auto ]; then unset NX_TRACE_BLOCKS_PER_CU; else export NX_TRACE_BLOCKS_PER_CU=$b; fi
  python bench.py --steps 32 --warmup 32 --frames-per-pass 32 --no-cpu-baseline > gpurun_out/direct_b$b.json 2>gpurun_out/direct_b$b.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/direct_b$b.json").read().strip().splitlines()[-1])
print("$b", d["value"], d["ms_per_step"], d.get("roofline",{}).get("achieved"), d.get("kernel_ms_per_frame"))
PY
done
