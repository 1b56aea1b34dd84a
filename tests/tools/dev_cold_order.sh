#!/bin/bash
# development aid: cold fused launches in index order (RMPC_NO_COLD_ORDER=1) against the start-state difficulty order
mkdir -p gpurun_out
for rep in 1 2; do
for m in 1 0; do
  if [ $m = 1 ]; then export RMPC_NO_COLD_ORDER=1; echo "index order"; else unset RMPC_NO_COLD_ORDER; echo "difficulty order"; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline 2>/dev/null > gpurun_out/co_$m.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/co_$m.json").read().strip().splitlines()[-1])
print("  bench: value %.4g single %.4g" % (d["value"], d["value_single_call"]), [round(x, 2) for x in d["single_call_ms"]],
      "roofline %.3f avg %.3f fullchip %.3f" % (d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["full_chip_frac"]))
PY
done
done
