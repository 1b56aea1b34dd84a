#!/bin/bash
# development aid: the bench's timed region with 4 / 6 / 8 solver handles in flight, driver protocol and 200 steps
mkdir -p gpurun_out
for S in 4 6 8; do
  for K in 20 200; do
    timeout -k 10 300 python bench.py --steps $K --warmup 5 --streams $S --no-cpu-baseline --no-kernel-events --cfg5-steps 20 > gpurun_out/st_${S}_${K}.json 2>/dev/null || exit 1
    python - <<PY
import json
d = json.load(open("gpurun_out/st_${S}_${K}.json"))
print("streams $S steps $K: cfg2 %.3f M  cfg4 %.3f M  cfg3 %.3f M" % (d["value"] / 1e6, d["legs"]["cfg4"]["value"] / 1e6, d["legs"]["cfg3"]["value"] / 1e6))
PY
  done
done
