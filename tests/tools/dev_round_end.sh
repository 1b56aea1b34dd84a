#!/bin/bash
# development aid, on the GPU box: everything profiles/ is made from, of ONE library -- rocprofv3 kernel traces and
# counter passes of the three configurations, the bench line in the driver's protocol and with the defaults, the phase
# stamps of k_fused (csrc/librmpc_hip_stamps.so: scripts/dev_build.sh 0x25 -DRMPC_STAMPS).  Afterwards, here:
#   for c in cfg2 cfg3 cfg4; do python scripts/summarize_profiles.py rNN $c; done; copy the bench lines and the stamps
TAG=${1:-r04}
for c in cfg2 cfg3 cfg4; do timeout -k 10 380 bash scripts/collect_profiles.sh $TAG $c > gpurun_out/collect_$c.log 2>&1 || exit 1; done
timeout -k 10 900 tests/tools/dev_bench_round.sh $TAG || exit 1
timeout -k 10 300 scripts/stamps_round.sh > gpurun_out/${TAG}_stamps.txt 2> gpurun_out/${TAG}_stamps.err
