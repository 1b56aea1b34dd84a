#!/bin/bash
# development aid (round 3, review item 1a): the lane-per-instance recursion of the pass kernels against the
# wave-per-instance (LDS) one -- parity against the oracle, then per-kernel times of the first 4 passes of a solve
# (every instance still iterating in every launch) at three batch sizes
export RMPC_NO_FUSED=1
RMPC_RIC_LANE=2 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "test_solve_matches_oracle or test_solve_matches_golden or full_size" 2>&1 | tail -3
for B in 1024 4096 16384 65536; do
  for L in 0 2; do
    echo "B=$B RMPC_RIC_LANE=$L"; DEV_BUDGET=4 RMPC_RIC_LANE=$L timeout -k 10 300 python tests/tools/dev_time.py cfg2 $B 3 2>&1 | grep -v amdgpu.ids
  done
done
