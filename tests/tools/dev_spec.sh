#!/bin/bash
# development aid (GPU box): generated views against the runtime tables (plans, flags, iterations), then timing
set -o pipefail
mkdir -p gpurun_out
RMPC_NO_SPEC=1 timeout -k 10 300 python tests/tools/ab_dump.py gpurun_out/ab_rt.npz > gpurun_out/ab_rt.log 2>&1 || { tail -5 gpurun_out/ab_rt.log; exit 1; }
timeout -k 10 300 python tests/tools/ab_dump.py gpurun_out/ab_spec.npz > gpurun_out/ab_spec.log 2>&1 || { tail -5 gpurun_out/ab_spec.log; exit 1; }
python tests/tools/ab_compare.py gpurun_out/ab_rt.npz gpurun_out/ab_spec.npz --tol 1e-8 || exit 1
echo views; timeout -k 10 600 python tests/tools/quick_time.py ${1:-cfg2} 2>&1 | grep -v amdgpu.ids; echo "runtime tables"; RMPC_NO_SPEC=1 timeout -k 10 600 python tests/tools/quick_time.py ${1:-cfg2} 2>&1 | grep -v amdgpu.ids
