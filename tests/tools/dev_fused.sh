#!/bin/bash
# development aid (GPU box): bitwise A/B of the fused kernel against the pass kernels, timing, phase stamps
set -o pipefail
mkdir -p gpurun_out
RMPC_NO_FUSED=1 python tests/tools/ab_dump.py gpurun_out/ab_pass.npz > gpurun_out/ab_pass.log 2>&1 || { tail -5 gpurun_out/ab_pass.log; exit 1; }
timeout -k 10 300 python tests/tools/ab_dump.py gpurun_out/ab_fused.npz > gpurun_out/ab_fused.log 2>&1 || { tail -5 gpurun_out/ab_fused.log; exit 1; }
python tests/tools/ab_compare.py gpurun_out/ab_pass.npz gpurun_out/ab_fused.npz --tol 1e-8 || exit 1
timeout -k 10 300 python tests/tools/quick_time.py ${1:-cfg2} 2>&1 | grep -v amdgpu.ids
if [ -f robot_mpcs_amd/csrc/librmpc_hip_stamps.so ]; then
  RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_stamps.so timeout -k 10 120 python scripts/fused_stamps.py cfg2 2>&1 | grep -v amdgpu.ids
fi
