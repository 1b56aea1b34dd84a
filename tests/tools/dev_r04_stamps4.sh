# development aid (round 4): phase stamps of the point robot's k_fused at 4096 and 128 instances
mkdir -p gpurun_out
export RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so
timeout -k 10 300 python scripts/fused_stamps.py cfg2 4096 2>&1 | grep -v amdgpu && timeout -k 10 300 python scripts/fused_stamps.py cfg2 128 2>&1 | grep -v amdgpu
