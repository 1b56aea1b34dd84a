# development aid (round 4): phase stamps of k_fused_arm at 128 .. 1024 wavefronts (where the loaded pass loses against the lone one)
set -e
mkdir -p gpurun_out
export RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so
for B in 128 256 512 1024; do timeout -k 10 200 python tests/tools/dev_arm_fused_stamps.py cfg4 $B 2>&1 | grep -v amdgpu; done > gpurun_out/r04_arm_load.txt
cat gpurun_out/r04_arm_load.txt
