# development aid (round 4): phase stamps of k_fused for one configuration (library built with -DRMPC_STAMPS)
set -e
mkdir -p gpurun_out
export RMPC_ALLOW_STALE=1
RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so timeout -k 10 200 python tests/tools/dev_arm_fused_stamps.py cfg4 1024 > gpurun_out/r04_armf_1024.txt 2>&1 || { tail gpurun_out/r04_armf_1024.txt; exit 1; }
RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so timeout -k 10 200 python tests/tools/dev_arm_fused_stamps.py cfg4 64 > gpurun_out/r04_armf_64.txt 2>&1
cat gpurun_out/r04_armf_1024.txt gpurun_out/r04_armf_64.txt
