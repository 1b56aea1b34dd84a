# development aid (round 4): phase stamps of the boxer's k_fused (library: scripts/dev_build.sh 0x20 -DRMPC_STAMPS)
set -e
mkdir -p gpurun_out
export RMPC_ALLOW_STALE=1
RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so timeout -k 10 300 python scripts/fused_stamps.py cfg3 4096 > gpurun_out/r04_cfg3_stamps.txt 2>&1 || { tail gpurun_out/r04_cfg3_stamps.txt; exit 1; }
grep -v amdgpu gpurun_out/r04_cfg3_stamps.txt
