# development aid: phases of the recursion inside k_fused (library: scripts/dev_build.sh <mask> -DRMPC_RIC_STAMPS)
export RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so
CFG=${1:-cfg2}
timeout -k 10 200 python tests/tools/dev_ric_stamps.py $CFG 4096 2>&1 | grep -v amdgpu || exit 1
timeout -k 10 200 python tests/tools/dev_ric_stamps.py $CFG 128 2>&1 | grep -v amdgpu || exit 1
