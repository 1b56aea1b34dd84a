# development aid: A/B timing of two development libraries (csrc/librmpc_hip_dev.so vs csrc/librmpc_hip_dev2.so), alternating
CFG=${1:-cfg4}
mkdir -p gpurun_out
export RMPC_ALLOW_STALE=1
for rep in 1 2; do for l in dev dev2; do
  echo "== $l"
  RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_$l.so timeout -k 10 300 python tests/tools/quick_time.py $CFG 2>&1 | grep -v amdgpu || exit 1
done; done
