# development aid: k_fused_arm with three parts per stage (default at N <= 21) against two (RMPC_ARM_TWO_PARTS=1): parity, timing
mkdir -p gpurun_out
export RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so
for c in "cfg4 128 1000" $EXTRA; do
  echo "== $c"
  timeout -k 10 300 python tests/tools/dev_vs_oracle.py $c > gpurun_out/r04_vs_one.log 2>&1 || { tail -5 gpurun_out/r04_vs_one.log; exit 1; }
  tail -1 gpurun_out/r04_vs_one.log
done
for rep in 1 2; do
echo "== three parts"; timeout -k 10 300 python tests/tools/quick_time.py cfg4 2>&1 | grep -v amdgpu || exit 1
echo "== two parts"; RMPC_ARM_TWO_PARTS=1 timeout -k 10 300 python tests/tools/quick_time.py cfg4 2>&1 | grep -v amdgpu || exit 1
done
