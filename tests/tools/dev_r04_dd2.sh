# development aid: boxer variants of the library under test (csrc/librmpc_hip_dev.so, mask 0x31) against the oracle, then timing
mkdir -p gpurun_out
export RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so
for c in "cfg3 256 1000" "cfg3 777 5" "wc_boxer 96 33" "wc_boxer_slack 96 34" "boxer 64 35"; do
  echo "== $c"
  timeout -k 10 300 python tests/tools/dev_vs_oracle.py $c > gpurun_out/r04_vs_one.log 2>&1 || { tail -5 gpurun_out/r04_vs_one.log; exit 1; }
  tail -1 gpurun_out/r04_vs_one.log | cut -c1-260
done
timeout -k 10 300 python tests/tools/quick_time.py cfg3 2>&1 | grep -v amdgpu
