# development aid: arm library under test (csrc/librmpc_hip_dev.so, mask 0x4 [+ -DRMPC_STAMPS]): parity with the oracle, timing, stamps
mkdir -p gpurun_out
export RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so
for c in "cfg4 128 1000" $EXTRA; do
  echo "== $c"
  timeout -k 10 300 python tests/tools/dev_vs_oracle.py $c > gpurun_out/r04_vs_one.log 2>&1 || { tail -5 gpurun_out/r04_vs_one.log; exit 1; }
  tail -1 gpurun_out/r04_vs_one.log
done
timeout -k 10 300 python tests/tools/quick_time.py cfg4 > gpurun_out/r04_cfg4_qt.log 2>&1 && grep -v amdgpu gpurun_out/r04_cfg4_qt.log
if [ -n "$STAMPS" ]; then timeout -k 10 200 python tests/tools/dev_arm_fused_stamps.py cfg4 1024 2>&1 | grep -v amdgpu; fi
