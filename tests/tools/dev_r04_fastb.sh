# development aid: the point robot's Schur-complement recursion in k_fused (library: scripts/dev_build.sh 0x41): parity, timing
mkdir -p gpurun_out
export RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so
for c in "cfg2 256 1000" "chain2 96 41" "wc_point 96 31" "cfg1 1 3"; do
  echo "== $c"
  timeout -k 10 300 python tests/tools/dev_vs_oracle.py $c > gpurun_out/r04_vs_one.log 2>&1 || { tail -5 gpurun_out/r04_vs_one.log; exit 1; }
  tail -1 gpurun_out/r04_vs_one.log
done
timeout -k 10 300 python tests/tools/quick_time.py cfg2 2>&1 | grep -v amdgpu
