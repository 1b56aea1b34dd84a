# development aid (round 4): the scaled-curvature retries (library: scripts/dev_build.sh 0x65) against the oracle on whole input sets, fused and pass kernels
mkdir -p gpurun_out
export RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so
for c in "cfg2 4096 1000" "cfg2 4096 1068" "cfg2 4096 1085" "chain2 2048 500" "cfg2 16384 3000" "cfg3 2048 2017" "cfg4 1024 2000"; do
  echo "== $c"
  timeout -k 10 300 python tests/tools/dev_vs_oracle.py $c > gpurun_out/r04_vs_one.log 2>&1 || { tail -5 gpurun_out/r04_vs_one.log; exit 1; }
  tail -1 gpurun_out/r04_vs_one.log | cut -c1-360
done
for c in "cfg2 4096 1000" "chain2 2048 500"; do
  echo "== pass kernels $c"
  RMPC_NO_FUSED=1 timeout -k 10 300 python tests/tools/dev_vs_oracle.py $c > gpurun_out/r04_vs_one.log 2>&1 || { tail -5 gpurun_out/r04_vs_one.log; exit 1; }
  tail -1 gpurun_out/r04_vs_one.log | cut -c1-360
  echo "== lane recursion $c"
  RMPC_NO_FUSED=1 RMPC_RIC_LANE=2 timeout -k 10 300 python tests/tools/dev_vs_oracle.py $c > gpurun_out/r04_vs_one.log 2>&1 || { tail -5 gpurun_out/r04_vs_one.log; exit 1; }
  tail -1 gpurun_out/r04_vs_one.log | cut -c1-360
done
