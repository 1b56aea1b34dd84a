mkdir -p gpurun_out
for c in "cfg3 256 1000" "boxer 65 4" "wc_boxer 96 32" "wc_boxer_slack 96 33"; do
  echo "== $c"
  timeout -k 10 300 python tests/tools/dev_vs_oracle.py $c > gpurun_out/r04_vs_one.log 2>&1 || { tail -5 gpurun_out/r04_vs_one.log; exit 1; }
  tail -1 gpurun_out/r04_vs_one.log
done
RMPC_NO_FUSED=1 timeout -k 10 300 python tests/tools/dev_vs_oracle.py cfg3 128 5 > gpurun_out/r04_vs_one.log 2>&1 || { tail -5 gpurun_out/r04_vs_one.log; exit 1; }
echo "== cfg3 pass kernels"; tail -1 gpurun_out/r04_vs_one.log
timeout -k 10 300 python tests/tools/quick_time.py cfg3 > gpurun_out/r04_cfg3_qt.log 2>&1; grep -v amdgpu gpurun_out/r04_cfg3_qt.log
