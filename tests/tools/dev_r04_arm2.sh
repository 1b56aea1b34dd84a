# development aid (round 4): runtime-table fused models of the development library against the oracle, then cfg3 timing
mkdir -p gpurun_out
for c in "cfg3 256 1000" "wc_boxer_slack 96 33" "chain2 96 41" "wc_point 96 31"; do
  echo "== $c"
  timeout -k 10 300 python tests/tools/dev_vs_oracle.py $c > gpurun_out/r04_vs_one.log 2>&1 || { tail -5 gpurun_out/r04_vs_one.log; exit 1; }
  tail -1 gpurun_out/r04_vs_one.log
done
timeout -k 10 300 python tests/tools/quick_time.py cfg3 > gpurun_out/r04_cfg3_qt.log 2>&1; grep -v amdgpu gpurun_out/r04_cfg3_qt.log
