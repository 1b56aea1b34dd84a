mkdir -p gpurun_out
for c in "chain5 48 43" "chain6 48 44" "cfg4 1024 5"; do
  echo "== $c"
  timeout -k 10 120 python tests/tools/dev_vs_oracle.py $c > gpurun_out/r04_vs_one.log 2>&1 || { tail -5 gpurun_out/r04_vs_one.log; exit 1; }
  tail -2 gpurun_out/r04_vs_one.log
done
