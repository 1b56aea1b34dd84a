for m in 0 0.01 0.1 1; do timeout -k 10 200 python bench.py --config cfg5 --steps 300 --warmup 10 --mu-regoal-boxer $m > gpurun_out/c5_$m.json 2>/dev/null; python - <<PY
import json
d=json.load(open("gpurun_out/c5_$m.json"))
print("mu_regoal $m: ms/step %.2f p99 %.2f hit %.2f" % (d["ms_per_step"], d["loop"]["ms_p99"], d["loop"]["deadline_10ms_hit_rate_rank0"]), {k:(round(v["usable_share"],3), round(v["iters_mean"],2), round(v["cut_by_deadline_or_iteration_cap_per_step"])) for k,v in d["per_fleet"].items()})
PY
done
