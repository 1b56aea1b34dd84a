#!/bin/bash
# development aid: the bench line in the driver's protocol (--steps 20 --warmup 5) and with the defaults, plus a digest
TAG=${1:-r04}
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_driver.json 2> gpurun_out/bench_${TAG}_driver.err || tail -c 600 gpurun_out/bench_${TAG}_driver.err
timeout -k 10 500 python bench.py > gpurun_out/bench_${TAG}_default.json 2> gpurun_out/bench_${TAG}_default.err || tail -c 600 gpurun_out/bench_${TAG}_default.err
python - <<PY
import json
for f in ("driver", "default"):
    d = json.load(open("gpurun_out/bench_${TAG}_%s.json" % f))
    print(f, "value %.4g single %.4g ms/step %.3f lat %.3f" % (d["value"], d["value_single_call"], d["ms_per_step"], d["batch_latency_ms"]), [round(x, 2) for x in d["single_call_ms"]])
    r = d["roofline"]
    print("  roofline frac %.3f avg_ms %.3f fp64 %.4f fullchip %.3f traffic %s" % (r["frac"], r["avg_launch_ms"], r["frac_fp64_vector"], r.get("full_chip_frac", 0), r["traffic"]))
    for k, v in d["legs"].items():
        print("  leg", k, "%.4g single %.4g" % (v["value"], v["value_single_call"]), "roof %.3f" % v["roofline"]["frac"], v["roofline"]["kernel"], "%.3f ms" % v["roofline"]["avg_launch_ms"])
    c = d["extra"]["cfg5"]
    print("  cfg5 ms %.2f p99 %.2f hit %.3f" % (c["ms_per_step"], c["loop"]["ms_p99"], c["loop"]["deadline_10ms_hit_rate_rank0"]), {k: round(v["usable_share"], 3) for k, v in c["per_fleet"].items()})
    print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
