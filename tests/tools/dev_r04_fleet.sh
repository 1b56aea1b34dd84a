# development aid (round 4): the mixed-fleet tests with the development settings
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --config cfg5 --steps 400 --warmup 10 > gpurun_out/r04_cfg5.json 2> gpurun_out/r04_cfg5.err || { tail gpurun_out/r04_cfg5.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_cfg5.json"))
print(d["value"], d["ms_per_step"], d["loop"])
for k, v in d["per_fleet"].items(): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
PY
timeout -k 10 900 python -m pytest tests/test_gpu_warm_start.py tests/test_gpu_scene.py -x -q -m gpu > gpurun_out/r04_fleet_tests.log 2>&1 || { tail -30 gpurun_out/r04_fleet_tests.log; exit 1; }
tail -3 gpurun_out/r04_fleet_tests.log
