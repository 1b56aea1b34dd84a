set -e
mkdir -p gpurun_out
export RMPC_ALLOW_STALE=1
RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_ricst.so timeout -k 10 200 python tests/tools/dev_ric_stamps.py cfg4 1024 > gpurun_out/r04_ricst_1024.txt 2>&1
RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_ricst.so timeout -k 10 200 python tests/tools/dev_ric_stamps.py cfg4 64 > gpurun_out/r04_ricst_64.txt 2>&1
cat gpurun_out/r04_ricst_1024.txt gpurun_out/r04_ricst_64.txt
unset RMPC_LIB_PATH; unset RMPC_ALLOW_STALE
timeout -k 10 300 python bench.py --config cfg4 --streams 1 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r04_cfg4_s1.json 2> gpurun_out/r04_cfg4_s1.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_cfg4_s1.json"))
print(d["value"], d["batch_latency_ms"])
for k, v in d["roofline"]["all_kernels"].items(): print(k, v["avg_ms"], v["launches"])
PY
