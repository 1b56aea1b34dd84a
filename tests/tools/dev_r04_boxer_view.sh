# development aid: the boxer's generated view (SpecBoxer, cfg3) against the runtime tables -- plans, wall time, phase stamps
# (libraries: scripts/dev_build.sh 0x20 -> csrc/librmpc_hip_devv.so, the same with -DRMPC_STAMPS -> csrc/librmpc_hip_devvs.so)
mkdir -p gpurun_out
export RMPC_ALLOW_STALE=1
L=$PWD/robot_mpcs_amd/csrc/librmpc_hip_devv.so
LS=$PWD/robot_mpcs_amd/csrc/librmpc_hip_devvs.so
cat > /tmp/bv_dump.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import make_scenario
out = {}
for name, B, seed in (("cfg3", 700, 2), ("cfg3", 4096, 11)):
    sc = make_scenario(name, B=B, seed=seed)
    s = Solver(sc.desc, max_batch=B)
    r = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    for k in ("z", "exitflag", "iters", "kkt", "obj"):
        out[f"{name}_{B}_{k}"] = r[k]
    print(name, B, "iters mean %.3f" % r["iters"].mean(), "flags", np.unique(r["exitflag"], return_counts=True), flush=True)
np.savez(sys.argv[1], **out)
PY
cat > /tmp/bv_cmp.py <<'PY'
import numpy as np
a, b = np.load("gpurun_out/bv_view.npz"), np.load("gpurun_out/bv_tab.npz")
for k in a.files:
    if a[k].dtype.kind == "i":
        print(k, "equal" if np.array_equal(a[k], b[k]) else f"DIFF in {int((a[k] != b[k]).sum())} of {a[k].size}")
    else:
        print(k, "max abs diff %.3e" % np.nanmax(np.abs(a[k] - b[k])), "bitwise equal" if np.array_equal(a[k], b[k]) else "")
PY
RMPC_LIB_PATH=$L timeout -k 10 300 python /tmp/bv_dump.py gpurun_out/bv_view.npz 2>&1 | grep -v amdgpu || exit 1
RMPC_LIB_PATH=$L RMPC_NO_SPEC=1 timeout -k 10 300 python /tmp/bv_dump.py gpurun_out/bv_tab.npz 2>&1 | grep -v amdgpu || exit 1
python /tmp/bv_cmp.py || exit 1
for rep in 1 2; do
  echo "== view"; RMPC_LIB_PATH=$L timeout -k 10 300 python tests/tools/quick_time.py cfg3 2>&1 | grep -v amdgpu || exit 1
  echo "== runtime tables"; RMPC_LIB_PATH=$L RMPC_NO_SPEC=1 timeout -k 10 300 python tests/tools/quick_time.py cfg3 2>&1 | grep -v amdgpu || exit 1
done
echo "== stamps, view"; RMPC_LIB_PATH=$LS timeout -k 10 300 python scripts/fused_stamps.py cfg3 4096 && RMPC_LIB_PATH=$LS timeout -k 10 300 python scripts/fused_stamps.py cfg3 128 || exit 1
echo "== stamps, runtime tables"; RMPC_LIB_PATH=$LS RMPC_NO_SPEC=1 timeout -k 10 300 python scripts/fused_stamps.py cfg3 4096 || exit 1
echo "== parity tests (view)"; RMPC_LIB_PATH=$L timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_scipy_golden.py -m gpu -q -x -k "cfg3" 2>&1 | tail -5
