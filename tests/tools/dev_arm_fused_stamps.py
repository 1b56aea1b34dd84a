#!/usr/bin/env python3
"""Development aid: phase breakdown of k_fused_arm (library built with scripts/dev_build.sh 0x4 -DRMPC_STAMPS).
  RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so python tests/tools/dev_arm_fused_stamps.py cfg4 1024"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robot_mpcs_amd import _lib  # noqa: E402
from robot_mpcs_amd._lib import Solver  # noqa: E402
from robot_mpcs_amd.scenarios import make_scenario  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sc = make_scenario(cfg, B=B, seed=1000)
s = Solver(sc.desc, max_batch=B)
lib = _lib.load_library()
out = (C.c_longlong * 8)()
s.solve(sc.xinit, sc.x0, sc.params)
lib.rmpc_debug_sweep_stamps(out)          # (clears)
r = s.solve(sc.xinit, sc.x0, sc.params)
lib.rmpc_debug_sweep_stamps(out)
sw = np.array(list(out), dtype=float)
nb = min(B, int(os.environ.get("RMPC_FUSED_GRID", "1024")))
st = s.fused_stamps(nb).astype(float)
passes = st[:, 5].sum()
print(f"{cfg} B={B}: wavefronts {nb}, instance passes {passes:.0f} ({passes / B:.1f} per instance), iters mean {r['iters'].mean():.2f}")
for i, nm in enumerate(["sweep call (step lengths + sweep + reductions)", "decisions", "recursion call", "queue: epilogue / prologue"]):
    print(f"  {nm:48s} {st[:, i].sum() / passes:9.0f} cycles per instance pass  {100 * st[:, i].sum() / st[:, 4].sum():5.1f} %")
print(f"  total {st[:, 4].sum() / passes:.0f} cycles per instance pass")
names = ["q + chain walk", "slots: rows, blocks", "sums over the parts", "joints: rows, stationarity, records", "block stores", "step lengths + reduction", "reduction of the partials"]
for i, nm in enumerate(names):
    print(f"     sweep call / {nm:38s} {sw[i] / sw[7]:9.0f}")
