#!/usr/bin/env python3
"""Development aid: cycles per section of k_sweep (pass kernels), library built with scripts/dev_build.sh <mask> -DRMPC_STAMPS.
  RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so python tests/tools/dev_sweep_stamps.py cfg4 1024"""
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
v = np.array(list(out), dtype=float)
n = v[7]
names = ["top loads + trial point", "objective, kinematics, distance rows", "single-variable rows", "dynamics, records, log"]
print(f"{cfg} B={B}: k_sweep wavefronts {n:.0f}, iters mean {r['iters'].mean():.2f}, passes {s.last_passes()}")
for i, nm in enumerate(names):
    print(f"  {nm:42s} {v[i] / n:9.0f} cycles per wavefront  {100 * v[i] / v[4]:5.1f} %")
print(f"  whole kernel {v[4] / n:.0f} cycles per wavefront (s_memtime)")
