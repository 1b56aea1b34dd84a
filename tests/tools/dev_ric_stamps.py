#!/usr/bin/env python3
"""Development aid: cycles per phase of the generic recursion path (the arm's k_riccati), library built with
scripts/dev_build.sh <mask> -DRMPC_RIC_STAMPS.
  RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so python tests/tools/dev_ric_stamps.py cfg4 1024"""
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
lib.rmpc_debug_ric_stamps(out)          # (clears)
r = s.solve(sc.xinit, sc.x0, sc.params)
lib.rmpc_debug_ric_stamps(out)
v = np.array(list(out), dtype=float)
n = v[7]
N = sc.desc["N"]
names = ["top: image out, record in, fetch, sync", "A: stage Hessian, P rc + p", "q: gradient", "B: Cholesky + gains", "C: cost-to-go", "forward: image to LDS, next request, sync", "forward: products, step out, dx+"]
print(f"{cfg} B={B}: recursions {n:.0f} (instance passes), iters mean {r['iters'].mean():.2f}, passes {s.last_passes()}")
tot = v[:7].sum()
for i, nm in enumerate(names):
    per = v[i] / n
    print(f"  {nm:42s} {per:9.0f} cycles per recursion  {per / N:7.0f} per stage  {100 * v[i] / tot:5.1f} %")
print(f"  total {tot / n:.0f} cycles per recursion (s_memtime)")
