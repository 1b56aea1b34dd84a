#!/usr/bin/env python3
"""Development aid: iterates after 1, 2, 3 iterations (max_iter) with the library named by RMPC_LIB_PATH, dumped for a
comparison of two builds -- finds the first iteration at which a changed recursion gives another step."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robot_mpcs_amd._lib import Solver  # noqa: E402
from robot_mpcs_amd.scenarios import make_scenario  # noqa: E402

cfg, B = sys.argv[2], int(sys.argv[3])
out = {}
for it in (1, 2, 3, 5, 8, 12, 16):
    sc = make_scenario(cfg, B=B, seed=3)
    d = dict(sc.desc); d["options"] = dict(d["options"]); d["options"]["max_iter"] = it
    s = Solver(d, max_batch=B)
    r = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    out[f"z{it}"] = r["z"]; out[f"f{it}"] = r["exitflag"]
np.savez(sys.argv[1], **out)
