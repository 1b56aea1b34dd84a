#!/usr/bin/env python3
"""Development aid: solve seeded batches with the library named by RMPC_LIB_PATH and dump the raw results, so that
two builds can be compared bit for bit (tests/tools/ab_compare.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robot_mpcs_amd._lib import Solver  # noqa: E402
from robot_mpcs_amd.scenarios import make_scenario  # noqa: E402

out = {}
cases = [("cfg1", 1, 0), ("cfg2", 700, 1), ("cfg3", 700, 2), ("cfg4", 300, 3), ("boxer", 130, 4), ("wc_point", 200, 5),
         ("wc_boxer", 200, 6), ("wc_boxer_slack", 200, 7), ("wc_panda", 64, 8), ("cfg2", 4096, 9)]
for name, B, seed in cases:
    sc = make_scenario(name, B=B, seed=seed)
    s = Solver(sc.desc, max_batch=B)
    r = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    for k in ("z", "exitflag", "iters", "kkt", "obj"):
        out[f"{name}_{B}_{k}"] = r[k]
    print(name, B, "iters mean", r["iters"].mean(), "flags", np.unique(r["exitflag"], return_counts=True), flush=True)
np.savez(sys.argv[1], **out)
