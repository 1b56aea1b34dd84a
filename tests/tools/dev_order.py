#!/usr/bin/env python3
"""Development aid: warm closed loop with and without the longest-first launch order (RMPC_NO_ORDER): identical results."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import make_scenario
for name, B in (("cfg2", 1000), ("cfg3", 777)):
    sc = make_scenario(name, B=B, seed=3)
    d = dict(sc.desc); d["options"] = dict(d["options"], max_iter=20, acc_iters=3)
    a = Solver(d, max_batch=B); a.set_warm_start(True)
    os.environ["RMPC_NO_ORDER"] = "1"
    b = Solver(d, max_batch=B); b.set_warm_start(True)
    os.environ.pop("RMPC_NO_ORDER")
    x = sc.xinit.copy(); x0 = sc.x0.copy()
    nxs = sc.desc["nx"] + sc.desc["ns"]
    for t in range(6):
        ra = a.solve(x, x0, sc.params); rb = b.solve(x, x0, sc.params)
        same = np.array_equal(ra["z"], rb["z"], equal_nan=True) and np.array_equal(ra["exitflag"], rb["exitflag"]) and np.array_equal(ra["iters"], rb["iters"])
        print(name, "step", t, "identical", same, "iters mean %.2f" % ra["iters"].mean(), "passes", a.last_passes(), b.last_passes(), flush=True)
        z = ra["z"]
        ok = ra["exitflag"] >= 0
        x0n = np.concatenate([z[:, 1:], z[:, -1:]], axis=1)
        x0 = np.where(ok[:, None, None], x0n, x0)
        # crude plant: next state = second stage of the plan
        x = np.where(ok[:, None], z[:, 1, :sc.desc["nx"]], x)
    a.close(); b.close()
