#!/usr/bin/env python3
"""Development aid: fused vs pass kernels on small variations of a config, in one process (two handles)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robot_mpcs_amd._lib import Solver  # noqa: E402
from robot_mpcs_amd.scenarios import make_scenario  # noqa: E402


def run(name, B, seed, **over):
    sc = make_scenario(name, B=B, seed=seed, **over)
    f = Solver(sc.desc, max_batch=B)
    os.environ["RMPC_NO_FUSED"] = "1"
    p = Solver(sc.desc, max_batch=B)
    del os.environ["RMPC_NO_FUSED"]
    a, b = f.solve(sc.xinit, sc.x0, sc.params), p.solve(sc.xinit, sc.x0, sc.params)
    f.close(); p.close()
    same = all(np.array_equal(a[k], b[k]) for k in ("z", "exitflag", "iters"))
    bad = np.flatnonzero((a["iters"] != b["iters"]) | (np.abs(a["z"] - b["z"]).reshape(B, -1).max(axis=1) > 0))
    print(name, over, "B", B, "identical" if same else f"DIFF in {len(bad)} instances: {bad[:10]}",
          "iters fused", a["iters"][bad[:6]], "pass", b["iters"][bad[:6]], "flags", a["exitflag"][bad[:6]], b["exitflag"][bad[:6]], flush=True)


run("cfg2", 64, 1, time_horizon=12)
run("cfg2", 64, 1, time_horizon=20)
run("cfg2", 64, 1, time_horizon=30)
run("cfg2", 64, 1, time_horizon=32)
run("wc_point", 64, 1, time_horizon=30)
run("cfg1", 1, 0, time_horizon=30)
run("cfg2", 2, 1)
for it in (1, 2, 3, 5, 8):
    sc = make_scenario("cfg2", B=64, seed=1)
    d = dict(sc.desc); d["options"] = dict(d["options"], max_iter=it)
    f = Solver(d, max_batch=64)
    os.environ["RMPC_NO_FUSED"] = "1"
    p = Solver(d, max_batch=64)
    del os.environ["RMPC_NO_FUSED"]
    a, b = f.solve(sc.xinit, sc.x0, sc.params), p.solve(sc.xinit, sc.x0, sc.params)
    dz = np.abs(a["z"] - b["z"]).reshape(64, -1).max(axis=1)
    print("max_iter", it, "max |dz|", dz.max(), "instances differing", (dz > 0).sum(), "first", np.flatnonzero(dz > 0)[:8])
    f.close(); p.close()
