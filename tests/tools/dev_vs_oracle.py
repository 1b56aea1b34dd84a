#!/usr/bin/env python3
"""Development aid: one configuration on the GPU against the oracle (flags, iterations, plans).
    python tests/tools/dev_vs_oracle.py cfg4 64 [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import make_scenario
name = sys.argv[1]; B = int(sys.argv[2]); seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
sc = make_scenario(name, B=B, seed=seed)
o = Oracle(sc.desc)
s = Solver(sc.desc, max_batch=B)
g = s.solve(sc.xinit, sc.x0, sc.params)
t0 = time.perf_counter(); g = s.solve(sc.xinit, sc.x0, sc.params); t1 = time.perf_counter()
r = o.solve_batch(sc.xinit, sc.x0, sc.params)
nxs = o.nx + o.ns
same = (g["exitflag"] == r["exitflag"])
du = np.abs(g["z"][:, 0, nxs:] - r["z"][:, 0, nxs:]).max(axis=1)
print(name, "B", B, "solve %.2f ms" % (1e3 * (t1 - t0)), "flags gpu", dict(zip(*np.unique(g["exitflag"], return_counts=True))), "oracle",
      dict(zip(*np.unique(r["exitflag"], return_counts=True))), "same flag %.3f" % same.mean(), "same iters %.3f" % (g["iters"] == r["iters"]).mean(),
      "iters gpu %.2f oracle %.2f" % (g["iters"].mean(), r["iters"].mean()), "max du1 (same flag) %.2e" % du[same].max(),
      "max dz %.2e" % np.abs(g["z"] - r["z"])[same].max())
