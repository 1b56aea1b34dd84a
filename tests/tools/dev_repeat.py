import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import make_scenario
sc = make_scenario("cfg2", B=4096, seed=9)
s = Solver(sc.desc, max_batch=4096)
ref = s.solve(sc.xinit, sc.x0, sc.params)
for rep in range(12):
    r = s.solve(sc.xinit, sc.x0, sc.params)
    bad = np.flatnonzero((r["iters"] != ref["iters"]) | (r["exitflag"] != ref["exitflag"]))
    dz = np.abs(r["z"] - ref["z"]).reshape(4096, -1).max(axis=1)
    print(rep, "differing instances", len(bad), bad[:8], "max dz", dz.max(), "flags", r["exitflag"][bad[:8]], ref["exitflag"][bad[:8]], "iters", r["iters"][bad[:8]], ref["iters"][bad[:8]])
