import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import make_scenario
for name, B, seed in [("cfg1", 1, 0), ("cfg2", 700, 1), ("cfg2", 700, 1), ("cfg1", 1, 0), ("cfg2", 4096, 9)]:
    sc = make_scenario(name, B=B, seed=seed)
    s = Solver(sc.desc, max_batch=B)
    g = s.solve(sc.xinit, sc.x0, sc.params)
    print(name, B, "spec", repr(s.spec_name()), "iters mean %.4f" % g["iters"].mean(), "flags", np.unique(g["exitflag"], return_counts=True), "zsum %.12e" % np.nansum(g["z"]), flush=True)
    s.close()
