#!/usr/bin/env python3
"""Development aid: solve one configuration a few times (fresh handle each time), print flag / iteration statistics."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import make_scenario
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 700
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sc = make_scenario(name, B=B, seed=1)
for r in range(reps):
    s = Solver(sc.desc, max_batch=B)
    g = s.solve(sc.xinit, sc.x0, sc.params)
    print(name, B, "spec", repr(s.spec_name()), "iters mean %.4f" % g["iters"].mean(), "flags", np.unique(g["exitflag"], return_counts=True),
          "zsum %.12e" % np.nansum(g["z"]), flush=True)
    s.close()
