#!/usr/bin/env python3
"""Development aid: the stale-LDS check of tests/test_gpu_parity.py against the library named by RMPC_LIB_PATH."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import make_scenario
for name, B in (("cfg1", 3), ("cfg2", 2048), ("cfg3", 512)):
    sc = make_scenario(name, B=B, seed=21)
    s = Solver(sc.desc, max_batch=B)
    clean = s.solve(sc.xinit, sc.x0, sc.params)
    s.poison_lds()
    dirty = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    print(name, "flags equal", np.array_equal(clean["exitflag"], dirty["exitflag"]), "z equal", np.array_equal(clean["z"], dirty["z"]),
          "dirty flags", np.unique(dirty["exitflag"], return_counts=True))
