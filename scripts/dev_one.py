import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import make_scenario
sc = make_scenario("cfg2", B=8, seed=1)
s = Solver(sc.desc, max_batch=8)
r = s.solve(sc.xinit, sc.x0, sc.params)
print(os.environ.get("RMPC_LIB_PATH", "default")[-22:], "fused" if not os.environ.get("RMPC_NO_FUSED") else "pass", r["iters"], r["exitflag"])
