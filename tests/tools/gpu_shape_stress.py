"""Development aid: GPU vs oracle over odd batch sizes and horizons (index logic of the pass kernels, the half-wave
Riccati blocks and the survivor migration)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robot_mpcs_amd.scenarios import make_scenario
from robot_mpcs_amd._lib import Solver
from oracle.oracle import Oracle
bad = 0
for name, horizons in (("cfg2", (5, 12, 30, 47)), ("cfg3", (8, 30)), ("cfg4", (6, 17, 20, 33)), ("chain5", (10, 24)), ("chain2", (9, 31))):
    for N in horizons:
        for B in (1, 63, 65, 511, 513, 1023, 1025, 1537):
            if name != "cfg2" and B > 600:
                continue
            sc = make_scenario(name, B=B, seed=B + N, time_horizon=N)
            cpu = Oracle(sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)
            s = Solver(sc.desc, max_batch=B)
            gpu = s.solve(sc.xinit, sc.x0, sc.params)
            s.close()
            conv = np.isin(cpu["exitflag"], (1, 2))
            scale = np.maximum(1.0, np.abs(cpu["z"]).reshape(B, -1).max(axis=1))
            dz = (np.abs(gpu["z"] - cpu["z"]).reshape(B, -1).max(axis=1) / scale)
            ef = (gpu["exitflag"] == cpu["exitflag"]).mean(); it = (gpu["iters"] == cpu["iters"]).mean()
            w = dz[conv].max() if conv.any() else 0.0
            flag = "" if (ef == 1.0 and w <= 1e-6) else "   <-- CHECK"
            bad += bool(flag)
            print(f"{name} N {N:3d} B {B:5d} exitflags equal {ef:.4f} iters equal {it:.4f} max rel dz {w:.2e} conv {conv.mean():.3f}{flag}")
print("cases to check:", bad)
