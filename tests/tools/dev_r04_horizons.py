#!/usr/bin/env python3
"""Development aid: short and odd horizons of the fused point-robot path against the oracle (flags, iterations, plans)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402
from robot_mpcs_amd._lib import Solver  # noqa: E402
from robot_mpcs_amd.scenarios import make_scenario  # noqa: E402

for cfg in ("cfg2", "chain2", "plug_point"):
    for N in (1, 2, 3, 5, 17, 31, 32):
        sc = make_scenario(cfg, B=48, seed=100 + N, time_horizon=N)
        cpu = Oracle(sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)
        s = Solver(sc.desc, max_batch=48)
        gpu = s.solve(sc.xinit, sc.x0, sc.params)
        fused = s.is_fused()
        s.close()
        same = gpu["exitflag"] == cpu["exitflag"]
        ok = same & (cpu["exitflag"] >= 1)
        dz = np.abs(gpu["z"][ok] - cpu["z"][ok]).max() if ok.any() else float("nan")
        print(f"{cfg} N={N:2d} fused {int(fused)} same flag {same.mean():.3f} same iters {np.mean(gpu['iters'][ok] == cpu['iters'][ok]):.3f} max dz {dz:.2e} flags {dict(zip(*np.unique(cpu['exitflag'], return_counts=True)))}")
