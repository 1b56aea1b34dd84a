"""Development aid: GPU vs oracle over several seeds and batch sizes (wider than the test-suite cases)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robot_mpcs_amd.scenarios import make_scenario
from robot_mpcs_amd._lib import Solver
from oracle.oracle import Oracle
worst = 0.0
for name, B in [('cfg2', 1500), ('cfg3', 1100), ('cfg4', 300), ('boxer', 130), ('pointRobot', 70), ('panda', 9), ('chain4', 100), ('chain6', 100), ('wc_panda', 60)]:
    for seed in (101, 202, 303):
        sc = make_scenario(name, B=B, seed=seed)
        cpu = Oracle(sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)
        s = Solver(sc.desc, max_batch=B)
        gpu = s.solve(sc.xinit, sc.x0, sc.params)
        s.close()
        conv = np.isin(cpu['exitflag'], (1, 2))
        scale = np.maximum(1.0, np.abs(cpu['z']).reshape(B, -1).max(axis=1))
        dz = np.abs(gpu['z'] - cpu['z']).reshape(B, -1).max(axis=1) / scale
        ef = (gpu['exitflag'] == cpu['exitflag']).mean()
        it = (gpu['iters'] == cpu['iters']).mean()
        w = dz[conv].max() if conv.any() else 0.0
        worst = max(worst, w)
        print(f"{name:10s} B {B:5d} seed {seed} exitflags equal {ef:.4f} iters equal {it:.4f} max rel dz (converged) {w:.2e} exits {dict(zip(*np.unique(gpu['exitflag'], return_counts=True)))}")
print("worst relative plan difference", worst)
