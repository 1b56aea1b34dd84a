#!/usr/bin/env python3
"""Evidence run (GPU box): every scenario of robot_mpcs_amd.scenarios at full batch sizes, several seeds each, the HIP
library (through the C-ABI, as the tests do) against the CPU oracle on the same inputs.  Prints one line per
(scenario, seed): exit flags on both sides, the share of instances with the same flag / the same iteration count, and --
over the instances where both converged with the same flag -- the largest difference of the applied control, of the whole
plan and of the objective.  The table is committed as profiles/rNN_parity_sweep.txt; the tests assert the same bars on
fewer seeds.

    python tests/tools/parity_sweep.py [seeds per scenario]  > profiles/r04_parity_sweep.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402
from robot_mpcs_amd._lib import Solver, source_hash  # noqa: E402
from robot_mpcs_amd.scenarios import make_scenario  # noqa: E402

SEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
CASES = [("cfg1", 64), ("cfg2", 4096), ("cfg3", 4096), ("cfg4", 1024), ("pointRobot", 1024), ("boxer", 1024), ("panda", 512),
         ("wc_point", 1024), ("wc_boxer", 1024), ("wc_boxer_slack", 1024), ("wc_panda", 512), ("chain2", 1024), ("chain4", 512),
         ("chain5", 512), ("chain6", 512), ("chain8", 256), ("plug_point", 1024), ("plug_boxer", 1024), ("plug_panda", 512)]

print("# HIP library %s against oracle/rmpc_oracle.c, same inputs; instances x seeds per scenario below" % source_hash())
print("# %-15s %5s %5s | %-22s | %-22s | %9s %9s | %10s %10s %10s" % ("scenario", "B", "seed", "flags HIP", "flags oracle", "same flag",
                                                                       "same iter", "max du1", "max dz", "max dobj"))
worst = dict(flag=1.0, it=1.0, du=0.0)
total = 0
for name, B in CASES:
    for si in range(SEEDS):
        seed = 4000 + 17 * si
        sc = make_scenario(name, B=B, seed=seed)
        o = Oracle(sc.desc)
        s = Solver(sc.desc, max_batch=B)
        g = s.solve(sc.xinit, sc.x0, sc.params)
        s.close()
        r = o.solve_batch(sc.xinit, sc.x0, sc.params)
        nxs = o.nx + o.ns
        same = g["exitflag"] == r["exitflag"]
        conv = same & (g["exitflag"] >= 1)
        du = np.abs(g["z"][:, 0, nxs:] - r["z"][:, 0, nxs:]).max(axis=1)
        dz = np.abs(g["z"] - r["z"]).reshape(B, -1).max(axis=1)
        dob = np.abs(g["obj"] - r["obj"]) / np.maximum(1.0, np.abs(r["obj"]))
        fl = lambda a: " ".join("%d:%d" % (k, v) for k, v in zip(*np.unique(a, return_counts=True)))
        sit = (g["iters"] == r["iters"]).mean()
        print("  %-15s %5d %5d | %-22s | %-22s | %9.4f %9.4f | %10.2e %10.2e %10.2e" % (
            name, B, seed, fl(g["exitflag"]), fl(r["exitflag"]), same.mean(), sit, du[conv].max() if conv.any() else 0.0,
            dz[conv].max() if conv.any() else 0.0, dob[conv].max() if conv.any() else 0.0), flush=True)
        worst["flag"] = min(worst["flag"], same.mean()); worst["it"] = min(worst["it"], sit)
        worst["du"] = max(worst["du"], du[conv].max() if conv.any() else 0.0)
        total += B
print("# %d instances; smallest share with the same flag %.4f, with the same iteration count %.4f; largest difference of the applied "
      "control over the converged instances %.2e" % (total, worst["flag"], worst["it"], worst["du"]))
