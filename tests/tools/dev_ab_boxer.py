#!/usr/bin/env python3
"""Development aid: boxer configurations through the fused kernel and through the pass kernels (RMPC_NO_FUSED), compared."""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "dump":
    from robot_mpcs_amd._lib import Solver
    from robot_mpcs_amd.scenarios import make_scenario
    out = {}
    for name, B, seed in (("cfg3", 700, 2), ("boxer", 130, 4), ("wc_boxer", 200, 6), ("wc_boxer_slack", 200, 7), ("cfg3", 4096, 11)):
        sc = make_scenario(name, B=B, seed=seed)
        s = Solver(sc.desc, max_batch=B)
        r = s.solve(sc.xinit, sc.x0, sc.params)
        s.close()
        for k in ("z", "exitflag", "iters", "kkt", "obj"):
            out[f"{name}_{B}_{k}"] = r[k]
        print(name, B, "iters mean %.3f" % r["iters"].mean(), "flags", np.unique(r["exitflag"], return_counts=True), flush=True)
    np.savez(sys.argv[2], **out)
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "dump", "gpurun_out/ab_fused.npz"], env=env)
    env["RMPC_NO_FUSED"] = "1"
    subprocess.check_call([sys.executable, __file__, "dump", "gpurun_out/ab_pass.npz"], env=env)
    a, b = np.load("gpurun_out/ab_fused.npz"), np.load("gpurun_out/ab_pass.npz")
    worst = 0.0
    for k in a.files:
        if a[k].dtype.kind == "i":
            print(k, "equal" if np.array_equal(a[k], b[k]) else f"DIFF in {int((a[k] != b[k]).sum())} of {a[k].size}")
        else:
            d = np.nanmax(np.abs(a[k] - b[k])) if a[k].size else 0.0
            worst = max(worst, d)
            print(k, "max abs diff %.3e" % d, "bitwise equal" if np.array_equal(a[k], b[k]) else "")
