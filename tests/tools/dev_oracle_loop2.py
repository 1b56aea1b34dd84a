#!/usr/bin/env python3
"""Development aid (CPU, oracle): warm-started closed loop of B instances; lists the slow solves per control step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle
from robot_mpcs_amd.scenarios import make_scenario
name = sys.argv[1]; B = int(sys.argv[2]); steps = int(sys.argv[3]); thr = int(sys.argv[4]) if len(sys.argv) > 4 else 9
trace_b = int(sys.argv[5]) if len(sys.argv) > 5 else -1; trace_t = int(sys.argv[6]) if len(sys.argv) > 6 else -1
sc = make_scenario(name, B=B, seed=7)
d = dict(sc.desc); d["options"] = dict(d["options"], max_iter=20, acc_iters=3)
o = Oracle(d)
nx, nv, N, nxs = o.nx, o.nv, o.N, o.nx + o.ns
x = sc.xinit.copy(); x0 = sc.x0.copy(); duals = [None] * B
for t in range(steps):
    its = []; slow = []
    for b in range(B):
        if b == trace_b and t == trace_t:
            os.environ["ORC_TRACE"] = "1"
        else:
            os.environ.pop("ORC_TRACE", None)
        r = o.solve_warm(x[b], x0[b], sc.params[b], duals[b])
        duals[b] = r["duals"] if r["exitflag"] >= 0 else (np.zeros((N, o.m)), np.zeros((N, nx)), d["options"]["mu0"] / 1000.0)
        its.append(r["iters"])
        if r["iters"] >= thr or r["exitflag"] not in (1,):
            slow.append((b, r["exitflag"], r["iters"], "%.1e" % r["res_stat"]))
        x[b] = o.dynamics(x[b], r["z"][0, nxs:])
        x0[b] = np.concatenate([r["z"][1:], r["z"][-1:]])
    print(t, "iters mean %.2f max %d" % (np.mean(its), max(its)), "slow:", slow, flush=True)
