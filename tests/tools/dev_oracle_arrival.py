#!/usr/bin/env python3
"""Development aid (CPU, oracle only): warm-started closed loop of a few instances; prints the distance of the end link
to the goal every few control steps -- why do the arms of the steady fleet loop time out on their goals?
    python tests/tools/dev_oracle_arrival.py cfg4 4 150"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle
from oracle import nlp_numpy as nn
from robot_mpcs_amd.scenarios import make_scenario
name = sys.argv[1]; B = int(sys.argv[2]); steps = int(sys.argv[3])
opts = {"max_iter": 40, "acc_iters": 3}
sc = make_scenario(name, B=B, seed=7)
d = dict(sc.desc); d["options"] = dict(d["options"], **opts)
o = Oracle(d)
nx, nv, N, nxs = o.nx, o.nv, o.N, o.nx + o.ns
x = sc.xinit.copy(); x0 = sc.x0.copy(); duals = [None] * B
goal = sc.extra["goal"]
def end_dist(b):
    p = nn.fk(d, x[b][:o.n], d["end_frame"])
    return float(np.linalg.norm(np.asarray(p)[:3] - goal[b]))
hist = []
for t in range(steps):
    dist = []
    for b in range(B):
        r = o.solve_warm(x[b], x0[b], sc.params[b], duals[b])
        duals[b] = r["duals"] if r["exitflag"] >= 0 else (np.zeros((N, o.m)), np.zeros((N, nx)), d["options"]["mu0"] / 1000.0)
        x[b] = o.dynamics(x[b], r["z"][0, nxs:])
        x0[b] = np.concatenate([r["z"][1:], r["z"][-1:]])
        dist.append((r["exitflag"], r["iters"], float(np.abs(r["z"][0, nxs:]).max()), float(np.abs(x[b][o.n:nx]).max()), end_dist(b)))
    hist.append(dist)
    if t % 10 == 0 or t == steps - 1:
        print(t, " | ".join("f%d it%2d |u| %.2f |qd| %.2f dist %.3f" % dd for dd in dist), flush=True)
