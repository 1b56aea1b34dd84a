#!/usr/bin/env python3
"""Development aid: first sweep of a configuration against the oracle's stage evaluation, field by field."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import make_scenario

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
B = 3
sc = make_scenario(name, B=B, seed=9)
o = Oracle(sc.desc)
s = Solver(sc.desc, max_batch=B)
print("spec:", repr(s.spec_name()))
dbg = s.debug_sweep(sc.xinit, sc.x0, sc.params)
s.close()
N, nv, nx, nh = o.N, o.nv, o.nx, o.nh
for key in dbg:
    a = np.asarray(dbg[key])
    print(key, a.shape, "nan count", int(np.isnan(a).sum()), "of", a.size)
b = 0
P = sc.params[b].reshape(N, o.npar)
for k in range(min(N, 2)):
    e = o.eval_stage(sc.x0[b, k], P[k], fixed_state=(k == 0))
    print("k", k, "f gpu", dbg["f"][b, k], "oracle", e["f"])
    print("   g diff", np.abs(np.asarray(dbg["g"][b, k])[:nh] - e["g"][:nh]).max())
    for key in ("Q", "q0", "q1", "rc"):
        a = np.asarray(dbg[key][b, k]); print("  ", key, "nan", int(np.isnan(a).sum()), "of", a.size)
print("f[:,0] gpu", dbg["f"][:, 0], " f[:,1]", dbg["f"][:, 1])
for b in range(B):
    P = sc.params[b].reshape(N, o.npar)
    print(" oracle f[b,0], f[b,1]", o.eval_stage(sc.x0[b, 0], P[0], fixed_state=True)["f"], o.eval_stage(sc.x0[b, 1], P[1])["f"])
    e = o.eval_stage(sc.x0[b, 0], P[0], fixed_state=True)
    print("   q0 diff k=0", np.abs(np.asarray(dbg["q0"][b, 0]) - (e["gf"] + e["Jg"].T @ ((sc.desc["options"]["mu0"] / np.maximum(e["g"], 1e-2)) * (e["g"] - np.maximum(e["g"], 1e-2)) / np.maximum(e["g"], 1e-2)))).max())
