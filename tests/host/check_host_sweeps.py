#!/usr/bin/env python3
"""Runs the host build of the single-lane device functions (tests/host/host_sweep.cpp, a sanitizer build) on seeded
instances of BASELINE configs[1..3] and checks what they return against the oracle at the first iterate: the objective,
the equality / inequality residuals, the complementarity products; the null pass and the zero-step trial must reproduce
the first sweep's sums, and the step phase must allow the full step along a zero step."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402
from robot_mpcs_amd import _lib  # noqa: E402
from robot_mpcs_amd.scenarios import make_scenario  # noqa: E402

L = C.CDLL(sys.argv[1])
dp = C.POINTER(C.c_double)
L.host_sweeps.argtypes = [C.POINTER(_lib.RmpcDesc), C.c_int, dp, dp, dp, dp, dp]
L.host_sweeps.restype = C.c_int
ok = True
for cfg, B in (("cfg2", 24), ("cfg3", 24), ("cfg4", 16), ("plug_point", 8)):
    sc = make_scenario(cfg, B=B, seed=123)
    N = sc.desc["N"]
    part = np.zeros((3, B, N, 10))
    step = np.zeros((B, N, 3))
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (sc.xinit, sc.x0, sc.params)]
    rc = L.host_sweeps(C.byref(_lib.make_desc(sc.desc)), B, *(x.ctypes.data_as(dp) for x in a), part.ctypes.data_as(dp), step.ctypes.data_as(dp))
    assert rc == 0, (cfg, rc)
    o = Oracle(sc.desc)
    f0 = np.zeros(B); req0 = np.zeros(B); rin0 = np.zeros(B); rc0 = np.zeros(B)
    for b in range(B):
        d = dict(sc.desc); d["options"] = dict(d["options"]); d["options"]["max_iter"] = 0
        r = Oracle(d).solve(sc.xinit[b], sc.x0[b], sc.params[b])
        f0[b], req0[b], rin0[b], rc0[b] = r["obj"], r["res_eq"], r["res_ineq"], r["res_comp"]
    f = part[:, :, :, 0].sum(axis=2)
    req = part[:, :, :, 4].max(axis=2); rin = part[:, :, :, 5].max(axis=2); rcomp = part[:, :, :, 6].max(axis=2)
    e = [np.abs(f[0] - f0).max() / max(1.0, np.abs(f0).max()), np.abs(req[0] - req0).max(), np.abs(rin[0] - rin0).max(), np.abs(rcomp[0] - rc0).max()]
    # (the objective does not depend on the slacks: both later passes reproduce it; the constraint violation only in the null
    #  pass -- a row whose slack starts above its value moves its slack even along a zero step of the variables)
    rep = [np.abs(f[p] - f[0]).max() / max(1.0, np.abs(f0).max()) for p in (1, 2)] + [np.abs(part[1, :, :, 1] - part[0, :, :, 1]).max()]
    good = max(e) <= 1e-9 and max(rep) <= 1e-9 and np.all(part[:, :, :, 9] == 0) and np.all(np.isfinite(part)) and np.all(step[:, :, :2] > 0.0) and np.all(step[:, :, :2] <= 1.0)
    ok &= bool(good)
    print(f"{cfg}: B={B} first sweep vs oracle: objective {e[0]:.2e} (rel), eq {e[1]:.2e}, ineq {e[2]:.2e}, comp {e[3]:.2e}; "
          f"null pass / zero-step trial reproduce it: {max(rep):.2e}; step phase: primal length {step[:, :, 0].min():.3f}, dual {step[:, :, 1].min():.3f}  {'ok' if good else 'MISMATCH'}")
print("host sweeps under the sanitizers:", "clean" if ok else "FAILED")
sys.exit(0 if ok else 1)
