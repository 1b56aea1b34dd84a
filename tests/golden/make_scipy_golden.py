#!/usr/bin/env python3
"""Independent full-horizon golden vectors: scipy solutions of the BASELINE configs at their real horizons (cfg2
N=30, cfg3 N=30 with slack, cfg4 N=20), computed from the numpy restatement of the NLP (oracle/nlp_numpy.py: values
only, stage-wise finite-difference Jacobians) -- nothing of rmpc_oracle.c or of the HIP kernels takes part.

  * 64 seeded instances per robot through SLSQP (active-set SQP);
  * the first few of each (6 / 2 / 2) again through trust-constr (trust-region interior point): a second algorithm family;
  * one warm-started closed loop per robot: 6 control steps, the plant is the numpy ERK2 map, every solve starts
    from the shifted previous plan (shiftHorizon, mpcPlanner.py:215-226);
  * a sensitivity line for the ERK2 tableau the reference's code generator uses (unverifiable here, SURVEY.md 8a
    row A3): the first 8 boxers solved again with Heun's rule instead of the explicit midpoint rule.

Writes ``tests/golden/scipy_full_horizon.npz``; about 15 minutes on 8 cores, run by hand, result committed.
``tests/test_scipy_golden.py`` (CPU: oracle, GPU: HIP solver) compares against it.
usage: python tests/golden/make_scipy_golden.py [workers]"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nlp_numpy as ref  # noqa: E402
from robot_mpcs_amd.scenarios import make_scenario  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scipy_full_horizon.npz")
CASES = [("cfg2", 64, 101), ("cfg3", 64, 102), ("cfg4", 64, 103)]
# instances solved again with trust-constr and its iteration limit (a BFGS interior-point method on 270 .. 420 variables
# with finite-difference Jacobians: 0.6 .. 3 s per iteration; it is stopped at the limit, by which it agrees with SLSQP to
# 1e-6 in the first control)
N_TC = {"cfg2": 6, "cfg3": 2, "cfg4": 2}
TC_MAXITER = {"cfg2": 250, "cfg3": 150, "cfg4": 120}
N_HEUN = 8      # boxers solved again with Heun's rule
LOOP_STEPS = 6

_SC = {}


def _scenario(name):
    if name not in _SC:
        B, seed = {n: (b, s) for n, b, s in CASES}[name]
        _SC[name] = make_scenario(name, B=B, seed=seed)
    return _SC[name]


def _violation(nlp, Z):
    y = nlp.pack(Z)
    return max(np.abs(nlp.eq(y)).max(), -min(0.0, nlp.ineq(y).min()))


def job(task):
    kind, name, b = task
    sc = _scenario(name)
    t0 = time.time()
    if kind == "heun":
        ref.ERK2_TABLEAU = "heun"
    try:
        nlp = ref.StructuredNLP(sc.desc, sc.xinit[b], sc.params[b])
        if kind == "tc":
            Z, res = nlp.solve_trust_constr(sc.x0[b], maxiter=TC_MAXITER[name])
            out = dict(z=Z, fun=float(res.fun), status=int(res.status), nit=int(res.nit), viol=_violation(nlp, Z))
        elif kind == "loop":
            x = sc.xinit[b].copy()
            Z0 = sc.x0[b].reshape(sc.desc["N"], -1).copy()
            nx, ns = sc.desc["nx"], sc.desc["ns"]
            xs, zs, funs = [], [], []
            for _ in range(LOOP_STEPS):
                nlp = ref.StructuredNLP(sc.desc, x, sc.params[b])
                Z, res = nlp.solve_slsqp(Z0)
                xs.append(x.copy()); zs.append(Z.copy()); funs.append(float(res.fun))
                x = ref.dynamics(sc.desc, x, Z[0, nx + ns:])
                Z0 = np.vstack([Z[1:], Z[-1:]])           # shifted plan, last stage repeated
            out = dict(x=np.array(xs), z=np.array(zs), fun=np.array(funs))
        else:
            Z, res = nlp.solve_slsqp(sc.x0[b])
            out = dict(z=Z, fun=float(res.fun), status=int(res.status), nit=int(res.nit), viol=_violation(nlp, Z))
    finally:
        ref.ERK2_TABLEAU = "midpoint"
    print(f"{kind} {name}[{b}] {time.time() - t0:.1f}s", flush=True)
    return task, out


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    tasks = []
    for name, B, _ in CASES:
        tasks += [("loop", name, 0)]
    for name, B, _ in reversed(CASES):            # longest jobs first
        tasks += [("tc", name, b) for b in range(N_TC[name])]
        tasks += [("slsqp", name, b) for b in range(B)]
    tasks += [("heun", "cfg3", b) for b in range(N_HEUN)]
    with Pool(workers) as pool:
        results = dict(pool.imap_unordered(job, tasks))
    out = {}
    for name, B, seed in CASES:
        sc = _scenario(name)
        # the inputs are the scenario generator's (same seed): a hash pins them, the test regenerates them.  Of the 64
        # SLSQP solutions only the first stage (the applied control) and the objective are kept: the fixture stays small.
        r = [results[("slsqp", name, b)] for b in range(B)]
        out.update({name + "_inputs_sha256": np.array(inputs_sha(sc)), name + "_seed": np.array([seed]),
                    name + "_z1": np.array([x["z"][0] for x in r]), name + "_status": np.array([x["status"] for x in r], dtype=np.int32),
                    name + "_fun": np.array([x["fun"] for x in r]), name + "_nit": np.array([x["nit"] for x in r], dtype=np.int32),
                    name + "_viol": np.array([x["viol"] for x in r])})
        t = [results[("tc", name, b)] for b in range(N_TC[name])]
        out.update({name + "_tc_z1": np.array([x["z"][0] for x in t]), name + "_tc_fun": np.array([x["fun"] for x in t]),
                    name + "_tc_viol": np.array([x["viol"] for x in t]), name + "_tc_status": np.array([x["status"] for x in t], dtype=np.int32)})
        lp = results[("loop", name, 0)]
        out.update({name + "_loop_x": lp["x"], name + "_loop_z1": lp["z"][:, 0], name + "_loop_fun": lp["fun"]})
    h = [results[("heun", "cfg3", b)] for b in range(N_HEUN)]
    out.update({"cfg3_heun_z1": np.array([x["z"][0] for x in h]), "cfg3_heun_fun": np.array([x["fun"] for x in h]),
                "cfg3_heun_viol": np.array([x["viol"] for x in h])})
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


def inputs_sha(sc) -> str:
    import hashlib
    hsh = hashlib.sha256()
    for a in (sc.xinit, sc.x0, sc.params):
        hsh.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return hsh.hexdigest()


if __name__ == "__main__":
    main()
