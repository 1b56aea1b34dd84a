#!/usr/bin/env python3
"""Independent full-horizon golden vectors: scipy SLSQP solutions of the BASELINE configs at their
real horizons (cfg2 N=30, cfg3 N=30, cfg4 N=20), 8 seeded instances per robot, computed from the
numpy restatement of the NLP (oracle/nlp_numpy.py: values only, stage-wise finite-difference
Jacobians) -- nothing of rmpc_oracle.c or of the HIP kernels takes part.  Writes
``tests/golden/scipy_full_horizon.npz``; minutes of CPU time, so it is run by hand and the result
is committed.  ``tests/test_scipy_golden.py`` (CPU: oracle, GPU: HIP solver) compares against it."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nlp_numpy as ref  # noqa: E402
from robot_mpcs_amd.scenarios import make_scenario  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scipy_full_horizon.npz")
CASES = [("cfg2", 8, 101), ("cfg3", 8, 102), ("cfg4", 8, 103)]


def main():
    out = {}
    only = sys.argv[1:]            # optional: regenerate these cases only, keep the others from the existing file
    if only and os.path.exists(OUT):
        out.update(dict(np.load(OUT)))
    for name, B, seed in CASES:
        if only and name not in only:
            continue
        sc = make_scenario(name, B=B, seed=seed)
        Z = np.zeros_like(sc.x0); status = np.zeros(B, dtype=np.int32); fun = np.zeros(B); nit = np.zeros(B, dtype=np.int32)
        viol = np.zeros(B)
        for b in range(B):
            t0 = time.time()
            nlp = ref.StructuredNLP(sc.desc, sc.xinit[b], sc.params[b])
            Zs, res = nlp.solve_slsqp(sc.x0[b])
            Z[b] = Zs; status[b] = res.status; fun[b] = res.fun; nit[b] = res.nit
            y = nlp.pack(Zs)
            viol[b] = max(np.abs(nlp.eq(y)).max(), -min(0.0, nlp.ineq(y).min()))
            print(f"{name}[{b}] status {res.status} nit {res.nit} f {res.fun:.9g} viol {viol[b]:.2e} {time.time() - t0:.1f}s", flush=True)
        out.update({name + "_xinit": sc.xinit, name + "_x0": sc.x0, name + "_params": sc.params, name + "_z": Z,
                    name + "_status": status, name + "_fun": fun, name + "_nit": nit, name + "_viol": viol,
                    name + "_seed": np.array([seed])})
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
