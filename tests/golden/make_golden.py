"""Generates the committed golden vectors from the CPU oracle.

The reference ships no golden vectors and cannot be imported here (SURVEY.md
8c), so these fixtures are outputs of oracle/rmpc_oracle.c -- itself pinned by
tests/test_oracle_*.py -- on the seeded scenarios of robot_mpcs_amd/scenarios.py.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.oracle import Oracle  # noqa: E402
from robot_mpcs_amd.scenarios import make_scenario  # noqa: E402

CASES = [("cfg1", 1, 0), ("cfg2", 8, 21), ("cfg3", 8, 22), ("cfg4", 8, 23), ("boxer", 4, 24), ("pointRobot", 2, 25)]


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, B, seed in CASES:
        sc = make_scenario(name, B=B, seed=seed)
        o = Oracle(sc.desc)
        r = o.solve_batch(sc.xinit, sc.x0, sc.params)
        assert np.all(np.isin(r["exitflag"], (1, 2))), (name, r["exitflag"])
        # stage-level vectors of the first instance at its start point
        p0 = sc.params[0].reshape(o.N, o.npar)[0]
        e = o.eval_stage(sc.x0[0, 0], p0)
        np.savez_compressed(
            os.path.join(out_dir, f"{name}.npz"),
            desc=json.dumps(sc.desc), xinit=sc.xinit, x0=sc.x0, params=sc.params,
            z=r["z"], exitflag=r["exitflag"], iters=r["iters"], obj=r["obj"],
            stage_f=e["f"], stage_gf=e["gf"], stage_H=e["H"], stage_g=e["g"], stage_Jg=e["Jg"],
            stage_xnext=e["xnext"], stage_A=e["A"], stage_B=e["B"],
        )
        print(name, "B", B, "iters", r["iters"].tolist())


if __name__ == "__main__":
    main()
