#!/usr/bin/env python3
"""Development aid: phase breakdown of the fused kernel (library built with -DRMPC_STAMPS, RMPC_LIB_PATH set)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robot_mpcs_amd._lib import Solver  # noqa: E402
from robot_mpcs_amd.scenarios import DEFAULT_BATCH, make_scenario  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else DEFAULT_BATCH[cfg]
sc = make_scenario(cfg, B=B, seed=1000)
s = Solver(sc.desc, max_batch=B)
s.solve(sc.xinit, sc.x0, sc.params)
r = s.solve(sc.xinit, sc.x0, sc.params)
# the launch is a queue drained by at most one wavefront per SIMD (4 x compute units; RMPC_FUSED_GRID overrides)
nb = min((B + 1) // 2, int(os.environ.get("RMPC_FUSED_GRID", "1024")))
both = s.fused_stamps(2 * nb).astype(float)
st, sec = both[:nb], both[nb:]
tot = st[:, 4]
print(f"{cfg} B={B}: wavefronts {len(st)}, passes per wavefront mean {st[:, 5].mean():.1f} max {st[:, 5].max():.0f}, "
      f"instance passes per wavefront pass {st[:, 7].sum() / st[:, 5].sum():.2f} (2 = both halves busy), "
      f"passes per instance mean {st[:, 7].sum() / B:.1f}")
for i, name in enumerate(["sweep", "decide", "riccati", "step"]):
    print(f"  {name:8s} {st[:, i].sum() / tot.sum() * 100:5.1f} %   cycles per pass {st[:, i].sum() / st[:, 5].sum():9.0f}")
if sec[:, :6].sum() > 0:
    for i, name in enumerate(["top loads + trial point", "objective, kinematics, distance rows", "single-variable rows", "dynamics, records, log",
                              "step lengths (inside the sweep call)", "their reduction", "after the call: unpark + reductions", "ordering point (wait for the stores)"]):
        print(f"     sweep / {name:38s} {sec[:, i].sum() / st[:, 5].sum():9.0f}")
print(f"  total cycles per pass {tot.sum() / st[:, 5].sum():.0f}   (s_memtime: shader clock)")
