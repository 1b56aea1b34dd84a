#!/usr/bin/env python3
"""Development aid: time of one configuration's solve (device-resident, per-kernel profile of the pass kernels).
    python tests/tools/dev_time.py cfg4 1024 [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import make_scenario
name = sys.argv[1]; B = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
sc = make_scenario(name, B=B, seed=1)
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
s = Solver(sc.desc, max_batch=B)
if os.environ.get("DEV_BUDGET"):   # only the first passes of a solve: every instance still iterates in each launch
    s.set_pass_budget(int(os.environ["DEV_BUDGET"]))
xi, x0, pa = t(sc.xinit), t(sc.x0), t(sc.params)
N, nv = sc.desc["N"], s.nvar
z = torch.empty((B, N, nv), dtype=torch.float64, device=dev); ef = torch.empty(B, dtype=torch.int32, device=dev)
it = torch.empty(B, dtype=torch.int32, device=dev); kk = torch.empty(B, dtype=torch.float64, device=dev); ob = torch.empty(B, dtype=torch.float64, device=dev)
for _ in range(2):
    s.solve_device(B, xi, x0, pa, z, ef, it, kk, ob)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    s.solve_device(B, xi, x0, pa, z, ef, it, kk, ob)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / reps
s.set_profiling(True)
s.solve_device(B, xi, x0, pa, z, ef, it, kk, ob)
torch.cuda.synchronize()
pr = s.get_profile()
print(name, B, "ms per solve %.3f" % ms, "iters mean %.2f" % it.float().mean().item(), "passes", s.last_passes(), "zsum %.10e" % float(z.nan_to_num().sum()),
      {k: (round(1e3 * v["total_ms"] / max(1, v["launches"]), 1), v["launches"]) for k, v in pr.items() if v["launches"]})
