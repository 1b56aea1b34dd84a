#!/usr/bin/env python3
"""Development aid: wall time of repeated solves of one batch, host-buffer entry point (rmpc_solve_batch) against the
device-buffer one (rmpc_solve_batch_device + synchronize) -- is an occasional slow call the kernels' or the transfers'?"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import make_scenario
dev = torch.device("cuda:0")
for name, B in (("boxer", 1024), ("chain5", 512), ("plug_point", 1024)):
    for rep in range(3):
        sc = make_scenario(name, B=B, seed=4000 + 17 * rep)
        d = sc.desc
        N, nv = d["N"], d["nx"] + d["ns"] + d["nu"]
        s = Solver(d, max_batch=B)
        th = []
        for _ in range(12):
            t0 = time.perf_counter(); s.solve(sc.xinit, sc.x0, sc.params); th.append(1e3 * (time.perf_counter() - t0))
        tx = torch.from_numpy(sc.xinit).to(dev); t0_ = torch.from_numpy(sc.x0).to(dev); tp = torch.from_numpy(sc.params).to(dev)
        z = torch.empty((B, N, nv), dtype=torch.float64, device=dev); e = torch.empty(B, dtype=torch.int32, device=dev)
        i = torch.empty(B, dtype=torch.int32, device=dev); k = torch.empty(B, dtype=torch.float64, device=dev); o = torch.empty(B, dtype=torch.float64, device=dev)
        td = []
        for _ in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s.solve_device(B, tx, t0_, tp, z, e, i, k, o); torch.cuda.synchronize(); td.append(1e3 * (time.perf_counter() - t0))
        s.close()
        print(name, B, "seed", 4000 + 17 * rep, "host-buffer calls ms:", " ".join("%.1f" % t for t in th), "| device-buffer calls ms:", " ".join("%.1f" % t for t in td), flush=True)
