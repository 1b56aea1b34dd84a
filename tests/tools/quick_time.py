#!/usr/bin/env python3
"""Development aid: wall time of rmpc_solve_batch_device for one batch alone and for S handles in flight."""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robot_mpcs_amd._lib import Solver  # noqa: E402
from robot_mpcs_amd.scenarios import DEFAULT_BATCH, make_scenario  # noqa: E402

cfgs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cfg2"]
dev = torch.device("cuda:0")
for cfg in cfgs:
    B = DEFAULT_BATCH[cfg]
    sc = make_scenario(cfg, B=B, seed=1000)
    d = sc.desc
    N, nv = d["N"], d["nx"] + d["ns"] + d["nu"]
    for S in tuple(int(x) for x in os.environ.get("QT_STREAMS", "1,2,4").split(",")):
        solvers = [Solver(d, max_batch=B) for _ in range(S)]
        tx = torch.from_numpy(sc.xinit).to(dev); t0 = torch.from_numpy(sc.x0).to(dev); tp = torch.from_numpy(sc.params).to(dev)
        outs = [dict(z=torch.empty((B, N, nv), dtype=torch.float64, device=dev), e=torch.empty(B, dtype=torch.int32, device=dev),
                     i=torch.empty(B, dtype=torch.int32, device=dev), k=torch.empty(B, dtype=torch.float64, device=dev),
                     o=torch.empty(B, dtype=torch.float64, device=dev)) for _ in range(S)]
        streams = [torch.cuda.Stream() for _ in range(S)]

        def run(i, n):
            o = outs[i]
            for _ in range(n):
                solvers[i].solve_device(B, tx, t0, tp, o["z"], o["e"], o["i"], o["k"], o["o"], stream=streams[i].cuda_stream)

        def run_all(n):
            th = [threading.Thread(target=run, args=(i, n)) for i in range(S)]
            [t.start() for t in th]; [t.join() for t in th]
            torch.cuda.synchronize()

        run_all(3)
        K = 10
        t = time.perf_counter(); run_all(K); el = time.perf_counter() - t
        it = outs[0]["i"].cpu().numpy()
        print(f"{cfg} B={B} streams={S}: {1e3 * el / (K * S):.3f} ms per batch, {B * K * S / el / 1e6:.3f} M solves/s, iters mean {it.mean():.2f} max {it.max()}, "
              f"passes {solvers[0].last_passes()}", flush=True)
        for s in solvers:
            s.close()
