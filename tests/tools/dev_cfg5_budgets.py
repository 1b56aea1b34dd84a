#!/usr/bin/env python3
"""Development aid: the steady fleet loop under different per-fleet pass budgets (what the control step hangs on)."""
import gc
import sys
import time
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robot_mpcs_amd import fleet  # noqa: E402

counts = {"cfg2": 4096, "cfg3": 3072, "cfg4": 1024}
dev = torch.device("cuda:0")
for pb in ({"cfg2": 32, "cfg3": 56, "cfg4": 16}, {"cfg2": 32, "cfg3": 56, "cfg4": 20},
           {"cfg2": 32, "cfg3": 56, "cfg4": 24}):
    shard = fleet.MixedFleetShard(counts, dev, seed=7, previous_plan=True, warm_duals=True, options={"max_iter": 40, "acc_iters": 3},
                                  pass_budget=pb, steady=True, max_dwell=150)
    for _ in range(10):
        shard.tick()
    shard.steady_stats(reset=True)
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    ms = []
    steps = 600
    for _ in range(steps):
        t = time.perf_counter(); shard.tick(); ms.append(1e3 * (time.perf_counter() - t))
    torch.cuda.synchronize()
    gc.enable()
    ss = shard.steady_stats()
    ms = np.array(ms)
    use = {k: round((v["acc"][0] + v["acc"][1]) / (counts[k] * steps), 4) for k, v in ss.items()}
    print(pb, "ms mean %.2f p50 %.2f p99 %.2f max %.2f" % (ms.mean(), np.percentile(ms, 50), np.percentile(ms, 99), ms.max()), "usable", use, flush=True)
    shard.close()
