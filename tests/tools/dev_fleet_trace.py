#!/usr/bin/env python3
"""Development aid: a few control steps of the cfg5 shard at the real-time settings (for rocprofv3 --kernel-trace)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from robot_mpcs_amd import fleet
dev = torch.device("cuda", 0)
sh = fleet.MixedFleetShard({"cfg2": 4096, "cfg3": 3072, "cfg4": 1024}, dev, seed=7, options={"max_iter": 20, "acc_iters": 3},
                           pass_budget={"cfg2": 24, "cfg3": 40, "cfg4": 24})
sh.reset()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    t0 = time.perf_counter(); sh.tick(); print("tick %d %.2f ms" % (i, 1e3 * (time.perf_counter() - t0)), flush=True)
