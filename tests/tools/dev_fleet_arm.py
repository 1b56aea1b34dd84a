#!/usr/bin/env python3
"""Development aid: the arm's block of the cfg5 shard alone (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from robot_mpcs_amd import fleet
mi = int(sys.argv[1]) if len(sys.argv) > 1 else 20
name = sys.argv[2] if len(sys.argv) > 2 else "cfg4"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda", 0)
sh = fleet.MixedFleetShard({name: B}, dev, seed=7, options={"max_iter": mi, "acc_iters": 3})
for _ in range(5):
    sh.tick()
sh.reset()
ts = []
for i in range(40):
    t0 = time.perf_counter(); sh.tick(); ts.append(1e3 * (time.perf_counter() - t0))
print("ms per tick: mean %.2f p50 %.2f" % (np.mean(ts), np.percentile(ts, 50)), "passes", sh.fleets[0]["s"].last_passes() if hasattr(sh.fleets[0]["s"], "last_passes") else "?")
