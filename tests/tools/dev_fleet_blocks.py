#!/usr/bin/env python3
"""Development aid: control-step time of each block of the cfg5 shard alone and of the whole shard
(python tests/tools/dev_fleet_blocks.py [max_iter] [acc_iters])."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from robot_mpcs_amd import fleet

mi = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ai = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
full = {"cfg2": 4096, "cfg3": 3072, "cfg4": 1024}
for counts in ({"cfg2": 4096}, {"cfg3": 3072}, {"cfg4": 1024}, {"cfg2": 4096, "cfg3": 3072}, full):
    sh = fleet.MixedFleetShard(counts, dev, seed=7, options={"max_iter": mi, "acc_iters": ai})
    for _ in range(5):
        sh.tick()
    ts = []
    for i in range(80):
        if i % 40 == 0:
            sh.reset()
        t0 = time.perf_counter(); sh.tick(); ts.append(1e3 * (time.perf_counter() - t0))
    ts = np.array(ts)
    print(list(counts), "mean %.2f p50 %.2f p90 %.2f max %.2f first %.2f" % (ts.mean(), np.percentile(ts, 50), np.percentile(ts, 90), ts.max(), ts[0]), flush=True)
    sh.close()
