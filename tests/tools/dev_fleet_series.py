#!/usr/bin/env python3
"""Development aid: per control step time, passes and flag counts of one cfg5 block."""
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
for _ in range(3):
    sh.tick()
sh.reset()
for i in range(44):
    if i == 40:
        sh.reset()
    t0 = time.perf_counter(); sh.tick(); ms = 1e3 * (time.perf_counter() - t0)
    st = sh.stats()[name]
    f = sh.fleets[0]
    it = f["it"].cpu().numpy(); ef = f["ef"].cpu().numpy()
    print("step %2d  %.2f ms  passes %3d  [conv acc cap fail] %s  iters mean %.2f max %d  failed flags %s" % (
        i, ms, f["s"].last_passes(), st[:4], st[4], it.max(), dict(zip(*np.unique(ef[ef < 0], return_counts=True)))), flush=True)
