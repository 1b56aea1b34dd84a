#!/usr/bin/env python3
"""Development aid: instances still iterating after each pass in warm control steps of one cfg5 block
(RMPC_DUMP_HIST=1 python tests/tools/dev_fleet_hist.py [max_iter] [cfg] [B])."""
import os, sys
os.environ["RMPC_DUMP_HIST"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from robot_mpcs_amd import fleet
mi = int(sys.argv[1]) if len(sys.argv) > 1 else 20
name = sys.argv[2] if len(sys.argv) > 2 else "cfg4"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda", 0)
sh = fleet.MixedFleetShard({name: B}, dev, seed=7, options={"max_iter": mi, "acc_iters": 3})
sh.reset()
for i in range(30):
    if i in (0, 1, 2, 5, 10, 20, 29):
        sh.fleets[0]["s"].set_profiling(True)
        print("control step", i, file=sys.stderr, flush=True)
    sh.tick()
    if sh.fleets[0]["s"]._prof if hasattr(sh.fleets[0]["s"], "_prof") else True:
        sh.fleets[0]["s"].set_profiling(False)
    if i in (0, 1, 2, 5, 10, 20, 29):
        print("  stats [conv, acc, cap, fail, iters]", sh.stats(), file=sys.stderr, flush=True)
