"""Development aid: cost of one control step of each fleet of examples/fleet_loop.py alone and together."""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
import torch
import __graft_entry__ as ge
ge.build()
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import (BOXER_LIMITS, BOXER_LIMITS_U, PANDA_LIMITS, PANDA_LIMITS_U, POINT_LIMITS, POINT_LIMITS_U, make_scenario)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
MAXIT = int(sys.argv[1]) if len(sys.argv) > 1 else 6
fleet = []
for name, B, lim, limu in (("cfg2", 4096, POINT_LIMITS, POINT_LIMITS_U), ("cfg3", 3072, BOXER_LIMITS, BOXER_LIMITS_U), ("cfg4", 1024, PANDA_LIMITS, PANDA_LIMITS_U)):
    sc = make_scenario(name, B=B, seed=7)
    d = dict(sc.desc); d["options"] = dict(d["options"], max_iter=MAXIT)
    s = Solver(d, max_batch=B)
    ten = dict(goal=t(sc.extra["goal"]), r_body=t(np.full(B, sc.extra["r_body"])), lower_limits=t(np.tile(lim[0], (B, 1))), upper_limits=t(np.tile(lim[1], (B, 1))),
               lower_limits_u=t(np.tile(limu[0], (B, 1))), upper_limits_u=t(np.tile(limu[1], (B, 1))))
    if "obst_dyn" in sc.extra: ten["obst_dyn"] = t(sc.extra["obst_dyn"])
    else:
        rad = sc.extra.get("obst_radius", np.full(sc.extra["obst_pos"].shape[:2], 0.1))
        ten["obst"] = t(np.concatenate([sc.extra["obst_pos"], rad[:, :, None]], axis=2))
    N, nv = d["N"], s.nvar
    fleet.append(dict(name=name, B=B, s=s, scene=s.make_scene(sc.setup["mpc"]["weights"], **ten), x=t(sc.xinit), x0=t(sc.x0),
                      z=torch.empty((B, N, nv), dtype=torch.float64, device=dev), ef=torch.empty(B, dtype=torch.int32, device=dev),
                      it=torch.empty(B, dtype=torch.int32, device=dev), kkt=torch.empty(B, dtype=torch.float64, device=dev),
                      obj=torch.empty(B, dtype=torch.float64, device=dev), stream=torch.cuda.Stream(device=dev)))
torch.cuda.synchronize()
def one(f):
    st = f["stream"].cuda_stream
    f["s"].solve_scene_device(f["B"], f["scene"], f["x"], f["x0"], f["z"], f["ef"], f["it"], f["kkt"], f["obj"], stream=st)
    f["s"].advance_device(f["B"], f["z"], f["x"], f["x0"], previous_plan=True, stream=st)
for f in fleet:
    for _ in range(3): one(f); torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); one(f); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(f["name"], "alone ms", round(np.mean(ts), 2), "passes", f["s"].last_passes())
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    th = [threading.Thread(target=one, args=(f,)) for f in fleet]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("together ms", round(np.mean(ts), 2))
