#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE GPU's shard: a mixed fleet (4096 point robots + 3072 boxers +
1024 pandas = 8192 instances per GPU, 65536 over 8 GPUs) in a closed loop with a 100 Hz target.

Everything stays on the device between control steps: parameters are expanded from compact scene
descriptors (rmpc_solve_batch_scene_device), the plant step and the warm start are
rmpc_advance_device; the three sub-fleets run on three HIP streams from three host threads and
are joined once per control step (a fleet controller needs every action before the next tick).

    python examples/fleet_loop.py --steps 50 [--max-iter 12] [--scale 0.25]

Prints one JSON line: loop rate, per-step wall time percentiles, share of steps inside the 10 ms
deadline, exitflag statistics.  Informational (the headline metric is bench.py).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--max-iter", type=int, default=0, help="iteration cap per solve (0 = the configs' 200)")
    ap.add_argument("--ls-max", type=int, default=0, help="step halvings allowed per line search (0 = the default 25)")
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the 8192-instance shard")
    ap.add_argument("--previous-plan", action="store_true", help="warm start from the shifted plan")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    ge.build()
    from robot_mpcs_amd._lib import Solver
    from robot_mpcs_amd.scenarios import (BOXER_LIMITS, BOXER_LIMITS_U, PANDA_LIMITS, PANDA_LIMITS_U, POINT_LIMITS,
                                          POINT_LIMITS_U, make_scenario)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    fleet = []
    for name, B0, lim, limu in (("cfg2", 4096, POINT_LIMITS, POINT_LIMITS_U), ("cfg3", 3072, BOXER_LIMITS, BOXER_LIMITS_U),
                                ("cfg4", 1024, PANDA_LIMITS, PANDA_LIMITS_U)):
        B = max(64, int(B0 * args.scale))
        sc = make_scenario(name, B=B, seed=7)
        d = dict(sc.desc)
        if args.max_iter > 0:
            d["options"] = dict(d["options"], max_iter=args.max_iter)
        if args.ls_max > 0:
            d["options"] = dict(d["options"], ls_max=args.ls_max)
        s = Solver(d, max_batch=B)
        ten = dict(goal=t(sc.extra["goal"]), r_body=t(np.full(B, sc.extra["r_body"])),
                   lower_limits=t(np.tile(lim[0], (B, 1))), upper_limits=t(np.tile(lim[1], (B, 1))),
                   lower_limits_u=t(np.tile(limu[0], (B, 1))), upper_limits_u=t(np.tile(limu[1], (B, 1))))
        if "obst_dyn" in sc.extra:
            ten["obst_dyn"] = t(sc.extra["obst_dyn"])
        else:
            rad = sc.extra.get("obst_radius", np.full(sc.extra["obst_pos"].shape[:2], 0.1))
            ten["obst"] = t(np.concatenate([sc.extra["obst_pos"], rad[:, :, None]], axis=2))
        N, nv = d["N"], s.nvar
        fleet.append(dict(name=name, B=B, s=s, scene=s.make_scene(sc.setup["mpc"]["weights"], **ten),
                          x=t(sc.xinit), x0=t(sc.x0), z=torch.empty((B, N, nv), dtype=torch.float64, device=dev),
                          ef=torch.empty(B, dtype=torch.int32, device=dev), it=torch.empty(B, dtype=torch.int32, device=dev),
                          kkt=torch.empty(B, dtype=torch.float64, device=dev), obj=torch.empty(B, dtype=torch.float64, device=dev),
                          stream=torch.cuda.Stream(device=dev)))
    torch.cuda.synchronize()

    def one(f):
        st = f["stream"].cuda_stream
        f["s"].solve_scene_device(f["B"], f["scene"], f["x"], f["x0"], f["z"], f["ef"], f["it"], f["kkt"], f["obj"], stream=st)
        f["s"].advance_device(f["B"], f["z"], f["x"], f["x0"], previous_plan=args.previous_plan, stream=st)

    def tick():
        th = [threading.Thread(target=one, args=(f,)) for f in fleet]
        for x in th:
            x.start()
        for x in th:
            x.join()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tick()
    times, stats = [], []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        tick()
        times.append(1e3 * (time.perf_counter() - t0))
        stats.append([(int((f["ef"] == 1).sum()), int((f["ef"] == 2).sum()), int((f["ef"] == 0).sum()), int((f["ef"] < 0).sum()),
                       float(f["it"].float().mean())) for f in fleet])
    times = np.array(times)
    st = np.array(stats)  # (steps, 3, 5)
    total = sum(f["B"] for f in fleet)
    out = {
        "workload": "BASELINE configs[4], one GPU's shard: " + " + ".join(f"{f['B']} {f['name']}" for f in fleet),
        "instances": total, "steps": args.steps, "max_iter": args.max_iter or 200, "ls_max": args.ls_max or 25,
        "warm_start": "previous_plan" if args.previous_plan else "current_state",
        "ms_per_control_step": {"mean": float(times.mean()), "p50": float(np.percentile(times, 50)),
                                "p90": float(np.percentile(times, 90)), "max": float(times.max())},
        "loop_rate_hz": float(1e3 / times.mean()), "solves_per_s": float(total * 1e3 / times.mean()),
        "deadline_10ms_hit_rate": float((times <= 10.0).mean()),
        "per_fleet": {f["name"]: {"converged": float(st[:, i, 0].mean()), "acceptable": float(st[:, i, 1].mean()),
                                  "iteration_cap": float(st[:, i, 2].mean()), "failed": float(st[:, i, 3].mean()),
                                  "iters_mean": float(st[:, i, 4].mean())} for i, f in enumerate(fleet)},
    }
    print(json.dumps(out))
    for f in fleet:
        f["s"].close()


if __name__ == "__main__":
    main()
