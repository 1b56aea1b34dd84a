#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE GPU's shard: a mixed fleet (4096 point robots + 3072 boxers +
1024 pandas = 8192 instances per GPU, 65536 over 8 GPUs) in a closed loop with a 100 Hz target.

Everything stays on the device between control steps (robot_mpcs_amd.fleet.MixedFleetShard):
parameters are expanded from compact scene descriptors, the plant step and the warm start of the
plan are rmpc_advance_device, the multipliers are warm-started by the solver handle
(rmpc_set_warm_start); the three blocks run on three HIP streams and are joined once per control
step (a fleet controller needs every action before the next tick).

    python examples/fleet_loop.py --steps 50 [--cold] [--max-iter 12] [--scale 0.25]

Prints one JSON line: loop rate, per-step wall time percentiles, share of steps inside the 10 ms
deadline, exitflag statistics.  Informational (the headline metric is bench.py; bench.py --config
cfg5 runs the same loop under the bench contract).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_loop(shard, steps, warmup):
    for _ in range(warmup):
        shard.tick()
    times, stats = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        shard.tick()
        times.append(1e3 * (time.perf_counter() - t0))
        stats.append(shard.stats())
    return np.array(times), stats


def summarize(shard, times, stats, extra=None):
    total = shard.instances
    per = {}
    for f in shard.fleets:
        a = np.array([s[f["name"]] for s in stats], dtype=float)
        per[f["name"]] = {"instances": f["B"], "converged": float(a[:, 0].mean()), "acceptable": float(a[:, 1].mean()),
                          "iteration_cap": float(a[:, 2].mean()), "failed": float(a[:, 3].mean()),
                          "usable_share": float((a[:, 0] + a[:, 1]).mean() / f["B"]), "iters_mean": float(a[:, 4].mean())}
    out = {
        "workload": "BASELINE configs[4], one GPU's shard: " + " + ".join(f"{f['B']} {f['name']}" for f in shard.fleets),
        "instances": total, "steps": len(times),
        "ms_per_control_step": {"mean": float(times.mean()), "p50": float(np.percentile(times, 50)),
                                "p90": float(np.percentile(times, 90)), "max": float(times.max())},
        "loop_rate_hz": float(1e3 / times.mean()), "solves_per_s": float(total * 1e3 / times.mean()),
        "deadline_10ms_hit_rate": float((times <= 10.0).mean()),
        "per_fleet": per,
    }
    out.update(extra or {})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--max-iter", type=int, default=0, help="iteration cap per solve (0 = the configs' 200)")
    ap.add_argument("--ls-max", type=int, default=0, help="step halvings allowed per line search (0 = the default 25)")
    ap.add_argument("--acc-iters", type=int, default=0, help="acceptable-termination window: consecutive stagnant feasible iterations (0 = the configs' 8)")
    ap.add_argument("--pass-budget", type=int, default=0, help="real-time deadline of a solve in passes (rmpc_set_pass_budget; 0 = none)")
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the 8192-instance shard")
    ap.add_argument("--cold", action="store_true", help="current-state initialisation and cold multipliers every step")
    ap.add_argument("--no-warm-duals", action="store_true", help="shifted plan, but cold multipliers")
    ap.add_argument("--only", default="", help="comma-separated subset of cfg2,cfg3,cfg4 (timing one block alone)")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    ge.build()
    from robot_mpcs_amd import fleet
    counts = {k: max(64, int((hi - lo) * args.scale)) for k, (lo, hi) in fleet.partition_mixed(8192, 1)[0].items()}
    if args.only:
        counts = {k: v for k, v in counts.items() if k in args.only.split(",")}
    opts = {}
    if args.max_iter > 0:
        opts["max_iter"] = args.max_iter
    if args.ls_max > 0:
        opts["ls_max"] = args.ls_max
    if args.acc_iters > 0:
        opts["acc_iters"] = args.acc_iters
    shard = fleet.MixedFleetShard(counts, torch.device("cuda:0"), previous_plan=not args.cold,
                                  warm_duals=not (args.cold or args.no_warm_duals), options=opts,
                                  pass_budget=args.pass_budget)
    times, stats = run_loop(shard, args.steps, args.warmup)
    print(json.dumps(summarize(shard, times, stats, {
        "max_iter": args.max_iter or 200, "ls_max": args.ls_max or 25, "acc_iters": args.acc_iters or 8, "pass_budget": args.pass_budget,
        "warm_start": "current_state, cold multipliers" if args.cold else
                      ("previous_plan" + ("" if args.no_warm_duals else " + multipliers"))})))
    shard.close()


if __name__ == "__main__":
    main()
