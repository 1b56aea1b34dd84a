#!/usr/bin/env python3
"""Headline benchmark: batched MPC solves/sec on MI355X.

A *step* is one pass of the hot path -- one ``rmpc_solve_batch_device`` call -- over one batch of synthetic
input that is already resident in HBM when the timed region starts.  Workload at N = 1 GPU: BASELINE.json
configs[1] (pointRobot, horizon 30, batch 4096, random start / goal + 3 static obstacles); with ``--gpus N``
every rank solves its own 4096 instances (weak scaling, no data-path collective; one RCCL all-reduce for the
max-over-ranks time and one all-gather of solve statistics).  Consecutive steps use different input sets
(``--input-sets``, default 6 x 55 MB per handle rotation) so that the inputs cannot live in the 256 MB
Infinity Cache.

Throughput mode: ``--streams S`` (default 4) solver handles per GPU, each on its own HIP stream, take the K steps
round robin; ``ms_per_step`` is elapsed / K, ``batch_latency_ms`` is one batch solved alone.

Prints ONE JSON line on rank 0 (contract in the task description) with
  ``roofline``      dominant kernel, EXCLUSIVE durations: a one-stream leg outside the timed region with HIP events
                    around every kernel of the handle's stream (so that kernel time x launches <= the leg's wall
                    time); algorithmic bytes = SURVEY.md 8(d)'s per-solve figure (compulsory I/O + iterations x
                    stage workspace) x the solves of one launch; ``traffic`` = HBM bytes per launch from the
                    committed PMC passes (profiles/);
  ``legs``          the other robots of the metric on the same box: panda (configs[3]) and boxer (configs[2]);
  ``cpu_baseline``  the CPU oracle port on this box's host cores, bounded sample, thread-scaling row.

``--config cfg5``: BASELINE configs[4], the mixed fleet as a device-resident closed loop (a step = one control step
of this rank's per-robot-type blocks, robot_mpcs_amd.fleet.MixedFleetShard).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_VECTOR_TFLOPS = 78.6
# what the counters (profiles/r0N_*_pmc.csv, *_phase_stamps.txt) show each kernel to be limited by; the HBM figures of
# `roofline` are the nominal accounting of SURVEY.md 8(d), not the limiter
BOUND_BY_COUNTERS = {
    "k_fused": "latency (one wavefront per SIMD: fp64 dependency chains + LDS round trips of the recursion; iterate streamed through L2 / Infinity Cache)",
    "k_fused_arm": "vector-instruction issue of one wavefront per SIMD (sweep: 15 k instructions per lane and pass, 40 of 64 lanes busy), then the recursion's dependency chains (7 x 7 factorisation) and LDS round trips",
    "k_sweep": "latency (memory requests at one wavefront per SIMD), then fp64 vector issue",
    "k_riccati": "latency (LDS round trips, 7x7 factorisation chain)",
    "k_step": "memory latency",
}
METRIC = "MPC solves/sec (batched) at N=30, pointRobot & panda; 1/2/4/8 MI355X"
WORKLOADS = {
    "cfg1": "BASELINE configs[0]: pointRobot N=10, 1 instance",
    "cfg2": "BASELINE configs[1]: pointRobot N=30, batch=4096 random start/goal + 3 static obstacles",
    "cfg3": "BASELINE configs[2]: boxer diff-drive N=30, batch=4096, 5 moving obstacles, slack",
    "cfg4": "BASELINE configs[3]: panda 7-DoF N=20, batch=1024, joint limits + sphere obstacle",
    "cfg5": "BASELINE configs[4]: mixed fleet, 8192 instances per GPU (4096 pointRobot + 3072 boxer + 1024 panda), "
            "device-resident closed loop, 100 Hz target",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="cfg2", help="cfg2 (headline), cfg3, cfg4, cfg1, cfg5 (mixed-fleet loop)")
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: BASELINE batch of the config)")
    ap.add_argument("--streams", type=int, default=4, help="solver handles per GPU, each on its own HIP stream")
    ap.add_argument("--input-sets", type=int, default=6, help="distinct resident input sets the steps rotate through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline config only (no panda / boxer legs)")
    ap.add_argument("--no-full-chip", action="store_true", help="skip the one-launch-fills-the-chip leg (profiles: keeps the kernel average to launches of one batch)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the exclusive per-kernel leg")
    ap.add_argument("--max-iter", type=int, default=40, help="cfg5 only: iteration cap of the real-time loop (what ends a slow solve is the pass budget)")
    ap.add_argument("--pass-budget", type=int, default=32, help="cfg5: real-time deadline of a solve in passes (horizon evaluations; 0 = none).  "
                    "What it cuts are line searches that backtrack and instances that need more than a couple of dozen iterations")
    ap.add_argument("--pass-budget-arm", type=int, default=16, help="cfg5: the arms' deadline if it differs (0 = --pass-budget).  The control step of "
                    "the shard hangs on this one: a pass of the arms is four kernel launches that queue behind the other fleets' fused launches")
    ap.add_argument("--pass-budget-boxer", type=int, default=56, help="cfg5: the boxers' deadline if it differs (0 = --pass-budget)")
    ap.add_argument("--acc-iters", type=int, default=3, help="cfg5: acceptable-termination window of the real-time loop (consecutive stagnant feasible iterations; the configs' default is 8)")
    ap.add_argument("--max-dwell", type=int, default=150, help="cfg5: control steps after which an instance takes its next goal even if it has not arrived")
    ap.add_argument("--mu-regoal-boxer", type=float, default=-1.0, help="cfg5 (development): barrier parameter a boxer's first solve after a goal hand-over restarts from (default: the fleet's)")
    ap.add_argument("--cfg5-steps", type=int, default=400, help="control steps of the cfg5 loop carried in the default line (extra.cfg5)")
    return ap.parse_args()


def survey_bytes_per_solve(d, iters_mean):
    """compulsory I/O + iterations x stage workspace exchanged once each way, and the Riccati +
    condensing flops per iteration."""
    N, nx, nh, npar = d["N"], d["nx"], int(d["nh"]), d["npar"]
    nv = nx + d["ns"] + d["nu"]
    nw = nv - nx
    io = 8 * (nx + 2 * N * nv + N * npar) + 16
    ws = 16 * N * (nx * nv + 2 * nx + 2 * nv + nv * (nv + 1) // 2 + nh + nh * nv)
    fl = N * (7.0 / 3.0 * nx ** 3 + 4 * nx ** 2 * nw + 2 * nx * nw ** 2 + nw ** 3 / 3.0) + 2.0 * N * nh * nv ** 2
    return io, ws, io + iters_mean * ws, fl


class Leg:
    """One config on this rank: S handles / streams, rotating resident inputs."""

    def __init__(self, cfg, B, S, nsets, dev, seed0):
        import numpy as np
        import torch
        from robot_mpcs_amd._lib import Solver
        from robot_mpcs_amd.scenarios import make_scenario
        self.torch, self.np, self.dev, self.cfg, self.B, self.S = torch, np, dev, cfg, B, S
        self.scs = [make_scenario(cfg, B=B, seed=seed0 + 17 * i) for i in range(nsets)]
        self.d = d = self.scs[0].desc
        self.N, self.nv = d["N"], d["nx"] + d["ns"] + d["nu"]
        dev_index = dev.index if dev.index is not None else 0
        self.solvers = [Solver(d, max_batch=B, device=dev_index) for _ in range(S)]
        self.inputs = [(torch.from_numpy(sc.xinit).to(dev), torch.from_numpy(sc.x0).to(dev), torch.from_numpy(sc.params).to(dev))
                       for sc in self.scs]
        f64, i32 = torch.float64, torch.int32
        self.outs = [dict(z=torch.empty((B, self.N, self.nv), dtype=f64, device=dev), exit=torch.empty(B, dtype=i32, device=dev),
                          iters=torch.empty(B, dtype=i32, device=dev), kkt=torch.empty(B, dtype=f64, device=dev),
                          obj=torch.empty(B, dtype=f64, device=dev)) for _ in range(S)]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        self.counter = 0
        torch.cuda.synchronize(dev)

    def _run(self, i, first, nsteps):
        o = self.outs[i]
        for j in range(nsteps):
            tx, t0, tp = self.inputs[(first + j * self.S) % len(self.inputs)]
            self.solvers[i].solve_device(self.B, tx, t0, tp, o["z"], o["exit"], o["iters"], o["kkt"], o["obj"],
                                         stream=self.streams[i].cuda_stream)

    def run(self, total):
        """total steps dealt round robin to the S handles; one host thread per handle (ctypes releases the GIL; the
        pass kernels of the arm poll a device counter from their thread)"""
        S = self.S
        counts = [total // S + (1 if i < total % S else 0) for i in range(S)]
        base = self.counter
        self.counter += total
        if S == 1:
            self._run(0, base, counts[0])
            return
        th = [threading.Thread(target=self._run, args=(i, base + i, counts[i])) for i in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    def exclusive(self, nsteps):
        """one handle, one stream, HIP events around every kernel: exclusive per-kernel durations and the batch latency"""
        torch = self.torch
        sv = self.solvers[0]
        # one call alone on an idle GPU, once per input set: SURVEY.md 8(d)'s "B / wall time of one rmpc_solve_batch"
        lat = []
        for j in range(len(self.inputs)):
            torch.cuda.synchronize(self.dev)
            t1 = time.perf_counter()
            self._run(0, self.counter + j, 1)
            torch.cuda.synchronize(self.dev)
            lat.append(1e3 * (time.perf_counter() - t1))
        self.counter += len(self.inputs) - 1
        self.single_call_ms = lat
        latency_ms = sum(lat) / len(lat)
        sv.set_profiling(True)
        t1 = time.perf_counter()
        self._run(0, self.counter + 1, nsteps)
        torch.cuda.synchronize(self.dev)
        wall_ms = 1e3 * (time.perf_counter() - t1)
        prof = sv.get_profile()
        sv.set_profiling(False)
        self.counter += nsteps + 1
        return latency_ms, wall_ms, prof

    def stats(self):
        o = self.outs[0]
        return o["exit"].cpu().numpy(), o["iters"].cpu().numpy(), o["kkt"].cpu().numpy()

    def close(self):
        for s in self.solvers:
            s.close()


def timed(leg, steps, warmup, fence):
    leg.run(max(warmup, leg.S))
    fence()
    t0 = time.perf_counter()
    leg.run(steps)
    fence()
    return time.perf_counter() - t0


def kernel_report(prof, wall_ms, nsteps, B, d, iters_mean):
    """per-kernel exclusive figures of the one-stream leg + the roofline object of the dominant kernel"""
    kern = {k: v for k, v in prof.items() if v["launches"] > 0 and v["total_ms"] > 0}
    if not kern:
        return None
    name = max(kern, key=lambda k: kern[k]["total_ms"])
    v = kern[name]
    avg_ms = v["total_ms"] / v["launches"]
    io, ws, per_solve, fl = survey_bytes_per_solve(d, iters_mean)
    if name in ("k_fused", "k_fused_arm"):
        # one launch = B whole solves: SURVEY 8(d)'s per-solve figure x B
        alg = B * per_solve
        note = name + " carries B whole solves per launch: algorithmic bytes = B x (IO + iters_mean x WS) of SURVEY.md 8(d)"
    else:
        alg = v["total_alg_bytes"] / v["launches"]   # pass kernels: bytes of the lanes still active, averaged (fill_lane_bytes)
        note = "pass kernel: algorithmic bytes of the lanes still iterating in each launch, averaged over the launches"
    achieved = alg / (avg_ms * 1e-3) / 1e9
    # HBM bytes per launch from the committed PMC passes -- only when they were collected with THIS library
    # (profiles/pmc_traffic.json records the source hash of the library its passes ran; another library: null)
    traffic, traffic_note = None, "no PMC passes committed for this configuration"
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            meta = tj.get("_meta", {}).get(d.get("_cfg", ""), {})
            if meta.get("library_source_hash") == d.get("_lib_hash"):
                traffic = tj.get(d.get("_cfg", ""), {}).get(name)
                traffic_note = meta.get("source", "")
            else:
                traffic_note = "profiles/pmc_traffic.json was collected with library %s, this run is library %s" % (
                    meta.get("library_source_hash"), d.get("_lib_hash"))
        except Exception:
            traffic = None
    return {
        # `bound`: the roofline `achieved` / `peak` / `frac` are priced against (the contract's HBM accounting of SURVEY.md
        # 8(d)); `bound_by_counters`: what the PMC passes and the phase stamps under profiles/ say limits the kernel
        "kernel": name, "bound": "hbm", "bound_by_counters": BOUND_BY_COUNTERS.get(name, "latency"),
        "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": traffic_note,
        "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms, "launches": int(v["launches"]),
        "exclusive_leg": {"steps": nsteps, "wall_ms": wall_ms, "kernel_ms_sum": sum(p["total_ms"] for p in kern.values()),
                          "note": "one handle on one stream, HIP events around every kernel; kernel time <= wall time"},
        "per_solve": {"compulsory_io_bytes": io, "workspace_bytes_per_iteration": ws, "iters_mean": iters_mean,
                      "flops_per_iteration": fl},
        "compulsory_io_GBps": B * io / (avg_ms * 1e-3) / 1e9 if name in ("k_fused", "k_fused_arm") else None,
        "note": note,
        "all_kernels": {k: {"total_ms": round(p["total_ms"], 3), "launches": int(p["launches"]),
                            "avg_ms": p["total_ms"] / p["launches"],
                            "alg_bytes_full_launch": int(p["full_launch_bytes"])} for k, p in kern.items()},
    }


def cpu_baseline(leg, gpu_z, gpu_exit, Solver, make_scenario, local_rank):
    """the oracle port (OpenMP over instances) on this box's host cores: bounded samples, thread-scaling row"""
    import numpy as np
    from oracle.oracle import Oracle
    from robot_mpcs_amd.fleet import flags_consistent
    cores = os.cpu_count() or 1
    # what this process may actually use: affinity mask and the cgroup CPU quota of the box (a GPU box hands a job a
    # share of the host's cores; the scaling row cannot go beyond it)
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = cores
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    usable = int(max(1, min(affinity, quota if quota else affinity)))
    sc = leg.scs[0]
    # the sample: input set 0 first (the GPU check below is element-wise on it), continued through the other input
    # sets of the same workload when a row needs more than one batch for its ~12 s of CPU work
    nsets = min(len(leg.scs), 4)
    xi_all = np.concatenate([c.xinit for c in leg.scs[:nsets]])
    x0_all = np.concatenate([c.x0 for c in leg.scs[:nsets]])
    pr_all = np.concatenate([c.params for c in leg.scs[:nsets]])
    o = Oracle(leg.d)
    o.solve_batch(sc.xinit[:8], sc.x0[:8], sc.params[:8], nthreads=1)
    t = time.perf_counter()
    r1 = o.solve_batch(sc.xinit[:24], sc.x0[:24], sc.params[:24], nthreads=1)
    per_solve_1t = (time.perf_counter() - t) / 24
    rows = []
    best = None
    for nt in sorted({1, max(1, usable // 2), usable, min(cores, 2 * usable)}):
        nb = int(min(len(xi_all), max(24, 12.0 * nt / per_solve_1t)))     # about 12 s of CPU work per row
        o.solve_batch(sc.xinit[: min(nb, 4 * nt)], sc.x0[: min(nb, 4 * nt)], sc.params[: min(nb, 4 * nt)], nthreads=nt)
        t = time.perf_counter()
        r = o.solve_batch(xi_all[:nb], x0_all[:nb], pr_all[:nb], nthreads=nt)
        el = time.perf_counter() - t
        rows.append({"threads": nt, "instances": nb, "seconds": el, "solves_per_s": nb / el})
        if best is None or nb / el > best[1] / best[2]:
            best = (nt, nb, el, r)
    nt, nb, el, r = best
    # element-wise check against the GPU plan of the same inputs (the sample, solved again here)
    s = Solver(leg.d, max_batch=nb, device=local_rank)
    g = s.solve(xi_all[:nb], x0_all[:nb], pr_all[:nb])
    s.close()
    conv = np.isin(r["exitflag"], (1, 2))
    dmax = float(np.abs(g["z"][conv] - r["z"][conv]).max()) if conv.any() else 0.0
    sc1 = make_scenario("cfg1", B=1, seed=0)
    o1 = Oracle(sc1.desc)
    o1.solve_batch(sc1.xinit, sc1.x0, sc1.params, nthreads=1)
    t = time.perf_counter()
    for _ in range(20):
        o1.solve_batch(sc1.xinit, sc1.x0, sc1.params, nthreads=1)
    cpu1_ms = 1e3 * (time.perf_counter() - t) / 20
    s1 = Solver(sc1.desc, max_batch=1, device=local_rank)
    s1.solve(sc1.xinit, sc1.x0, sc1.params)
    t = time.perf_counter()
    for _ in range(20):
        s1.solve(sc1.xinit, sc1.x0, sc1.params)
    gpu1_ms = 1e3 * (time.perf_counter() - t) / 20
    s1.close()
    return {
        # cores: what the run could actually occupy (threads beyond the cgroup quota share its CPUs)
        "value": nb / el, "unit": "solves/s", "cores": int(min(nt, usable)), "threads": nt, "kind": "port",
        "sample": f"{nb} instances of the same workload (input sets 0 .. {(nb - 1) // leg.B} of the timed region), one pass, OpenMP over instances; "
                  "the reference's own CPU path (FORCES Pro) cannot run here",
        "seconds": el, "thread_scaling": rows, "single_thread_ms_per_solve": 1e3 * per_solve_1t,
        "host": {"cpus": cores, "affinity": affinity, "cgroup_cpu_quota": quota, "usable": usable},
        "exitflags_consistent_with_gpu": bool(flags_consistent(g["exitflag"], r["exitflag"], g["kkt"], 1e-6)),
        "max_abs_diff_vs_gpu_plan": dmax,
        "single_instance_cfg1": {"cpu_port_1_thread_ms": cpu1_ms, "hip_host_entry_ms": gpu1_ms,
                                 "note": "BASELINE configs[0], one solve per call, host-pointer entry (PCIe included)"},
    }


def run_cfg5(args, fleet, dev, rank, world, dd, fence, steps, warmup, cdev=None):
    """BASELINE configs[4] on this rank's shard: the device-resident closed loop (scene packing + solve + plant step +
    shifted plan + warm multipliers + goal hand-over), timed like the headline (barrier + synchronize on both sides,
    max over ranks).  A STEADY loop: no episodes, no resets of the fleet -- an instance takes a new goal when it
    arrives, an instance whose solve failed goes back to its start state (rmpc_retarget_device); the statistics of
    every control step are summed on the device.  Returns value / ms_per_step / config / loop / per_fleet on rank 0."""
    import numpy as np
    part = fleet.partition_mixed(8192 * world, world)[rank]
    counts = {k: hi - lo for k, (lo, hi) in part.items()}
    shard = fleet.MixedFleetShard(counts, dev, seed=7 + rank, previous_plan=True, warm_duals=True,
                                  options={"max_iter": args.max_iter, "acc_iters": args.acc_iters},
                                  pass_budget={"cfg2": args.pass_budget, "cfg3": args.pass_budget_boxer or args.pass_budget,
                                               "cfg4": args.pass_budget_arm or args.pass_budget}, steady=True, max_dwell=args.max_dwell,
                                  mu_regoal=({"cfg3": args.mu_regoal_boxer} if args.mu_regoal_boxer >= 0 else None))
    for _ in range(warmup):
        shard.tick()
    shard.steady_stats(reset=True)
    fence()
    times = []
    # (a real-time loop in CPython keeps the cyclic garbage collector out of its control steps: with the objects of the
    #  other legs alive a generation-2 collection stalled one step in a few hundred for 27 ms)
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for i in range(steps):
        t1 = time.perf_counter()
        shard.tick()
        times.append(1e3 * (time.perf_counter() - t1))
    fence()
    gc.enable()
    elapsed = fleet.max_over_ranks(time.perf_counter() - t0, dd, cdev or dev)
    ss = shard.steady_stats(reset=True)
    NST = 13   # acc (5) + events (4) + more (4) per fleet
    st = np.array([np.concatenate([ss[k]["acc"], ss[k]["events"], ss[k]["more"]]) for k in ("cfg2", "cfg3", "cfg4")]).ravel()
    allst = fleet.gather_stats(st, dd, cdev or dev)
    shard.close()
    if rank != 0:
        return None
    times = np.array(times)
    total = 8192 * world
    per = {}
    for i, k in enumerate(("cfg2", "cfg3", "cfg4")):
        a = allst[:, NST * i: NST * i + NST].sum(axis=0)
        n = counts[k] * world * steps
        handovers = max(1.0, a[10])
        per[k] = {"instances_per_gpu": counts[k], "usable_share": float((a[0] + a[1]) / n),
                  "cut_by_deadline_or_iteration_cap_per_step": float(a[2] / steps), "failed_per_step": float(a[3] / steps),
                  "iters_mean": float(a[4] / n),
                  # goal hand-overs per control step: within the arrival tolerance / come to rest at the objective's
                  # equilibrium (the reference's N w / h term keeps a robot off a goal next to an obstacle) / dwell time-out
                  "arrivals_per_step": float(a[5] / steps), "settled_per_step": float(a[6] / steps),
                  "dwell_timeouts_per_step": float(a[7] / steps),
                  "mean_distance_to_goal_at_handover_m": float(a[9] / handovers),
                  # a failed solve keeps its state and drives on (mpcPlanner.py:263-264); a reset (back to the start state)
                  # only after %d failed control steps in a row
                  "resets_per_step": float(a[8] / steps), "reset_share_per_step": float(a[8] / n),
                  "share_of_instance_steps_inside_a_run_of_failed_solves": float(a[11] / n),
                  "resets_because_out_of_workspace_per_step": float(a[12] / steps)}
    return dict(value=total * steps / elapsed, ms_per_step=1e3 * elapsed / steps, steps=steps,
                config={"workload": WORKLOADS["cfg5"], "instances_per_gpu": 8192, "max_iter": args.max_iter,
                        "acc_iters": args.acc_iters, "pass_budget": args.pass_budget,
                        "pass_budget_boxer": args.pass_budget_boxer or args.pass_budget,
                        "pass_budget_arm": args.pass_budget_arm or args.pass_budget,
                        "loop": ("steady: new goal on arrival, on coming to rest or after %d control steps; a failed solve keeps its state "
                                 "(reset to the start state after %d failures in a row); the boxers' obstacles move every control step; "
                                 "no episodes" % (args.max_dwell, fleet.MixedFleetShard.FAIL_RESET_AFTER)),
                        "warm_start": "shifted plan + multipliers (rmpc_set_warm_start)",
                        "parallelism": f"{world} x per-robot-type blocks (fleet.partition_mixed), no data-path collective"},
                loop={"rate_hz": float(steps / elapsed), "ms_p50": float(np.percentile(times, 50)),
                      "ms_p90": float(np.percentile(times, 90)), "ms_p99": float(np.percentile(times, 99)), "ms_max": float(times.max()),
                      "deadline_10ms_hit_rate_rank0": float((times <= 10.0).mean()), "consecutive_steps": int(steps)},
                per_fleet=per)


def self_launch(args):
    """``python bench.py --gpus N`` typed without a launcher: start the N ranks the contract's launch line would start
    (``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...``) as a child process, BEFORE this process
    has touched a GPU, and leave with the child's exit code.  A plain run must never report ``n_gpus: 1`` for
    ``--gpus 8``."""
    import subprocess
    # (--standalone: the launcher's own c10d rendezvous on a free local port -- no port picked here and lost to
    #  another job between bind and launch)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: a launcher line must carry --gpus N "
                         f"(python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)")
    if not torch.cuda.is_available():
        # No HIP device: there is no CPU fallback.  Under a multi-rank launch the rendezvous and the rank / argument
        # plumbing are still exercised (gloo), so that a launch line can be checked on a CPU box; the exit is non-zero.
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo")
            seen = [None] * world
            dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank, "gpus": args.gpus, "steps": args.steps,
                                          "warmup": args.warmup, "config": args.config})
            if rank == 0:
                print(json.dumps({"error": "bench.py needs a HIP device (no CPU fallback)", "world_size": world, "ranks": seen}))
            dist.barrier()
            dist.destroy_process_group()
            raise SystemExit(3)
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # RMPC_BENCH_REHEARSE=1: every rank uses GPU 0 and the collectives run over gloo on host tensors -- the multi-rank
    # control flow of this file with real GPU work on a one-GPU box (a rehearsal of the launch, never a measurement)
    rehearse = world > 1 and os.environ.get("RMPC_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = torch.device("cpu") if rehearse else dev     # where the (tiny) collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from robot_mpcs_amd import _lib, fleet
    from robot_mpcs_amd._lib import Solver
    from robot_mpcs_amd.scenarios import DEFAULT_BATCH, make_scenario
    dd = dist if world > 1 else None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # which device this rank drives (an N-rank line must show N distinct devices)
    props = torch.cuda.get_device_properties(dev)
    ident = "%s %s pci %s" % (props.name, getattr(props, "gcnArchName", "?"), getattr(props, "pci_bus_id", "?"))
    if hasattr(props, "uuid"):
        ident += " uuid %s" % props.uuid
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    else:
        idents = [ident]
    cfg = args.config
    base = {"metric": METRIC, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "library_source_hash": _lib.source_hash(), "devices": idents}
    if rehearse:
        base["rehearsal"] = "%d ranks share GPU 0, collectives over gloo: launch plumbing only, NOT a multi-GPU measurement" % world

    # ------------------------------------------------------------------ cfg5: mixed-fleet closed loop
    if cfg == "cfg5":
        res = run_cfg5(args, fleet, dev, rank, world, dd, fence, args.steps, max(3, args.warmup), cdev)
        if rank == 0:
            print(json.dumps(dict(base, **res)))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ------------------------------------------------------------------ headline leg
    B = args.batch or DEFAULT_BATCH[cfg]
    S = max(1, min(args.streams, args.steps))
    leg = Leg(cfg, B, S, max(1, args.input_sets), dev, 1000 + 101 * rank)   # every rank owns different instances
    elapsed = timed(leg, args.steps, args.warmup, fence)
    elapsed_max = fleet.max_over_ranks(elapsed, dd, cdev)     # RCCL all-reduce(MAX)
    exitflag, iters, kkt = leg.stats()
    # RCCL all-gather of the solve statistics per rank (56 B): the only other collective
    allstats = fleet.gather_stats(fleet.solve_stats(exitflag, iters, kkt), dd, cdev)
    latency_ms, excl_wall_ms, prof = (None, None, None)
    n_excl = 0
    if not args.no_kernel_events:
        # whole rotations of the input sets, so that the exclusive average and the rocprofv3 average of a
        # --streams 1 run (profiles/) cover the same mix of instances
        n_excl = 2 * max(1, args.input_sets)
        latency_ms, excl_wall_ms, prof = leg.exclusive(n_excl)
    passes = leg.solvers[0].last_passes()
    full_chip = None
    if not args.no_kernel_events and not args.no_full_chip and cfg == "cfg2":
        # the same kernel with the chip full from ONE launch: a batch of S x B instances on one stream (what the
        # S overlapped streams of the timed region add up to), exclusive HIP-event duration
        big = Leg(cfg, B * S, 1, 2, dev, 3000 + 101 * rank)
        big.run(2)
        torch.cuda.synchronize(dev)
        _, bwall, bprof = big.exclusive(4)
        bef, bit, bk = big.stats()
        full_chip = (big.B, bwall, bprof, float(bit.mean()))
        big.close()

    legs = {}
    if not args.no_legs and cfg == "cfg2":
        # the other robots of the metric, same box, same run (shorter legs)
        for other in ("cfg4", "cfg3"):
            lo = Leg(other, DEFAULT_BATCH[other], S, 4, dev, 2000 + 101 * rank)
            # (at least 48 steps = 12 launches per stream: a stream's last launch ends with its slowest instances alone
            #  on the chip, and with four launches per stream that tail was a quarter of the leg -- cfg4 0.78 M over 16
            #  steps, 0.83-0.85 M over 50; the two legs together still take under half a second)
            k = max(48, args.steps // 4)
            el = fleet.max_over_ranks(timed(lo, k, 4, fence), dd, cdev)
            ef, it, kk = lo.stats()
            st = fleet.summarize(fleet.gather_stats(fleet.solve_stats(ef, it, kk), dd, cdev), lo.B)
            lat, wall, pr = lo.exclusive(4) if not args.no_kernel_events else (None, None, None)
            d2 = dict(lo.d, _cfg=other, _lib_hash=_lib.source_hash())
            legs[other] = {"workload": WORKLOADS[other], "value": lo.B * world * k / el, "unit": "solves/s", "steps": k,
                           "ms_per_step": 1e3 * el / k, "batch_latency_ms": lat,
                           "value_single_call": (lo.B * world / (lat * 1e-3)) if lat else None, "solve_stats": st,
                           "roofline": kernel_report(pr, wall, 4, lo.B, d2, st["iters_mean"]) if pr else None}
            lo.close()

    # BASELINE configs[4] (mixed fleet, 100 Hz loop) in the same line: 400 consecutive control steps of the steady loop
    # (`--config cfg5` runs it alone under the bench contract)
    cfg5_extra = None
    if not args.no_legs and cfg == "cfg2" and world == 1:
        cfg5_extra = run_cfg5(args, fleet, dev, rank, world, dd, fence, args.cfg5_steps, 10)

    if rank == 0:
        d = dict(leg.d, _cfg=cfg, _lib_hash=_lib.source_hash())
        stats = fleet.summarize(allstats, B)
        value = B * world * args.steps / elapsed_max
        out = dict(base, value=value, ms_per_step=1e3 * elapsed_max / args.steps, batch_latency_ms=latency_ms,
                   # `value` is the pipelined rate of the timed region (S launches in flight); this is SURVEY.md 8(d)'s
                   # definition, B / wall time of ONE call alone on the GPU (mean over the input sets), per GPU x world
                   value_single_call=(B * world / (latency_ms * 1e-3)) if latency_ms else None,
                   single_call_ms=getattr(leg, "single_call_ms", None),
                   single_call_ms_median=(float(np.median(leg.single_call_ms)) if getattr(leg, "single_call_ms", None) else None),
                   value_note=("value: %d solver handles on %d streams take the steps round robin (launches overlap); "
                               "value_single_call: one rmpc_solve_batch_device at a time" % (S, S)),
                   config={"workload": WORKLOADS[cfg], "batch_per_gpu": B, "horizon": leg.N, "nvar": leg.nv,
                           "npar": d["npar"], "nh": d["nh"],
                           "warm_start": "current_state (cold multipliers, every step solves from scratch)",
                           "tolerances": {k: d["options"][k] for k in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp", "max_iter")},
                           "parallelism": f"{world} x independent shards, no data-path collective",
                           "streams_per_gpu": S, "input_sets": len(leg.scs)},
                   solve_stats=dict(stats, passes_last_step=passes))
        if prof:
            rf = kernel_report(prof, excl_wall_ms, n_excl, B, d, stats["iters_mean"])
            if rf:
                io, ws, per_solve, fl = survey_bytes_per_solve(d, stats["iters_mean"])
                rf["solve_level"] = {
                    "bytes_per_solve": per_solve, "achieved_GBps": value * per_solve / 1e9 / world,
                    "frac_hbm": value * per_solve / 1e9 / world / HBM_PEAK_GBPS,
                    "achieved_fp64_TFLOPs": value * stats["iters_mean"] * fl / 1e12 / world,
                    "frac_fp64_vector": value * stats["iters_mean"] * fl / 1e12 / world / FP64_VECTOR_TFLOPS,
                    "note": "per GPU, throughput mode (all streams): SURVEY.md 8(d) bytes and flops per solve x solves/s",
                }
                if full_chip:
                    fb, fwall, fprof, fit = full_chip
                    fr = kernel_report(fprof, fwall, 4, fb, d, fit)
                    rf["full_chip_launch"] = {k: fr[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "launches", "exclusive_leg")}
                    rf["full_chip_launch"]["instances_per_launch"] = fb
                    rf["full_chip_launch"]["note"] = ("one launch of %d instances (= %d x the step's batch) alone on the GPU: the kernel with "
                                                      "every CU busy, as in the timed region where %d launches overlap" % (fb, S, S))
                # flat copies (a consumer that keeps only the scalar keys of `roofline` still sees them)
                rf["frac_fp64_vector"] = rf["solve_level"]["frac_fp64_vector"]
                rf["frac_hbm_timed_region"] = rf["solve_level"]["frac_hbm"]
                if full_chip:
                    rf["full_chip_frac"] = rf["full_chip_launch"]["frac"]
                    rf["full_chip_avg_launch_ms"] = rf["full_chip_launch"]["avg_launch_ms"]
                out["roofline"] = rf
        if legs:
            out["legs"] = legs
        if cfg5_extra:
            out["extra"] = {"cfg5": cfg5_extra}
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(leg, None, None, Solver, make_scenario, local_rank)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    leg.close()


if __name__ == "__main__":
    main()
