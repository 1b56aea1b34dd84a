#!/usr/bin/env python3
"""Headline benchmark: batched MPC solves/sec on MI355X.

A *step* is one pass of the hot path -- one ``rmpc_solve_batch_device`` call --
over one batch of synthetic input that is already resident in HBM when the
timed region starts.  Workload at N = 1 GPU: BASELINE.json configs[1]
(pointRobot, horizon 30, batch 4096, random start / goal + 3 static obstacles);
with ``--gpus N`` every rank solves its own 4096 instances (weak scaling, no
data-path collective; one RCCL all-reduce for the max-over-ranks time and one
gather of solve statistics).

Throughput mode: ``--streams S`` (default 4) solver handles per GPU, each on its own HIP stream and
host thread, take the K steps round robin, so the latency-bound iteration tail of one batch (a few
straggler instances, tiny kernels) overlaps the bandwidth-bound bulk of the next; ``ms_per_step`` is
elapsed / K, ``batch_latency_ms`` is one batch solved alone.  ``--streams 1`` runs the steps back to back.

Prints ONE JSON line on rank 0 (contract in the task description) with two
extra objects: ``roofline`` for the dominant kernel (algorithmic bytes per
launch / average launch duration from HIP events on the solver's stream) and
``cpu_baseline`` (the CPU oracle port timed on this box's host cores on the
same inputs; a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
METRIC = "MPC solves/sec (batched) at N=30, pointRobot & panda; 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="cfg2", help="cfg2 (headline), cfg3, cfg4, cfg1")
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: BASELINE batch of the config)")
    ap.add_argument("--streams", type=int, default=4,
                    help="solver handles per GPU, each on its own HIP stream and host thread; steps are dealt round "
                         "robin, so the latency-bound iteration tail of one batch overlaps the bulk of the next")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not record per-kernel HIP events inside the timed region")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from robot_mpcs_amd import fleet
    from robot_mpcs_amd._lib import Solver
    from robot_mpcs_amd.scenarios import DEFAULT_BATCH, make_scenario

    cfg = args.config
    B = args.batch or DEFAULT_BATCH[cfg]
    sc = make_scenario(cfg, B=B, seed=1000 + rank)  # every rank owns different instances
    d = sc.desc
    N, nv = d["N"], d["nx"] + d["ns"] + d["nu"]
    S = max(1, min(args.streams, args.steps))
    solvers = [Solver(d, max_batch=B, device=local_rank) for _ in range(S)]

    # inputs resident in HBM before the timed region
    t_xinit = torch.from_numpy(sc.xinit).to(dev)
    t_x0 = torch.from_numpy(sc.x0).to(dev)
    t_params = torch.from_numpy(sc.params).to(dev)
    outs = []
    for _ in range(S):
        outs.append(dict(z=torch.empty((B, N, nv), dtype=torch.float64, device=dev),
                         exit=torch.empty(B, dtype=torch.int32, device=dev), iters=torch.empty(B, dtype=torch.int32, device=dev),
                         kkt=torch.empty(B, dtype=torch.float64, device=dev), obj=torch.empty(B, dtype=torch.float64, device=dev)))
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    torch.cuda.synchronize(dev)

    def run_steps(i, nsteps):
        o = outs[i]
        for _ in range(nsteps):
            solvers[i].solve_device(B, t_xinit, t_x0, t_params, o["z"], o["exit"], o["iters"], o["kkt"], o["obj"],
                                    stream=streams[i].cuda_stream)

    def run_all(total):
        # one host thread per handle (ctypes releases the GIL; a handle is driven by one thread only)
        import threading
        counts = [total // S + (1 if i < total % S else 0) for i in range(S)]
        if S == 1:
            run_steps(0, counts[0])
            return
        th = [threading.Thread(target=run_steps, args=(i, counts[i])) for i in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run_all(max(args.warmup, S))
    fence()
    # HIP events around every kernel of ONE of the S handles (all of its launches in the timed region):
    # events on all handles cost ~10 % of the throughput they are meant to measure
    solvers[0].set_profiling(not args.no_kernel_events)
    t0 = time.perf_counter()
    run_all(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    prof = {}
    for sv in solvers:
        for kname, v in sv.get_profile().items():
            acc = prof.setdefault(kname, dict(total_ms=0.0, launches=0, total_alg_bytes=0.0, full_launch_bytes=v["full_launch_bytes"]))
            acc["total_ms"] += v["total_ms"]; acc["launches"] += v["launches"]; acc["total_alg_bytes"] += v["total_alg_bytes"]
        sv.set_profiling(False)
    # latency of ONE batch solved alone (outside the timed region, reported for context)
    solvers[0].set_profiling(not args.no_kernel_events)
    t1 = time.perf_counter()
    run_steps(0, 1)
    torch.cuda.synchronize(dev)
    batch_latency_ms = 1e3 * (time.perf_counter() - t1)
    prof_solo = solvers[0].get_profile()
    solvers[0].set_profiling(False)
    passes = solvers[0].last_passes()
    solver = solvers[0]
    t_z, t_exit, t_iters, t_kkt = outs[0]["z"], outs[0]["exit"], outs[0]["iters"], outs[0]["kkt"]

    dd = dist if world > 1 else None
    elapsed_max = fleet.max_over_ranks(elapsed, dd, dev)   # RCCL all-reduce(MAX)

    exitflag = t_exit.cpu().numpy()
    iters = t_iters.cpu().numpy()
    # RCCL all-gather of six solve statistics per rank (48 B): the only other collective
    allstats = fleet.gather_stats(fleet.solve_stats(exitflag, iters, t_kkt.cpu().numpy()), dd, dev)

    if rank == 0:
        total_solves = B * world * args.steps
        value = total_solves / elapsed_max
        out = {
            "metric": METRIC,
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "batch_latency_ms": batch_latency_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": {"cfg1": "BASELINE configs[0]: pointRobot N=10, 1 instance",
                             "cfg2": "BASELINE configs[1]: pointRobot N=30, batch=4096 random start/goal + 3 static obstacles",
                             "cfg3": "BASELINE configs[2]: boxer diff-drive N=30, batch=4096, 5 moving obstacles, slack",
                             "cfg4": "BASELINE configs[3]: panda 7-DoF N=20, batch=1024, joint limits + sphere obstacle"}[cfg],
                "batch_per_gpu": B, "horizon": N, "nvar": nv, "npar": d["npar"], "nh": d["nh"],
                "warm_start": "current_state (cold multipliers, every step solves from scratch)",
                "tolerances": {k: d["options"][k] for k in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp", "max_iter")},
                "parallelism": f"{world} x independent shards, no data-path collective",
                "streams_per_gpu": S,
            },
            "solve_stats": dict(fleet.summarize(allstats, B), passes_last_step=passes),
        }
        # ---- roofline of the dominant kernel (rank 0's HIP events) -------------------
        if not args.no_kernel_events:
            loop = {k: v for k, v in prof.items() if k in ("k_sweep", "k_riccati", "k_step") and v["launches"] > 0}
            if loop:
                name = max(loop, key=lambda k: loop[k]["total_ms"])
                v = loop[name]
                avg_ms = v["total_ms"] / v["launches"]
                alg_per_launch = v["total_alg_bytes"] / v["launches"]  # active lanes only, averaged
                achieved = alg_per_launch / (avg_ms * 1e-3) / 1e9
                traffic = None
                tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
                if os.path.exists(tfile):
                    try:
                        traffic = json.load(open(tfile)).get(cfg, {}).get(name)
                    except Exception:
                        traffic = None
                # whole-solve figures with SURVEY.md 8(d)'s formulas: compulsory I/O + one stage workspace each way
                # per iteration through HBM, and the Riccati + condensing flop count per iteration
                nx_, nv_, nh_, np_ = d["nx"], nv, int(d.get("nh", out["config"]["nh"])), d["npar"]
                nu_w = nv_ - nx_
                it_avg = float(out["solve_stats"]["iters_mean"])
                io_b = 8 * (nx_ + 2 * N * nv_ + N * np_) + 16
                ws_b = 16 * N * (nx_ * nv_ + 2 * nx_ + 2 * nv_ + nv_ * (nv_ + 1) // 2 + nh_ + nh_ * nv_)
                fl_it = N * (7.0 / 3.0 * nx_ ** 3 + 4 * nx_ ** 2 * nu_w + 2 * nx_ * nu_w ** 2 + nu_w ** 3 / 3.0) + 2.0 * N * nh_ * nv_ ** 2
                solve_level = {
                    "bytes_per_solve": io_b + it_avg * ws_b, "achieved_GBps": value * (io_b + it_avg * ws_b) / 1e9,
                    "frac_hbm": value * (io_b + it_avg * ws_b) / 1e9 / HBM_PEAK_GBPS,
                    "flops_per_iteration": fl_it, "achieved_fp64_TFLOPs": value * it_avg * fl_it / 1e12,
                    "frac_fp64_vector": value * it_avg * fl_it / 1e12 / 78.6,
                    "note": "SURVEY.md 8(d): IO + iters_avg * WS bytes and Riccati + condensing flops per solve, times solves/s",
                }
                out["roofline"] = {
                    "solve_level": solve_level,
                    "kernel": name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg_per_launch,
                    "algorithmic_bytes_full_launch": int(v["full_launch_bytes"]),
                    "avg_launch_ms": avg_ms, "launches": int(v["launches"]),
                    "note": "bytes count only lanes still active in each launch; average over all launches of one of the "
                            "solver handles in the timed region (HIP events on its stream), in which the kernels of "
                            "several streams share the GPU (durations include that contention); solo_batch = the same "
                            "per-kernel figures for one batch solved alone",
                    "solo_batch": {k: {"avg_ms": p["total_ms"] / p["launches"],
                                       "alg_GBps": p["total_alg_bytes"] / (p["total_ms"] * 1e-3) / 1e9}
                                   for k, p in prof_solo.items() if p["launches"] and p["total_ms"] > 0},
                    "all_kernels": {k: {"total_ms": round(p["total_ms"], 3), "launches": int(p["launches"]),
                                        "avg_ms": (p["total_ms"] / p["launches"]) if p["launches"] else None,
                                        "alg_GBps": (p["total_alg_bytes"] / (p["total_ms"] * 1e-3) / 1e9)
                                        if p["launches"] and p["total_ms"] > 0 else None,
                                        "alg_bytes_full_launch": int(p["full_launch_bytes"])}
                                    for k, p in prof.items()},
                }
        # ---- CPU baseline: the oracle port on this box's host cores, same inputs ----
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            from oracle.oracle import Oracle
            cores = os.cpu_count() or 1
            o = Oracle(d)
            nb = min(B, 4096)
            o.solve_batch(sc.xinit[:64], sc.x0[:64], sc.params[:64], nthreads=cores)  # warm the threads
            tc = time.perf_counter()
            cpu = o.solve_batch(sc.xinit[:nb], sc.x0[:nb], sc.params[:nb], nthreads=cores)
            tcpu = time.perf_counter() - tc
            gz = t_z.cpu().numpy()[:nb]
            same = bool(np.array_equal(cpu["exitflag"], exitflag[:nb]))
            conv = cpu["exitflag"] == 1
            dmax = float(np.abs(gz[conv] - cpu["z"][conv]).max()) if conv.any() else 0.0
            # one instance, one thread, cfg1 -- the closest analogue of the reference's own loop
            # (one FORCES solve per control step on the CPU) -- and the same single solve through the HIP library
            sc1 = make_scenario("cfg1", B=1, seed=0)
            o1 = Oracle(sc1.desc)
            o1.solve_batch(sc1.xinit, sc1.x0, sc1.params, nthreads=1)
            t1c = time.perf_counter()
            for _ in range(20):
                o1.solve_batch(sc1.xinit, sc1.x0, sc1.params, nthreads=1)
            cpu1_ms = 1e3 * (time.perf_counter() - t1c) / 20
            s1 = Solver(sc1.desc, max_batch=1, device=local_rank)
            s1.solve(sc1.xinit, sc1.x0, sc1.params)
            t1g = time.perf_counter()
            for _ in range(20):
                s1.solve(sc1.xinit, sc1.x0, sc1.params)
            gpu1_ms = 1e3 * (time.perf_counter() - t1g) / 20
            s1.close()
            out["cpu_baseline"] = {
                "single_instance_cfg1": {"cpu_port_1_thread_ms": cpu1_ms, "hip_host_entry_ms": gpu1_ms,
                                         "note": "BASELINE configs[0], one solve per call, host-pointer entry (PCIe included)"},
                "value": nb / tcpu, "unit": "solves/s", "cores": cores, "kind": "port",
                "sample": f"{nb} instances of the same workload (rank 0's inputs), one pass, OpenMP over instances",
                "seconds": tcpu, "exitflags_equal_gpu": same, "max_abs_diff_vs_gpu_plan": dmax,
            }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for sv in solvers:
        sv.close()


if __name__ == "__main__":
    main()
