"""N > 1 path on CPU: two gloo ranks shard a batch, each 'solves' its shard
(the CPU oracle stands in for the GPU solver, which is a per-rank black box on
this path), statistics are gathered and the max-over-ranks time is reduced
exactly as bench.py does with RCCL.  No data-path collective exists."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle.oracle import Oracle
    from robot_mpcs_amd import fleet
    from robot_mpcs_amd.scenarios import make_scenario
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 23
    sc = make_scenario("cfg2", B=total, seed=77, time_horizon=12)  # same seed: every rank sees the same fleet
    lo, hi = fleet.shard_range(total, rank, world)
    r = Oracle(sc.desc).solve_batch(sc.xinit[lo:hi], sc.x0[lo:hi], sc.params[lo:hi], nthreads=2)
    kkt = np.maximum(r["res_stat"], r["res_comp"])
    allstats = fleet.gather_stats(fleet.solve_stats(r["exitflag"], r["iters"], kkt), dist)
    tmax = fleet.max_over_ranks(1.0 + rank, dist)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), lo=lo, hi=hi, z=r["z"], allstats=allstats, tmax=tmax,
             iters=r["iters"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_stats_gather(tmp_path, oracle_lib):
    world, port = 2, 29000 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle.oracle import Oracle
    from robot_mpcs_amd import fleet
    from robot_mpcs_amd.scenarios import make_scenario
    sc = make_scenario("cfg2", B=23, seed=77, time_horizon=12)
    full = Oracle(sc.desc).solve_batch(sc.xinit, sc.x0, sc.params, nthreads=2)
    parts = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    # shards are disjoint, contiguous, cover the fleet, and reproduce the unsharded solve bit for bit
    assert parts[0]["lo"] == 0 and parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == 23
    z = np.concatenate([p["z"] for p in parts])
    assert np.array_equal(z, full["z"])
    # every rank holds the same gathered statistics; max-over-ranks time is the slowest rank's
    assert np.array_equal(parts[0]["allstats"], parts[1]["allstats"]) and parts[0]["allstats"].shape == (2, fleet.N_STATS)
    s = fleet.summarize(parts[0]["allstats"], instances_per_rank=23 / 2)
    assert s["converged"] == int((full["exitflag"] == 1).sum()) and s["iters_max"] == int(full["iters"].max())
    assert abs(s["iters_mean"] - full["iters"].mean()) < 1e-12
    assert float(parts[0]["tmax"]) == 2.0 and float(parts[1]["tmax"]) == 2.0


def _mixed_worker(rank, world, port, outdir):
    """One rank of the mixed fleet (BASELINE configs[4]): per-robot-type blocks, one homogeneous solve per type,
    one gather of the statistics per type -- the collectives bench.py --config cfg5 issues over RCCL."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle.oracle import Oracle
    from robot_mpcs_amd import fleet
    from robot_mpcs_amd.scenarios import make_scenario
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 16
    part = fleet.partition_mixed(total, world)[rank]
    out = {}
    for cfg, (lo, hi) in part.items():
        n_type = fleet.partition_mixed(total, 1)[0][cfg][1]
        sc = make_scenario(cfg, B=n_type, seed=5, time_horizon=6)
        r = Oracle(sc.desc).solve_batch(sc.xinit[lo:hi], sc.x0[lo:hi], sc.params[lo:hi], nthreads=2)
        kkt = np.maximum(r["res_stat"], r["res_comp"])
        out[cfg + "_stats"] = fleet.gather_stats(fleet.solve_stats(r["exitflag"], r["iters"], kkt), dist)
        out[cfg + "_z"] = r["z"]
        out[cfg + "_span"] = np.array([lo, hi])
    np.savez(os.path.join(outdir, f"mixed{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_mixed_fleet_partition(tmp_path, oracle_lib):
    world, port = 2, 31000 + (os.getpid() % 2000)
    mp.spawn(_mixed_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle.oracle import Oracle
    from robot_mpcs_amd import fleet
    from robot_mpcs_amd.scenarios import make_scenario
    parts = [np.load(os.path.join(str(tmp_path), f"mixed{r}.npz")) for r in range(world)]
    counts = {cfg: hi for cfg, (lo, hi) in fleet.partition_mixed(16, 1)[0].items()}
    assert counts == {"cfg2": 8, "cfg3": 6, "cfg4": 2}
    for cfg, n_type in counts.items():
        sc = make_scenario(cfg, B=n_type, seed=5, time_horizon=6)
        full = Oracle(sc.desc).solve_batch(sc.xinit, sc.x0, sc.params, nthreads=2)
        assert parts[0][cfg + "_span"][0] == 0 and parts[0][cfg + "_span"][1] == parts[1][cfg + "_span"][0]
        assert parts[1][cfg + "_span"][1] == n_type
        assert np.array_equal(np.concatenate([p[cfg + "_z"] for p in parts]), full["z"])
        assert np.array_equal(parts[0][cfg + "_stats"], parts[1][cfg + "_stats"])
        s = fleet.summarize(parts[0][cfg + "_stats"], instances_per_rank=n_type / 2)
        assert s["converged"] + s["acceptable"] == int(np.isin(full["exitflag"], (1, 2)).sum())


def test_partition_mixed_is_the_survey_partition():
    from robot_mpcs_amd.fleet import partition_mixed
    parts = partition_mixed(65536, 8)
    assert len(parts) == 8
    for r, p in enumerate(parts):
        assert {k: hi - lo for k, (lo, hi) in p.items()} == {"cfg2": 4096, "cfg3": 3072, "cfg4": 1024}
        assert p["cfg2"][0] == r * 4096 and p["cfg3"][0] == r * 3072 and p["cfg4"][0] == r * 1024
    one = partition_mixed(8192, 1)[0]
    assert {k: hi - lo for k, (lo, hi) in one.items()} == {"cfg2": 4096, "cfg3": 3072, "cfg4": 1024}


def test_shard_range_properties():
    from robot_mpcs_amd.fleet import shard_range
    for total in (1, 7, 64, 4096, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_launch_line_plumbing_without_a_device(tmp_path):
    """The driver's multi-GPU launch line (python -m torch.distributed.run ... bench.py --gpus N ...) on a box without a
    HIP device: rendezvous, RANK / LOCAL_RANK / WORLD_SIZE and the arguments reach every rank, then the program stops
    with a non-zero exit and says why -- it never falls back to a CPU solver."""
    import json
    import subprocess
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    port = 33000 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert line, r.stdout + r.stderr
    d = json.loads(line[-1])
    assert "HIP device" in d["error"] and d["world_size"] == 2
    assert sorted(x["rank"] for x in d["ranks"]) == [0, 1] and all(x["gpus"] == 2 and x["steps"] == 3 for x in d["ranks"])


def test_bench_gpus_n_without_a_launcher_starts_n_ranks(tmp_path):
    """``python bench.py --gpus 2`` typed without torch.distributed.run: the program starts the two ranks itself (before
    it touches a GPU) instead of measuring one GPU and printing ``n_gpus: 1``; ``--gpus`` that contradicts WORLD_SIZE is
    refused."""
    import json
    import subprocess
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert line, r.stdout + r.stderr
    d = json.loads(line[-1])
    assert d["world_size"] == 2 and sorted(x["rank"] for x in d["ranks"]) == [0, 1]
    assert all(x["gpus"] == 2 for x in d["ranks"])
    # a launcher's WORLD_SIZE that contradicts --gpus is an error, also for --gpus 1
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                        timeout=120, cwd=ROOT, env=env2)
    assert r2.returncode != 0 and "WORLD_SIZE" in (r2.stderr + r2.stdout)
