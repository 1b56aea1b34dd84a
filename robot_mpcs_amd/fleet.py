"""Sharding of independent MPC instances over ranks (one process per GPU).

The solve path has no exchange step: every rank owns a
contiguous block of instances and runs its own solver handle.  The only
collectives are report-level: max over ranks of the elapsed time and a gather
of a few solve statistics (RCCL on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def shard_range(total: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank ``rank``; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def solve_stats(exitflag, iters, kkt):
    exitflag = np.asarray(exitflag); iters = np.asarray(iters); kkt = np.asarray(kkt)
    conv = exitflag == 1
    return np.array([conv.sum(), (exitflag == 0).sum(), (exitflag < 0).sum(), iters.sum(), iters.max(initial=0),
                     kkt[conv].max(initial=0.0)], dtype=np.float64)


def gather_stats(stats, dist=None, device=None):
    """all_gather of the per-rank statistics vector; returns (world, 6)."""
    import torch
    t = torch.as_tensor(np.asarray(stats, dtype=np.float64), device=device)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return t.cpu().numpy()[None]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()


def max_over_ranks(value: float, dist=None, device=None) -> float:
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def summarize(allstats, instances_per_rank: int):
    a = np.asarray(allstats)
    return {
        "converged": int(a[:, 0].sum()), "iteration_cap": int(a[:, 1].sum()), "failed": int(a[:, 2].sum()),
        "iters_mean": float(a[:, 3].sum() / (instances_per_rank * a.shape[0])),
        "iters_max": int(a[:, 4].max()), "kkt_res_max_converged": float(a[:, 5].max()),
    }
