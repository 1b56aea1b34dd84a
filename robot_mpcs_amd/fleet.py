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


# BASELINE.json configs[4] / SURVEY.md 8(d): mix of the fleet, and the config each robot type runs
MIXED_FLEET = (("cfg2", 0.5), ("cfg3", 0.375), ("cfg4", 0.125))


def partition_mixed(total: int, world: int, mix=MIXED_FLEET):
    """Per-robot-type partition of a mixed fleet: every type is split over the ranks in
    contiguous blocks, so that each kernel launch on a GPU is homogeneous.  Returns, for every rank,
    ``{config: (lo, hi)}`` -- indices into that type's own instance list.  65536 instances on 8 GPUs:
    4096 point robots + 3072 boxers + 1024 pandas per GPU."""
    counts, acc = [], 0
    for i, (_, frac) in enumerate(mix):
        c = int(round(total * frac)) if i < len(mix) - 1 else total - acc
        counts.append(c)
        acc += c
    return [{name: shard_range(c, r, world) for (name, _), c in zip(mix, counts)} for r in range(world)]


N_STATS = 7


def solve_stats(exitflag, iters, kkt):
    """[converged (1), acceptable (2), iteration cap (0), failed (<0), sum iters, max iters, max KKT of converged]"""
    exitflag = np.asarray(exitflag); iters = np.asarray(iters); kkt = np.asarray(kkt)
    conv = exitflag == 1
    return np.array([conv.sum(), (exitflag == 2).sum(), (exitflag == 0).sum(), (exitflag < 0).sum(), iters.sum(),
                     iters.max(initial=0), kkt[conv].max(initial=0.0)], dtype=np.float64)


def flags_consistent(flag_a, flag_b, kkt_a, tol, slack: float = 2.0) -> bool:
    """Exit flags of two implementations of the same algorithm agree, allowing a converged (1) <->
    acceptable (2) flip only for instances whose KKT residual sits within ``slack`` x the tolerance
    (a rounding-level difference in the last residual decides which of the two stops fires first)."""
    flag_a = np.asarray(flag_a); flag_b = np.asarray(flag_b); kkt_a = np.asarray(kkt_a)
    diff = flag_a != flag_b
    if not diff.any():
        return True
    flip = np.isin(flag_a, (1, 2)) & np.isin(flag_b, (1, 2)) & (kkt_a <= slack * tol)
    return bool(np.all(~diff | flip))


def gather_stats(stats, dist=None, device=None):
    """all_gather of the per-rank statistics vector; returns (world, N_STATS)."""
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
        "converged": int(a[:, 0].sum()), "acceptable": int(a[:, 1].sum()), "iteration_cap": int(a[:, 2].sum()),
        "failed": int(a[:, 3].sum()),
        "iters_mean": float(a[:, 4].sum() / (instances_per_rank * a.shape[0])),
        "iters_max": int(a[:, 5].max()), "kkt_res_max_converged": float(a[:, 6].max()),
    }


# ---------------------------------------------------------------------------------------------------------
# One GPU's shard of the mixed fleet as a device-resident closed loop (BASELINE configs[4])
# ---------------------------------------------------------------------------------------------------------
class MixedFleetShard:
    """The per-robot-type blocks of one rank (``partition_mixed``), each with its own solver handle and HIP
    stream; ``tick()`` runs one control step of all of them: parameters expanded from the compact scene on the
    device (``rmpc_solve_batch_scene_device``, counterpart of the planner setters, mpcPlanner.py:83-210), solve,
    plant step with the model's ERK2 map and warm start of the next solve (``rmpc_advance_device``; shifted
    plan = ``shiftHorizon``, mpcPlanner.py:215-226).  Nothing crosses PCIe between control steps.  ``pass_budget``
    (one number, or one per block name): deadline of every solve in passes (instances still iterating return their
    last accepted iterate, flag 0)."""

    # steady loop: arrival tolerance of the end link [m] and goals per instance
    ARRIVE_TOL = {"cfg2": 0.25, "cfg3": 0.35, "cfg4": 0.10}
    POOL = 16
    # a robot that has come to rest on its goal has SETTLED (largest joint / wheel speed below this after SETTLE_MIN control
    # steps on the goal): with the reference's objective -- N w / h on the first row of a module, as written in
    # constraint_avoidance.py:22-31 -- a goal next to an obstacle is an equilibrium at a distance (an oracle closed loop of
    # the panda example's scene comes to rest 0.19 .. 0.43 m from its goal with every solve converged: tests/tools/
    # dev_oracle_arrival.py), not a point that is reached
    SETTLE_VEL = {"cfg2": 0.02, "cfg3": 0.02, "cfg4": 0.03}
    SETTLE_MIN = 20
    # a failed solve keeps its state and drives on (mpcPlanner.py:263-264); only this many failed control steps in a row
    # put an instance back to its start state
    FAIL_RESET_AFTER = 25
    # the moving obstacles of the boxers stay inside the arena [m] (they are mirrored at its walls)
    ARENA = 9.0
    # barrier parameter the first solve after a goal hand-over restarts from (rmpc_retarget_device)
    MU_REGOAL = {"cfg2": 0.0, "cfg3": 1e-1, "cfg4": 0.0}

    def __init__(self, counts: dict, device, seed: int = 7, previous_plan: bool = True, warm_duals: bool = True,
                 options: dict | None = None, pass_budget=0, steady: bool = False, max_dwell: int = 150,
                 mu_regoal: dict | None = None):
        import torch
        from robot_mpcs_amd._lib import Solver
        from robot_mpcs_amd import scenarios as sn
        self.torch, self.dev = torch, device
        self.previous_plan = bool(previous_plan)
        self.warm_duals = bool(warm_duals and previous_plan)
        # steady: no episodes -- an instance takes a new goal when it arrives (or after max_dwell control steps), an
        # instance whose solve failed goes back to its start state (rmpc_retarget_device); the loop runs indefinitely
        self.steady = bool(steady)
        self.max_dwell = int(max_dwell)
        self.mu_regoal = dict(self.MU_REGOAL, **(mu_regoal or {}))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)
        limits = {"cfg2": (sn.POINT_LIMITS, sn.POINT_LIMITS_U), "cfg3": (sn.BOXER_LIMITS, sn.BOXER_LIMITS_U),
                  "cfg4": (sn.PANDA_LIMITS, sn.PANDA_LIMITS_U)}
        self.fleets = []
        dev_index = device.index if device.index is not None else 0
        for name, B in counts.items():
            if B <= 0:
                continue
            lim, limu = limits[name]
            sc = sn.make_scenario(name, B=B, seed=seed)
            d = dict(sc.desc)
            if options:
                d["options"] = dict(d["options"], **options)
            s = Solver(d, max_batch=B, device=dev_index)
            s.set_warm_start(bool(warm_duals and previous_plan))
            pb = pass_budget.get(name, 0) if isinstance(pass_budget, dict) else int(pass_budget)
            if pb > 0:
                s.set_pass_budget(pb)   # real-time deadline of a control step (rmpc_set_pass_budget)
            ten = dict(goal=t(sc.extra["goal"]), r_body=t(np.full(B, sc.extra["r_body"])),
                       lower_limits=t(np.tile(lim[0], (B, 1))), upper_limits=t(np.tile(lim[1], (B, 1))),
                       lower_limits_u=t(np.tile(limu[0], (B, 1))), upper_limits_u=t(np.tile(limu[1], (B, 1))))
            if "obst_dyn" in sc.extra:
                ten["obst_dyn"] = t(sc.extra["obst_dyn"])
            else:
                rad = sc.extra.get("obst_radius", np.full(sc.extra["obst_pos"].shape[:2], 0.1))
                ten["obst"] = t(np.concatenate([sc.extra["obst_pos"], rad[:, :, None]], axis=2))
            N, nv = d["N"], s.nvar
            extra = {}
            if self.steady:
                pool = goal_pool(name, sc, self.POOL, seed + 1000)
                ten["goal"] = t(pool[:, 0])   # (the mobile robots' first goal is a local one too: goal_pool)
                extra = dict(pool=t(pool), cursor=torch.zeros(B, dtype=torch.int32, device=device),
                             dwell=torch.zeros(B, dtype=torch.int32, device=device),
                             failrun=torch.zeros(B, dtype=torch.int32, device=device),
                             # rmpc_retarget_device's counters (include/rmpc.h), summed over the control steps on the device
                             counts=torch.zeros(16, dtype=torch.int64, device=device))
            self.fleets.append(dict(extra, 
                name=name, B=B, s=s, sc=sc, scene=s.make_scene(sc.setup["mpc"]["weights"], **ten),
                x=t(sc.xinit), x0=t(sc.x0), x_start=t(sc.xinit), x0_start=t(sc.x0),
                z=torch.empty((B, N, nv), dtype=torch.float64, device=device),
                ef=torch.empty(B, dtype=torch.int32, device=device), it=torch.empty(B, dtype=torch.int32, device=device),
                kkt=torch.empty(B, dtype=torch.float64, device=device), obj=torch.empty(B, dtype=torch.float64, device=device),
                goal=ten["goal"], obst_dyn=ten.get("obst_dyn"), dt=float(d["dt"]),
                lim_lo=ten["lower_limits"], lim_hi=ten["upper_limits"],
                stream=torch.cuda.Stream(device=device)))   # (a higher dispatch priority for the arm's stream: no effect, measured twice)
        torch.cuda.synchronize(device)

    @property
    def instances(self) -> int:
        return sum(f["B"] for f in self.fleets)

    def _pack(self, f):
        f["s"].pack_scene_workspace(f["B"], f["scene"], stream=f["stream"].cuda_stream)

    def _solve(self, f):
        st = f["stream"].cuda_stream
        f["s"].solve_packed_device(f["B"], f["x"], f["x0"], f["z"], f["ef"], f["it"], f["kkt"], f["obj"], stream=st)
        f["s"].advance_device(f["B"], f["z"], f["x"], f["x0"], previous_plan=self.previous_plan, stream=st,
                              exitflag=f["ef"])
        if self.steady:
            f["s"].retarget_device(f["B"], f["x"], f["x0"], f["ef"], f["goal"], f["pool"], f["cursor"], f["dwell"], f["x_start"],
                                   self.ARRIVE_TOL[f["name"]], self.max_dwell, counts=f["counts"], iters=f["it"],
                                   mu_regoal=self.mu_regoal[f["name"]], stream=st, failrun=f["failrun"],
                                   fail_reset_after=self.FAIL_RESET_AFTER, settle_vel=self.SETTLE_VEL[f["name"]],
                                   settle_min_dwell=self.SETTLE_MIN, lower_limits=f["lim_lo"], upper_limits=f["lim_hi"])
            if f["obst_dyn"] is not None:
                # the world moves on: the next control step's scene predicts from the obstacles' new state
                # (updateDynamicObstacles(ob[nx:]), mpcPlanner.py:243-244,144-161)
                f["s"].advance_obstacles_device(f["obst_dyn"], f["dt"], arena=self.ARENA, stream=st)

    def tick(self, sync: bool = True):
        """One control step of the whole shard.  A fused solve fills every SIMD with one long-lived wavefront, and a
        small kernel enqueued behind it on another stream waits for a free slot (the boxers' scene packing 1.2 ms in
        the first version of this loop): so the scene packing of every block goes first, then the fused solves, the
        longest block first (they need no host look and are simply enqueued with their plant step), then the blocks on
        the pass kernels (the arm), whose host loop polls a counter, on the calling thread.  (The arm's first kernel still
        waits until the fused launches have no workgroup pending any more, 2.5 ms: the dispatcher serves the older
        queues first, whatever the stream priority; enqueuing the arm's passes first without a host look just runs
        the blocks one after the other.)"""
        for f in self.fleets:
            self._pack(f)
        fused = [f for f in self.fleets if f["s"].is_fused()]
        for f in sorted(fused, key=lambda f: -self._cost(f)):
            self._solve(f)
        # (under a pass budget the arm's passes are enqueued whole as well, rmpc_is_async: nothing below blocks)
        for f in self.fleets:
            if not f["s"].is_fused():
                self._solve(f)
        if sync:
            self.torch.cuda.synchronize(self.dev)

    @staticmethod
    def _cost(f):
        # expected duration of a block's launch: the diff-drive base's passes cost twice the point robot's
        return f["B"] * (2.0 if f["name"] == "cfg3" else 1.0)

    def reset(self):
        """New episode: every instance back to its start state with a cold plan and cold multipliers (the scenarios
        have no terminal set and a 1.5 s horizon: in a long run robots that arrive fast at an obstacle or at a limit
        end up in states from which the NLP is infeasible -- a property of the task, not of the solver; a fleet gets
        new goals long before)."""
        for f in self.fleets:
            # (x_start / x0_start were created on the default stream: __init__ synchronises the device behind them)
            with self.torch.cuda.stream(f["stream"]):
                f["x"].copy_(f["x_start"]); f["x0"].copy_(f["x0_start"])
            f["s"].set_warm_start(self.warm_duals)   # (re-arming forgets the stored multipliers)

    def stats(self):
        """per block: [converged, acceptable, iteration cap, failed, mean iterations] of the last control step"""
        out = {}
        cur = self.torch.cuda.current_stream(self.dev)
        for f in self.fleets:
            # the flags are written by kernels on the block's own stream: order the reads below behind them (after
            # tick(sync=False) nothing else does)
            cur.wait_stream(f["stream"])
            ef = f["ef"]
            out[f["name"]] = [int((ef == 1).sum()), int((ef == 2).sum()), int((ef == 0).sum()), int((ef < 0).sum()),
                              float(f["it"].float().mean())]
        return out

    def steady_stats(self, reset: bool = True):
        """steady loop, summed over the control steps since the last call (one host read per block):
        ``acc`` = [converged, acceptable, cut (iteration cap / deadline), failed, iterations],
        ``events`` = [arrivals, settled, dwell time-outs, resets],
        ``more`` = [sum of the distance to the goal at the hand-overs (m), hand-overs, instance-steps inside a run of failed
        solves, resets because the robot had left its workspace]"""
        out = {}
        cur = self.torch.cuda.current_stream(self.dev)
        for f in self.fleets:
            cur.wait_stream(f["stream"])
            c = f["counts"].cpu().numpy().astype(np.float64)
            out[f["name"]] = dict(acc=c[4:9].copy(), events=c[:4].copy(), more=np.array([c[9] * 1e-6, c[10], c[11], c[12]]))
            if reset:
                f["counts"].zero_()
                f["stream"].wait_stream(cur)
        return out

    def close(self):
        for f in self.fleets:
            f["s"].close()


def goal_pool(name: str, sc, P: int, seed: int):
    """[B, P, 3] goals per instance for the steady loop.  The mobile robots get LOCAL goals, a random walk of 2 .. 4 m steps
    (boxers 1.5 .. 3 m) that starts at the robot's start position, inside the arena and clear of the instance's obstacles
    -- what the reference's driver with a global planner hands the planner (the next waypoint of a path, get_local_goal,
    examples/boxer_example_global.py:203-212), not a task at the other end of the map: the scenarios' own goals are up to
    20 m away, and a 1.5 s horizon without a terminal set accelerates towards such a goal until it cannot stop in front of
    the workspace limits any more (oracle closed loop: a third of the point robots beyond their joint limits after 100
    control steps).  The arms get goals drawn like the scenario's own (goal 0 is the scenario's)."""
    rng = np.random.default_rng(seed)
    g0 = np.asarray(sc.extra["goal"], dtype=np.float64)
    B = g0.shape[0]
    pool = np.zeros((B, P, 3))
    pool[:, 0] = g0
    r_body = float(sc.extra["r_body"])
    if name == "cfg4":
        pool[:, 1:] = np.array([0.1, -0.6, 0.4]) + rng.uniform(-0.15, 0.15, size=(B, P - 1, 3))
        return pool
    start = np.asarray(sc.xinit, dtype=np.float64)[:, :2]
    if "obst_dyn" in sc.extra:
        opos = sc.extra["obst_dyn"].reshape(B, -1, 9)[:, :, :2]
        clear = np.full(opos.shape[:2], 0.1 + r_body + 1.0)
        lim, step = 8.5, (1.5, 3.0)
    else:
        opos = sc.extra["obst_pos"][:, :, :2]
        clear = sc.extra["obst_radius"] + r_body + 0.2
        lim, step = 8.0, (2.0, 4.0)
    for j in range(0, P):
        todo = np.ones(B, dtype=bool)
        while todo.any():
            idx = np.flatnonzero(todo)
            k = idx.size
            ang = rng.uniform(-np.pi, np.pi, size=k)
            ln = rng.uniform(step[0], step[1], size=k)
            prev = start[idx] if j == 0 else pool[idx, j - 1, :2]
            cand = prev + ln[:, None] * np.stack([np.cos(ang), np.sin(ang)], axis=1)
            ok = np.all(np.abs(cand) <= lim, axis=1)
            ok &= np.all(np.linalg.norm(cand[:, None, :] - opos[idx], axis=2) > clear[idx], axis=1)
            pool[idx[ok], j, :2] = cand[ok]
            todo[idx[ok]] = False
    return pool
