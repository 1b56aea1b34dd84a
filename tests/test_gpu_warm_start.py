"""Warm start of the multipliers (rmpc_set_warm_start): closed loops on the GPU against the oracle's
``orc_solve_warm`` stepping the same instances, and the mixed fleet of BASELINE configs[4] on one GPU's shard."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-6


@pytest.fixture(scope="module")
def rt():
    import __graft_entry__ as g
    g.build()
    from oracle.oracle import Oracle
    from robot_mpcs_amd._lib import Solver
    from robot_mpcs_amd.scenarios import make_scenario
    return dict(Oracle=Oracle, Solver=Solver, make_scenario=make_scenario)


@pytest.mark.parametrize("name,B,steps", [("cfg2", 24, 8), ("cfg3", 24, 8), ("cfg4", 8, 6), ("wc_boxer", 16, 6)])
def test_warm_started_closed_loop_matches_oracle(rt, name, B, steps):
    """previous_plan initialisation + multiplier warm start, plant = the model's ERK2 map.  Every step: same exit
    flags (1 <-> 2 flips only at the tolerance), applied control within 1e-6, and the warm start pays:
    fewer iterations than the cold first step."""
    from robot_mpcs_amd.fleet import flags_consistent
    sc = rt["make_scenario"](name, B=B, seed=41)
    o = rt["Oracle"](sc.desc)
    s = rt["Solver"](sc.desc, max_batch=B)
    s.set_warm_start(True)
    nx, nv, N, nxs = o.nx, o.nv, o.N, o.nx + o.ns
    x = sc.xinit.copy()
    x0 = sc.x0.copy()
    duals = [None] * B
    zc = [None] * B
    it_first = it_later = 0
    for t in range(steps):
        g = s.solve(x, x0, sc.params)
        cz = np.zeros_like(g["z"]); cf = np.zeros(B, dtype=np.int32); ci = np.zeros(B, dtype=np.int32)
        for b in range(B):
            r = o.solve_warm(x[b], x0[b], sc.params[b], duals[b])
            cz[b] = r["z"]; cf[b] = r["exitflag"]; ci[b] = r["iters"]
            # (a failed solve leaves zero multipliers and mu0 behind, on both sides)
            duals[b] = r["duals"] if r["exitflag"] >= 0 else (np.zeros((N, o.m)), np.zeros((N, nx)), sc.desc["options"]["mu0"] / 1000.0)
        assert flags_consistent(g["exitflag"], cf, g["kkt"], 1e-6), (t, g["exitflag"], cf)
        ok = np.isin(cf, (1, 2))
        du = np.abs(g["z"][:, 0, nxs:] - cz[:, 0, nxs:]).max(axis=1)
        us = np.maximum(1.0, np.abs(cz[:, 0, nxs:]).max(axis=1))
        assert np.all(du[ok] <= TOL * us[ok]), (t, du.max())
        assert (g["iters"] == ci).mean() >= 0.9, (t, g["iters"], ci)
        if t == 0:
            it_first = g["iters"].mean()
        else:
            it_later += g["iters"].mean() / (steps - 1)
        # plant + shifted plan (device counterpart: rmpc_advance_device), from the ORACLE's plan so that both sides
        # keep solving identical problems
        for b in range(B):
            x[b] = o.dynamics(x[b], cz[b, 0, nxs:])
            x0[b] = np.concatenate([cz[b, 1:], cz[b, -1:]])
    s.close()
    assert it_later < 0.85 * it_first, (it_first, it_later)


def test_warm_start_is_per_instance_and_forgets_on_mode_change(rt):
    sc = rt["make_scenario"]("cfg2", B=32, seed=5)
    s = rt["Solver"](sc.desc, max_batch=32)
    cold = s.solve(sc.xinit, sc.x0, sc.params)
    s.set_warm_start(True)
    a = s.solve(sc.xinit, sc.x0, sc.params)              # no multipliers stored yet: cold
    assert np.array_equal(a["z"], cold["z"]) and np.array_equal(a["iters"], cold["iters"])
    b = s.solve(sc.xinit, cold["z"], sc.params)           # warm: from the solution and its multipliers
    assert np.all(b["exitflag"] >= 1) and b["iters"].mean() < 0.6 * cold["iters"].mean()
    # the same local solution up to the solver's tolerances (flat directions of the later stages move more)
    assert np.abs(b["z"][:, 0] - cold["z"][:, 0]).max() <= 1e-3
    assert np.all(np.abs(b["obj"] - cold["obj"]) <= 1e-5 * np.maximum(1.0, np.abs(cold["obj"])))
    s.set_warm_start(False)
    c = s.solve(sc.xinit, sc.x0, sc.params)
    assert np.array_equal(c["z"], cold["z"]) and np.array_equal(c["iters"], cold["iters"])
    s.close()


@pytest.mark.parametrize("name,B", [("cfg2", 64), ("cfg3", 64), ("cfg4", 32)])
def test_warm_start_recovers_after_a_poisoned_solve(rt, name, B):
    """A warm-started handle whose instance 4 gets a NaN start state for ONE control step: that solve fails (flag < 0), its
    multipliers are not kept, and the next solve from clean inputs converges again to the cold solve's objective; every other
    instance goes through the three steps bit for bit like a twin handle that never saw the NaN."""
    sc = rt["make_scenario"](name, B=B, seed=15)
    a = rt["Solver"](sc.desc, max_batch=B); a.set_warm_start(True)
    b = rt["Solver"](sc.desc, max_batch=B); b.set_warm_start(True)
    cold = a.solve(sc.xinit, sc.x0, sc.params)
    b.solve(sc.xinit, sc.x0, sc.params)
    xi = sc.xinit.copy(); xi[4, 1] = np.nan
    bad = a.solve(xi, cold["z"], sc.params)
    twin2 = b.solve(sc.xinit, cold["z"], sc.params)
    assert bad["exitflag"][4] < 0 and bad["iters"][4] == 0
    others = np.arange(B) != 4
    for k in ("z", "exitflag", "iters"):
        assert np.array_equal(bad[k][others], twin2[k][others]), k
    rec = a.solve(sc.xinit, cold["z"], sc.params)
    twin3 = b.solve(sc.xinit, cold["z"], sc.params)
    a.close(); b.close()
    assert rec["exitflag"][4] >= 1
    assert abs(rec["obj"][4] - cold["obj"][4]) <= 1e-5 * max(1.0, abs(cold["obj"][4]))
    for k in ("z", "exitflag", "iters"):
        assert np.array_equal(rec[k][others], twin3[k][others]), k


def test_mixed_fleet_shard_closed_loop(rt):
    """BASELINE configs[4] ("mixed fleet ... 100 Hz real-time loop"), one GPU's shard in the SURVEY 8(e) proportions
    (4 : 3 : 1, scaled down to 1024 instances so that the oracle can follow a sample), device-resident loop:
    scene packing + solve + plant step + shifted plan + warm multipliers, 24 control steps, no iteration cap below
    the configs' 200.  Bars per robot type: usable plans (exitflag 1 or 2) for >= 95 % of the instances in every
    step after the first, no failures; a sample of instances is re-solved by the oracle from the device's own
    inputs of the last step."""
    import torch
    from robot_mpcs_amd import fleet
    counts = {"cfg2": 512, "cfg3": 384, "cfg4": 128}
    dev = torch.device("cuda:0")
    shard = fleet.MixedFleetShard(counts, dev, seed=11, previous_plan=True, warm_duals=True)
    assert shard.instances == 1024
    iters = {f["name"]: [] for f in shard.fleets}
    for step in range(24):
        if step == 23:   # inputs of the last control step, for the oracle
            snap = {f["name"]: (f["x"].cpu().numpy().copy(), f["x0"].cpu().numpy().copy()) for f in shard.fleets}
        shard.tick()
        for name, (c1, c2, c0, neg, itmean) in shard.stats().items():
            B = counts[name]
            # (a shifted plan can touch an inverse-barrier row -- h <= 0 by rounding -- and stop with -7: such an
            #  instance restarts from its state in the next step, rmpc_advance_device_flags)
            assert neg <= 0.03 * B, (step, name, neg)
            if step >= 1:
                assert (c1 + c2) >= 0.95 * B, (step, name, c1, c2, c0)
            iters[name].append(itmean)
    for name, it in iters.items():
        # the warm start pays: about half the iterations for the point robot and the boxer, and -- with the second
        # derivatives of its kinematics in the curvature terms -- less than half for the arm (Gauss-Newton blocks alone
        # left it crawling to the acceptable-termination test: more iterations warm than cold)
        assert np.mean(it[2:]) < (0.8 if name != "cfg4" else 0.6) * it[0], (name, it[0], np.mean(it[2:]))
    # last step against the oracle started from the same state, plan and (device) multipliers is not possible
    # without exporting the multipliers; instead: the oracle, cold-started from the same state and shifted plan,
    # reaches the same plan (same NLP, same basin) within the solver tolerances on a sample
    for f in shard.fleets:
        o = rt["Oracle"](f["sc"].desc)
        x, x0 = snap[f["name"]]
        z = f["z"].cpu().numpy(); ef = f["ef"].cpu().numpy()
        nxs = o.nx + o.ns
        params = f["sc"].params
        if "obst_dyn" in f["sc"].extra:      # the scene is static data: parameters as packed on the host
            pass
        tried = same = 0
        for b in range(0, f["B"], max(1, f["B"] // 12)):
            r = o.solve(x[b], x0[b], params[b])
            if r["exitflag"] in (1, 2) and ef[b] in (1, 2):
                tried += 1
                du = np.abs(r["z"][0, nxs:] - z[b, 0, nxs:]).max()
                same += du <= 1e-3 * max(1.0, np.abs(r["z"][0, nxs:]).max())
        # (the NLP is non-convex: a cold and a warm start may settle in different local solutions for a few instances)
        assert tried >= 8 and same >= 0.75 * tried, (f["name"], tried, same)
    shard.close()


def test_mixed_fleet_real_time_settings(rt):
    """The loop as `bench.py --config cfg5` runs it: iteration limit 20, acceptable-termination window 3, deadline of
    24 passes per solve (40 for the boxers, rmpc_set_pass_budget), one episode of 40 control steps on a scaled-down
    shard.  Bars: usable plans (exit flag 1 or 2) for >= 95 % of every robot type over the episode and >= 90 % in
    every control step, the arms converged (flag 1, not merely acceptable) for >= 90 %, the pass kernels inside
    their budget."""
    import torch
    from robot_mpcs_amd import fleet
    counts = {"cfg2": 512, "cfg3": 384, "cfg4": 128}
    shard = fleet.MixedFleetShard(counts, torch.device("cuda:0"), seed=7, previous_plan=True, warm_duals=True,
                                  options={"max_iter": 20, "acc_iters": 3}, pass_budget={"cfg2": 24, "cfg3": 40, "cfg4": 24})
    usable = {k: [] for k in counts}
    conv = []
    for step in range(40):
        shard.tick()
        for name, (c1, c2, c0, neg, itmean) in shard.stats().items():
            usable[name].append((c1 + c2) / counts[name])
            if name == "cfg4":
                conv.append(c1 / counts[name])
        arm = [f for f in shard.fleets if f["name"] == "cfg4"][0]
        assert arm["s"].last_passes() <= 24
    for name, u in usable.items():
        assert np.mean(u[1:]) >= 0.95, (name, np.mean(u[1:]))
        assert np.min(u[1:]) >= 0.90, (name, np.min(u[1:]))
    assert np.mean(conv[1:]) >= 0.90, np.mean(conv[1:])
    shard.close()


def test_mixed_fleet_steady_loop_full_shard(rt):
    """BASELINE configs[4] at the real per-GPU shard (4096 point robots + 3072 boxers + 1024 arms) as a STEADY loop, the
    way `bench.py --config cfg5` runs it (iteration limit 40, acceptable window 3, deadlines 32 / 56 / 16 passes): 400
    consecutive control steps, nothing is reset by the harness.  The world is the reference's: the boxers' obstacles
    move every control step (rmpc_advance_obstacles_device; the planner gets their current state,
    mpcPlanner.py:243-244), an instance whose solve failed keeps its state and drives on with the action it got
    (mpcPlanner.py:263-264) -- only 25 failed control steps in a row put it back to its start state --, and an instance
    takes its next goal when it has arrived, has come to rest (the reference's N w / h term keeps a robot off a goal next
    to an obstacle: tests/tools/dev_oracle_arrival.py) or has dwelt 150 control steps on it (rmpc_retarget_device).
    Bars: usable plans (exit flag 1 or 2) for >= 95 % of every robot type over the 400 steps; per type at least as many
    robots reach or settle on their goal as time out on it; resets <= 0.02 % of the instances per control step; the
    control step stays near the 10 ms of the 100 Hz loop (the rate itself is bench.py's business: >= 90 % of the steps
    inside 12 ms on whatever box this runs); and the oracle -- cold-started from the device's own state, shifted plan,
    current goal and current obstacles of the last step -- reaches the same first control wherever both converge to the
    same objective."""
    import time
    import torch
    from robot_mpcs_amd import fleet
    counts = {"cfg2": 4096, "cfg3": 3072, "cfg4": 1024}
    dev = torch.device("cuda:0")
    shard = fleet.MixedFleetShard(counts, dev, seed=7, previous_plan=True, warm_duals=True,
                                  options={"max_iter": 40, "acc_iters": 3}, pass_budget={"cfg2": 32, "cfg3": 56, "cfg4": 16},
                                  steady=True, max_dwell=150)
    assert shard.instances == 8192
    for _ in range(10):
        shard.tick()
    shard.steady_stats(reset=True)
    steps = 400
    ms = []
    for step in range(steps):
        if step == steps - 1:   # inputs of the last control step, for the oracle
            torch.cuda.synchronize()
            snap = {f["name"]: (f["x"].cpu().numpy().copy(), f["x0"].cpu().numpy().copy(), f["goal"].cpu().numpy().copy(),
                                None if f["obst_dyn"] is None else f["obst_dyn"].cpu().numpy().copy())
                    for f in shard.fleets}
        t = time.perf_counter()
        shard.tick()
        ms.append(1e3 * (time.perf_counter() - t))
    ss = shard.steady_stats()
    ms = np.array(ms)
    assert (ms <= 12.0).mean() >= 0.90, (np.percentile(ms, [50, 90, 99]), ms.max())
    for name, st in ss.items():
        conv, acc, cut, failed, iters = st["acc"]
        arrived, settled, timeouts, resets = st["events"]
        n = counts[name] * steps
        assert conv + acc + cut + failed == n, (name, st)
        assert (conv + acc) / n >= 0.95, (name, st)
        assert failed / n <= 0.01, (name, st)
        assert arrived + settled > 0 and arrived + settled >= timeouts, (name, st)   # goals are reached or settled on, not timed out
        assert resets / n <= 2e-4, (name, st)
    for f in shard.fleets:
        o = rt["Oracle"](f["sc"].desc)
        x, x0, goal, od = snap[f["name"]]
        if od is not None:
            od = od.reshape(od.shape[0], -1, 9)
        z = f["z"].cpu().numpy(); ef = f["ef"].cpu().numpy(); ob = f["obj"].cpu().numpy()
        nxs = o.nx + o.ns
        og = int(f["sc"].desc["off_goal"])
        oo = int(f["sc"].desc["off_obst"])
        dt = float(f["sc"].desc["dt"])
        tried = same = 0
        for b in range(0, f["B"], max(1, f["B"] // 12)):
            params = f["sc"].params[b].reshape(o.N, -1).copy()
            params[:, og:og + 3] = goal[b]
            if od is not None:
                # updateDynamicObstacles (mpcPlanner.py:144-161) from the obstacles' state at this control step
                kk = np.arange(o.N)[:, None]
                for j in range(od.shape[1]):
                    pos, vel, accl = od[b, j, 0:3], od[b, j, 3:6], od[b, j, 6:9]
                    params[:, oo + 4 * j: oo + 4 * j + 3] = pos + vel * dt * kk + 0.5 * (dt * kk) ** 2 * accl
            r = o.solve(x[b], x0[b], params)
            if r["exitflag"] in (1, 2) and ef[b] in (1, 2) and abs(r["obj"] - ob[b]) <= 1e-6 * max(1.0, abs(ob[b])):
                tried += 1
                du = np.abs(r["z"][0, nxs:] - z[b, 0, nxs:]).max()
                same += du <= 1e-3 * max(1.0, np.abs(r["z"][0, nxs:]).max())
        assert tried >= 6 and same == tried, (f["name"], tried, same)
    shard.close()


@pytest.mark.parametrize("name,B", [("cfg2", 333), ("cfg3", 200)])
def test_launch_order_of_warm_fused_launches_changes_no_result(rt, name, B, monkeypatch):
    """A warm-started fused launch takes the instances in the order of their previous solve's passes, longest first
    (k_order); RMPC_NO_ORDER=1 keeps the index order.  Which slot an instance runs in -- and which instance shares
    its wavefront -- must not change a bit of what it computes."""
    sc = rt["make_scenario"](name, B=B, seed=3)
    monkeypatch.delenv("RMPC_NO_ORDER", raising=False)
    a = rt["Solver"](sc.desc, max_batch=B); a.set_warm_start(True)
    monkeypatch.setenv("RMPC_NO_ORDER", "1")     # (read once, at rmpc_create)
    b = rt["Solver"](sc.desc, max_batch=B); b.set_warm_start(True)
    nx = sc.desc["nx"]
    x = sc.xinit.copy(); x0 = sc.x0.copy()
    for t in range(4):
        ra = a.solve(x, x0, sc.params); rb = b.solve(x, x0, sc.params)
        assert np.array_equal(ra["exitflag"], rb["exitflag"]) and np.array_equal(ra["iters"], rb["iters"]), t
        assert np.array_equal(ra["z"], rb["z"]), t
        assert a.last_passes() == b.last_passes()
        ok = ra["exitflag"] >= 0
        z = ra["z"]
        x0 = np.where(ok[:, None, None], np.concatenate([z[:, 1:], z[:, -1:]], axis=1), x0)
        x = np.where(ok[:, None], z[:, 1, :nx], x)
    a.close(); b.close()


def test_launch_order_of_cold_fused_launches_changes_no_result(rt, monkeypatch):
    """A cold fused launch that is larger than the chip takes the instances closest to a constraint boundary first
    (k_difficulty + k_order_t<64>: an estimate of who takes long; holonomic chains); RMPC_NO_COLD_ORDER=1 keeps the
    index order.  The place in the queue must not change a bit of what an instance computes."""
    B = 4096            # (more than two instances per wavefront of the chip: there is a queue to order)
    sc = rt["make_scenario"]("cfg2", B=B, seed=11)
    monkeypatch.delenv("RMPC_NO_COLD_ORDER", raising=False)
    a = rt["Solver"](sc.desc, max_batch=B)
    monkeypatch.setenv("RMPC_NO_COLD_ORDER", "1")     # (read once, at rmpc_create)
    b = rt["Solver"](sc.desc, max_batch=B)
    ra = a.solve(sc.xinit, sc.x0, sc.params); rb = b.solve(sc.xinit, sc.x0, sc.params)
    a.close(); b.close()
    for key in ("z", "exitflag", "iters", "obj", "kkt"):
        assert np.array_equal(ra[key], rb[key]), key
