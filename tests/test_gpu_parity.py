"""Parity tests proper: the HIP solver, called through the C ABI
(include/rmpc.h via robot_mpcs_amd._lib), against the CPU oracle on the same
seeded inputs, against the committed golden vectors, and -- at BASELINE.json's
full sizes -- through size-independent properties.

Stated floating-point tolerance (fp64 throughout):
    |u_1^GPU - u_1^oracle|_inf <= 1e-6 * max(1, |u_1|_inf)      (applied control)
    |z^GPU - z^oracle|_inf     <= 1e-6 * max(1, |z|_inf)        (whole plan)
for every instance both sides report converged; exitflags must be equal.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-6
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def rt():
    import __graft_entry__ as g
    g.build()
    from oracle.oracle import Oracle
    from robot_mpcs_amd._lib import Solver
    from robot_mpcs_amd.scenarios import make_scenario
    return dict(Oracle=Oracle, Solver=Solver, make_scenario=make_scenario)


def _check_plan(gpu, cpu, nxs, tol_stat=1e-6):
    # equal exit flags; a converged <-> acceptable flip is tolerated only when the KKT residual sits within
    # 2x the tolerance (rounding decides which stop fires first there; fleet.flags_consistent)
    from robot_mpcs_amd.fleet import flags_consistent
    assert flags_consistent(gpu["exitflag"], cpu["exitflag"], gpu["kkt"], tol_stat), \
        np.flatnonzero(gpu["exitflag"] != cpu["exitflag"])
    conv = np.isin(cpu["exitflag"], (1, 2))  # converged or acceptable level
    zs = np.maximum(1.0, np.abs(cpu["z"]).reshape(len(conv), -1).max(axis=1))
    dz = np.abs(gpu["z"] - cpu["z"]).reshape(len(conv), -1).max(axis=1)
    du = np.abs(gpu["z"][:, 0, nxs:] - cpu["z"][:, 0, nxs:]).max(axis=1)
    us = np.maximum(1.0, np.abs(cpu["z"][:, 0, nxs:]).max(axis=1))
    assert np.all(du[conv] <= TOL * us[conv]), du[conv].max()
    assert np.all(dz[conv] <= TOL * zs[conv]), dz[conv].max()
    # same algorithm, same arithmetic: iteration counts agree (allow rare borderline flips)
    assert (gpu["iters"] == cpu["iters"]).mean() >= 0.98


@pytest.mark.parametrize("name,B,seed", [
    ("cfg1", 1, 0), ("cfg2", 192, 1), ("cfg3", 192, 2), ("cfg4", 96, 3), ("boxer", 65, 4), ("pointRobot", 7, 5), ("panda", 3, 6),
    # VelLimitConstraints rows and a non-zero ConstraintAvoidance weight on every module (single-variable rows
    # as "first row of a module", Linear rows, self-collision rows)
    ("wc_point", 96, 31), ("wc_boxer", 96, 32), ("wc_boxer_slack", 96, 33), ("wc_panda", 40, 34),
    # kernel variants beyond the three shipped robots (mpcBase.py:52-55: n = fk.n() of any URDF chain): a 2-joint
    # gantry (fused kernel) and 4 / 5 / 6 / 8-joint arms (pass kernels; n = 5, 6 with the cost-to-go update on the matrix cores)
    ("chain2", 96, 41), ("chain4", 48, 42), ("chain5", 48, 43), ("chain6", 48, 44),
    # n = 8 = RMPC_MAX_JOINTS: the panda's chain with one more revolute joint (test asset panda_tool8.urdf)
    ("chain8", 32, 45),
])
def test_solve_matches_oracle(rt, name, B, seed):
    sc = rt["make_scenario"](name, B=B, seed=seed)
    cpu = rt["Oracle"](sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)
    s = rt["Solver"](sc.desc, max_batch=B)
    gpu = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    _check_plan(gpu, cpu, sc.desc["nx"] + sc.desc["ns"])
    np.testing.assert_allclose(gpu["obj"], cpu["obj"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name,B,seed", [("cfg4", 96, 71), ("wc_panda", 40, 72), ("chain5", 48, 73), ("chain6", 48, 74)])
def test_arm_pass_kernels_match_oracle(rt, name, B, seed, monkeypatch):
    """The arms with 5 .. 7 joints run k_fused_arm by default (round 4); their pass kernels -- what horizons beyond 32
    stages use, with the Schur-complement recursion of round 4 -- are kept under test through RMPC_NO_FUSED=1."""
    monkeypatch.setenv("RMPC_NO_FUSED", "1")
    sc = rt["make_scenario"](name, B=B, seed=seed)
    cpu = rt["Oracle"](sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)
    s = rt["Solver"](sc.desc, max_batch=B)
    assert not s.is_fused()
    gpu = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    _check_plan(gpu, cpu, sc.desc["nx"] + sc.desc["ns"])


@pytest.mark.parametrize("name,B,seed,kw", [("cfg4", 96, 75, {}), ("chain5", 48, 76, {}), ("chain6", 48, 77, {}), ("cfg4", 32, 79, {"time_horizon": 21}), ("cfg4", 32, 80, {"time_horizon": 22})])
def test_fused_arm_two_parts_per_stage_match_oracle(rt, name, B, seed, kw, monkeypatch):
    """k_fused_arm deals a stage's slots and joints to THREE lanes when the horizon leaves room (3 N <= 64: 63 of the 64
    lanes carry rows at N = 21, 60 at BASELINE's N = 20) and to two otherwise.  The two-part kernel forced at short
    horizons (RMPC_ARM_TWO_PARTS=1, read at rmpc_create) against the oracle, and both dealings against each other: the
    same flags and iteration counts, plans to the parity tolerance (the parts' partial sums are added in another order)."""
    sc = rt["make_scenario"](name, B=B, seed=seed, **kw)
    cpu = rt["Oracle"](sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)
    s3 = rt["Solver"](sc.desc, max_batch=B)
    assert s3.is_fused()
    three = s3.solve(sc.xinit, sc.x0, sc.params)
    s3.close()
    monkeypatch.setenv("RMPC_ARM_TWO_PARTS", "1")
    s2 = rt["Solver"](sc.desc, max_batch=B)
    assert s2.is_fused()
    two = s2.solve(sc.xinit, sc.x0, sc.params)
    s2.close()
    _check_plan(two, cpu, sc.desc["nx"] + sc.desc["ns"])
    _check_plan(three, cpu, sc.desc["nx"] + sc.desc["ns"])
    assert np.array_equal(two["exitflag"], three["exitflag"]) and np.array_equal(two["iters"], three["iters"])


@pytest.mark.parametrize("name,B,seed", [("cfg2", 192, 81), ("chain2", 96, 82)])
def test_lane_per_instance_recursion_matches_oracle(rt, name, B, seed, monkeypatch):
    """k_riccati_lane (one lane per instance: the default recursion of the pass kernels for lists of at least 16384
    instances of a small chain) forced for a small batch (RMPC_RIC_LANE=2, RMPC_NO_FUSED=1): flags, iteration counts and
    plans against the oracle like every other path."""
    monkeypatch.setenv("RMPC_NO_FUSED", "1")
    monkeypatch.setenv("RMPC_RIC_LANE", "2")
    sc = rt["make_scenario"](name, B=B, seed=seed)
    cpu = rt["Oracle"](sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)
    s = rt["Solver"](sc.desc, max_batch=B)
    assert not s.is_fused()
    gpu = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    _check_plan(gpu, cpu, sc.desc["nx"] + sc.desc["ns"])


# (cfg3: seed 66, not 64 -- instance 14 of seed 64 is one of the rare boxers whose iteration is sensitive to rounding: library and
#  oracle agree to 4e-13 after 10 iterations, 3e-8 after 20, 6e-5 after 30, and part ways in a crawl that one of them leaves after
#  109 iterations and the other not within 200; the full-size sweep, profiles/r04_parity_sweep.txt, counts such instances)
@pytest.mark.parametrize("name,B,seed", [("cfg2", 300, 61), ("wc_point", 200, 62), ("chain2", 200, 63), ("cfg3", 200, 66)])
def test_queue_and_cold_order_on_a_tiny_grid(rt, name, B, seed, monkeypatch):
    """RMPC_FUSED_GRID=4: eight half-wavefronts drain a queue of hundreds of instances, and a cold launch of the
    chains orders that queue by k_difficulty -- the refill, the per-instance pass counters, the mixed first / later
    sweeps of a wavefront's two halves and the launch order, exercised at test sizes, against the oracle."""
    monkeypatch.setenv("RMPC_FUSED_GRID", "4")       # (read once, at rmpc_create)
    sc = rt["make_scenario"](name, B=B, seed=seed)
    cpu = rt["Oracle"](sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)
    s = rt["Solver"](sc.desc, max_batch=B)
    gpu = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    _check_plan(gpu, cpu, sc.desc["nx"] + sc.desc["ns"])


@pytest.mark.parametrize("N", [12, 17, 30])
def test_arm_horizons_around_the_lds_image_slots(rt, N):
    """The arm's recursion keeps the gain images of its first 16 stages in LDS between the backward and the forward
    pass and takes the rest from the gain record, four stages ahead (riccati_recursion, RicLds::IMG_SLOTS): a horizon
    that fits the slots entirely, one that needs a single stage from the record, and one with more record stages than
    prefetch buffers -- each against the oracle."""
    sc = rt["make_scenario"]("cfg4", B=24, seed=50 + N, time_horizon=N)
    assert sc.desc["N"] == N
    cpu = rt["Oracle"](sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)
    s = rt["Solver"](sc.desc, max_batch=24)
    gpu = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    _check_plan(gpu, cpu, sc.desc["nx"] + sc.desc["ns"])
    np.testing.assert_allclose(gpu["obj"], cpu["obj"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("B,seed,kw", [(160, 91, {}), (48, 92, {"time_horizon": 7}), (48, 93, {"time_horizon": 32})])
def test_fused_point_robot_with_slack_matches_oracle(rt, B, seed, kw):
    """The point robot with the slack variable (`slack: true`: Cfg<CHAIN, 3, 1>) is the one model that still runs the
    gain-form in-LDS recursion of k_fused (`FAST` in riccati_recursion) since the chains without slack moved to the block
    form: kept under test on its own -- flags, iteration counts and plans against the oracle."""
    sc = rt["make_scenario"]("cfg2", B=B, seed=seed, slack=True, **kw)
    assert sc.desc["ns"] == 1
    cpu = rt["Oracle"](sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)
    s = rt["Solver"](sc.desc, max_batch=B)
    assert s.is_fused()
    gpu = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    _check_plan(gpu, cpu, sc.desc["nx"] + sc.desc["ns"])


@pytest.mark.parametrize("name", ["cfg2", "chain2"])
@pytest.mark.parametrize("N", [1, 2, 5, 31, 32])
def test_fused_point_robot_horizons_match_oracle(rt, name, N):
    """The block-form recursion of k_fused (chains with n <= 3 joints and no slack: the cost-to-go stays in the registers
    of the lanes of an n x n grid from stage to stage, two ordering points per backward stage) at the edges of its stage
    loop: a single stage (no recursion to speak of), two stages, and the last horizons the 32 slots of an instance hold."""
    sc = rt["make_scenario"](name, B=48, seed=100 + N, time_horizon=N)
    cpu = rt["Oracle"](sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)
    s = rt["Solver"](sc.desc, max_batch=48)
    assert s.is_fused()
    gpu = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    _check_plan(gpu, cpu, sc.desc["nx"] + sc.desc["ns"])
    np.testing.assert_allclose(gpu["obj"], cpu["obj"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name,B,kw", [
    ("cfg1", 3, {}), ("cfg2", 2048, {}), ("cfg3", 512, {}),
    # the pass kernels: the arm (its sweep keeps the q blocks, the next stage's state and the costates in LDS columns,
    # its recursion the stage matrices; 420 B of scratch per lane), a weighted arm, and horizons beyond the fused
    # kernel's 32 stages (point robot and boxer through k_sweep / k_riccati / k_step, with survivor migration)
    ("cfg4", 256, {}), ("wc_panda", 64, {}), ("cfg2", 1536, {"time_horizon": 40}), ("cfg3", 192, {"time_horizon": 36}),
    ("chain2", 64, {}), ("chain5", 64, {}), ("chain8", 24, {}),
])
def test_solve_does_not_depend_on_stale_lds(rt, name, B, kw):
    """Every word a kernel reads from LDS, scratch or the handle's workspace must have been written by the same solve:
    the stage records of the fused kernel (round 2: the defect entries of the last stage were not -- 0 * NaN reached
    the gains on a box whose previous kernel had left NaN patterns there), the LDS columns and spill slots of the
    arm's sweep, the iterate buffers a first pass reads and discards.  Before the second solve the LDS of every CU,
    the scratch memory and the whole workspace are filled with NaN patterns (``rmpc_debug_poison_lds``)."""
    sc = rt["make_scenario"](name, B=B, seed=21, **kw)
    s = rt["Solver"](sc.desc, max_batch=B)
    clean = s.solve(sc.xinit, sc.x0, sc.params)
    s.poison_lds()
    dirty = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    assert np.array_equal(clean["exitflag"], dirty["exitflag"]) and np.array_equal(clean["iters"], dirty["iters"])
    assert np.array_equal(clean["z"], dirty["z"])
    assert np.isin(clean["exitflag"], (1, 2)).mean() > 0.95


@pytest.mark.parametrize("name,B,seed", [("cfg2", 2048, 7), ("cfg3", 1280, 8)])
def test_survivor_migration_is_transparent(rt, name, B, seed):
    """Batches of >= 1024 instances move their last survivors to the compact workspace
    (k_migrate).  Same arithmetic in other columns: results must be bit-identical to a solve with
    migration switched off, and match the oracle like any other batch."""
    sc = rt["make_scenario"](name, B=B, seed=seed)
    s = rt["Solver"](sc.desc, max_batch=B)
    gpu = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    os.environ["RMPC_NO_MIGRATE"] = "1"   # debugging switch, read once when a handle is created
    try:
        s2 = rt["Solver"](sc.desc, max_batch=B)
    finally:
        del os.environ["RMPC_NO_MIGRATE"]
    ref = s2.solve(sc.xinit, sc.x0, sc.params)
    s2.close()
    for key in ("z", "exitflag", "iters", "obj", "kkt"):
        assert np.array_equal(gpu[key], ref[key]), key
    assert gpu["iters"].max() > 16   # some instances did outlive the migration point
    cpu = rt["Oracle"](sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)
    _check_plan(gpu, cpu, sc.desc["nx"] + sc.desc["ns"])


def test_handle_reuse_after_migration(rt):
    """A handle that has just run a migrating batch solves a small batch exactly like a fresh handle."""
    big = rt["make_scenario"]("cfg2", B=2048, seed=21)
    small = rt["make_scenario"]("cfg2", B=100, seed=22)
    s = rt["Solver"](big.desc, max_batch=2048)
    s.solve(big.xinit, big.x0, big.params)
    a = s.solve(small.xinit, small.x0, small.params)
    s.close()
    f = rt["Solver"](small.desc, max_batch=100)
    b = f.solve(small.xinit, small.x0, small.params)
    f.close()
    for key in ("z", "exitflag", "iters", "obj", "kkt"):
        assert np.array_equal(a[key], b[key]), key


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "boxer", "pointRobot", "panda"])
def test_solve_matches_golden_vectors(rt, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    desc = json.loads(str(g["desc"]))
    desc["lb"] = [float(v) for v in desc["lb"]]; desc["ub"] = [float(v) for v in desc["ub"]]
    B = g["xinit"].shape[0]
    s = rt["Solver"](desc, max_batch=B)
    gpu = s.solve(g["xinit"], g["x0"], g["params"])
    s.close()
    assert np.array_equal(gpu["exitflag"], g["exitflag"])
    scale = np.maximum(1.0, np.abs(g["z"]).max())
    assert np.abs(gpu["z"] - g["z"]).max() <= TOL * scale
    assert np.array_equal(gpu["iters"], g["iters"])


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4", "boxer", "wc_point", "wc_boxer", "wc_boxer_slack", "wc_panda"])
def test_sweep_blocks_match_oracle(rt, name):
    """Kernel-level parity: condensed stage Hessian / gradient blocks, dynamics
    defect, rows and cost of the first stage-parallel sweep."""
    B = 5
    sc = rt["make_scenario"](name, B=B, seed=9)
    o = rt["Oracle"](sc.desc)
    s = rt["Solver"](sc.desc, max_batch=B)
    dbg = s.debug_sweep(sc.xinit, sc.x0, sc.params)
    s.close()
    N, nv, nx, nh = o.N, o.nv, o.nx, o.nh
    mu0 = sc.desc["options"]["mu0"]
    for b in range(B):
        P = sc.params[b].reshape(N, o.npar)
        for k in range(N):
            e = o.eval_stage(sc.x0[b, k], P[k], fixed_state=(k == 0))
            t = np.maximum(e["g"], 1e-2); lam = mu0 / t
            rg = e["g"] - t
            Q = e["H"] + e["Jg"].T @ np.diag(lam / t) @ e["Jg"]
            q0 = e["gf"] + e["Jg"].T @ (lam * rg / t)
            q1 = e["Jg"].T @ (1.0 / t)
            np.testing.assert_allclose(dbg["f"][b, k], e["f"], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(dbg["g"][b, k], e["g"][:nh], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(dbg["Q"][b, k], Q, rtol=1e-11, atol=1e-11 * max(1.0, np.abs(Q).max()))
            np.testing.assert_allclose(dbg["q0"][b, k], q0, rtol=1e-11, atol=1e-11 * max(1.0, np.abs(q0).max()))
            np.testing.assert_allclose(dbg["q1"][b, k], q1, rtol=1e-11, atol=1e-11 * max(1.0, np.abs(q1).max()))
            if k < N - 1:
                np.testing.assert_allclose(dbg["rc"][b, k], e["xnext"] - sc.x0[b, k + 1, :nx], rtol=0, atol=1e-13)


def test_full_size_cfg2_properties(rt):
    """BASELINE configs[1] at full size (B = 4096): KKT bar, feasibility recomputed
    independently in numpy, permutation equivariance and replica determinism."""
    sc = rt["make_scenario"]("cfg2", seed=0)
    B, N = sc.B, sc.desc["N"]
    s = rt["Solver"](sc.desc, max_batch=B)
    r = s.solve(sc.xinit, sc.x0, sc.params)
    assert np.all(r["exitflag"] >= 0) and (r["exitflag"] == 1).mean() > 0.995
    conv = r["exitflag"] == 1
    assert r["kkt"][conv].max() <= 1e-6
    assert r["iters"].max() <= 80  # exact constraint curvature: no long Gauss-Newton iteration tail
    z = r["z"]
    dt = sc.desc["dt"]
    q, v, u = z[:, :, 0:3], z[:, :, 3:6], z[:, :, 6:9]
    assert np.array_equal(z[:, 0, :6], sc.xinit)
    # double-integrator shooting defects
    dq = q[:, 1:] - (q[:, :-1] + dt * v[:, :-1] + 0.5 * dt * dt * u[:, :-1])
    dv = v[:, 1:] - (v[:, :-1] + dt * u[:, :-1])
    assert np.abs(dq[conv]).max() <= 1e-7 and np.abs(dv[conv]).max() <= 1e-7
    # obstacle clearance, input and joint limits
    pos = np.concatenate([q[:, :, :2], np.full((B, N, 1), 0.05)], axis=2)
    op, orad = sc.extra["obst_pos"], sc.extra["obst_radius"]
    dist = np.linalg.norm(pos[:, :, None, :] - op[:, None, :, :], axis=3) - orad[:, None, :] - sc.extra["r_body"]
    assert dist[conv].min() >= -1e-7
    assert np.abs(u[conv][:, :, :2]).max() <= 1.0 + 1e-7 and np.abs(q[conv]).max() <= 10.0 + 1e-7
    # permutation of the batch permutes the outputs bit for bit
    perm = np.random.default_rng(0).permutation(B)
    rp = s.solve(sc.xinit[perm], sc.x0[perm], sc.params[perm])
    assert np.array_equal(rp["z"], z[perm]) and np.array_equal(rp["iters"], r["iters"][perm])
    # B copies of one instance give bit-identical plans
    one = lambda a: np.repeat(a[5:6], 300, axis=0)
    rr = s.solve(one(sc.xinit), one(sc.x0), one(sc.params))
    assert np.all(rr["z"] == rr["z"][0:1]) and np.array_equal(rr["z"][0], z[5])
    s.close()


@pytest.mark.parametrize("name", ["cfg3", "cfg4"])
def test_full_size_feasibility_against_oracle_rows(rt, name):
    """Full-size boxer / panda batches: every converged plan satisfies the
    oracle's inequality rows and dynamics at the returned point (sampled)."""
    sc = rt["make_scenario"](name, seed=0)
    s = rt["Solver"](sc.desc, max_batch=sc.B)
    r = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    assert np.all(r["exitflag"] >= 0) and np.isin(r["exitflag"], (1, 2)).mean() > 0.99 and (r["exitflag"] == 1).mean() > 0.85
    conv = np.flatnonzero(np.isin(r["exitflag"], (1, 2)))
    assert r["kkt"][r["exitflag"] == 1].max() <= 1e-6
    o = rt["Oracle"](sc.desc)
    rng = np.random.default_rng(1)
    for b in rng.choice(conv, size=24, replace=False):
        P = sc.params[b].reshape(o.N, o.npar)
        for k in range(o.N):
            e = o.eval_stage(r["z"][b, k], P[k], derivs=False, fixed_state=(k == 0))
            assert e["g"].min() >= -2e-6
            if k < o.N - 1:
                assert np.abs(e["xnext"] - r["z"][b, k + 1, : o.nx]).max() <= 2e-6


@pytest.mark.parametrize("name,B,chunk", [("cfg2", 32768, 4096), ("cfg3", 16384, 4096), ("cfg4", 8192, 1024)])
def test_large_batch_equals_its_chunks(rt, name, B, chunk):
    """Eight times BASELINE's batch in ONE call (the queue behind the grid is 16 to 32 instances deep per half-wavefront)
    against the same instances solved chunk by chunk through a handle of BASELINE's size: plans, flags, iteration counts
    and objectives bit for bit -- what an instance returns depends neither on the batch it travels in nor on its place
    in the queue."""
    sc = rt["make_scenario"](name, B=B, seed=77)
    big = rt["Solver"](sc.desc, max_batch=B)
    r = big.solve(sc.xinit, sc.x0, sc.params)
    big.close()
    assert np.all(r["exitflag"] >= 0) and np.isin(r["exitflag"], (1, 2)).mean() > 0.99
    small = rt["Solver"](sc.desc, max_batch=chunk)
    for c in range(0, B, chunk):
        q = small.solve(sc.xinit[c:c + chunk], sc.x0[c:c + chunk], sc.params[c:c + chunk])
        for k in ("z", "exitflag", "iters", "obj"):
            assert np.array_equal(q[k], r[k][c:c + chunk]), (k, c)
    small.close()


def test_edge_cases(rt):
    sc = rt["make_scenario"]("cfg1", B=1)
    s = rt["Solver"](sc.desc, max_batch=4)
    # infeasible start (inside the obstacle) -> negative exitflag, plan = start point
    xinit = sc.xinit.copy(); xinit[0, 0:2] = [4.0, -0.5]
    x0 = sc.x0.copy(); x0[0, :, :6] = xinit[0]
    r = s.solve(xinit, x0, sc.params)
    cpu = rt["Oracle"](sc.desc).solve_batch(xinit, x0, sc.params)
    assert r["exitflag"][0] < 0 and r["exitflag"][0] == cpu["exitflag"][0]
    # batch larger than the handle was created for -> API error, not a crash
    from robot_mpcs_amd._lib import RmpcError
    big = lambda a: np.repeat(a, 5, axis=0)
    with pytest.raises(RmpcError):
        s.solve(big(sc.xinit), big(sc.x0), big(sc.params))
    # x0 whose first-stage state differs from xinit: stage 1 is overwritten by xinit
    x0b = sc.x0.copy(); x0b[0, 0, :6] += 0.3
    r2 = s.solve(sc.xinit, x0b, sc.params)
    r1 = s.solve(sc.xinit, sc.x0, sc.params)
    assert np.array_equal(r1["z"], r2["z"])
    s.close()


@pytest.mark.parametrize("name,B,kw", [("cfg2", 96, {}), ("cfg3", 96, {}), ("cfg4", 48, {}), ("wc_boxer_slack", 64, {}),
                                       ("cfg2", 96, {"RMPC_NO_FUSED": "1"}), ("cfg4", 48, {"RMPC_NO_FUSED": "1"})])
def test_non_finite_inputs_are_contained(rt, name, B, kw, monkeypatch):
    """NaN / Inf in the inputs of some instances (start state, one parameter, the initial guess, every parameter): those
    instances stop at iteration 0 with a negative exit flag on both sides (the reference prints the flag and goes on,
    mpcPlanner.py:263-264), the launch ends, and EVERY OTHER instance -- also the one that shares a wavefront with a
    poisoned one -- returns bit for bit what it returns from clean inputs (fused kernels and pass kernels)."""
    for k, v in kw.items():
        monkeypatch.setenv(k, v)
    sc = rt["make_scenario"](name, B=B, seed=21)
    d = sc.desc
    N, nx, npar = d["N"], d["nx"], d["npar"]
    s = rt["Solver"](d, max_batch=B)
    clean = s.solve(sc.xinit, sc.x0, sc.params)
    xi, x0, pr = sc.xinit.copy(), sc.x0.copy(), sc.params.copy()
    xi[3, 0] = np.nan
    pr.reshape(B, N, npar)[7, 2, 1] = np.inf
    x0.reshape(B, N, -1)[11, 5, nx:] = np.nan
    pr.reshape(B, N, npar)[12, :, :] = np.nan
    bad = np.array([3, 7, 11, 12])
    gpu = s.solve(xi, x0, pr)
    again = s.solve(sc.xinit, sc.x0, sc.params)      # the handle carries nothing over from the poisoned launch
    s.close()
    cpu = rt["Oracle"](d).solve_batch(xi, x0, pr)
    good = np.setdiff1d(np.arange(B), bad)
    assert np.all(gpu["exitflag"][bad] < 0) and np.all(cpu["exitflag"][bad] < 0)
    assert np.all(gpu["iters"][bad] == 0) and np.all(cpu["iters"][bad] == 0)
    for k in ("z", "exitflag", "iters", "obj"):
        assert np.array_equal(gpu[k][good], clean[k][good]), k
        assert np.array_equal(again[k], clean[k]), k
    assert np.array_equal(cpu["exitflag"][good], gpu["exitflag"][good])


def test_line_search_cap_matches_oracle(rt):
    """ls_max (rmpc_desc.ls_max): instances that exhaust their halvings stop with -8 on both sides."""
    sc = rt["make_scenario"]("cfg3", B=256, seed=12)
    d = dict(sc.desc)
    d["options"] = dict(d["options"], ls_max=1, max_iter=12)
    cpu = rt["Oracle"](d).solve_batch(sc.xinit, sc.x0, sc.params)
    s = rt["Solver"](d, max_batch=256)
    gpu = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    from robot_mpcs_amd.fleet import flags_consistent
    assert flags_consistent(gpu["exitflag"], cpu["exitflag"], gpu["kkt"], 1e-6)
    assert (cpu["exitflag"] == -8).any() or (cpu["exitflag"] == 0).any()   # the caps do bite
    ok = np.abs(gpu["z"] - cpu["z"]).reshape(256, -1).max(axis=1) <= TOL * np.maximum(1.0, np.abs(cpu["z"]).reshape(256, -1).max(axis=1))
    assert ok.mean() >= 0.98


def test_iteration_cap_returns_usable_plan(rt):
    sc = rt["make_scenario"]("cfg2", B=32, seed=3)
    d = dict(sc.desc); d["options"] = dict(d["options"], max_iter=4)
    s = rt["Solver"](d, max_batch=32)
    r = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    cpu = rt["Oracle"](d).solve_batch(sc.xinit, sc.x0, sc.params)
    assert np.all(r["exitflag"] == 0) and np.all(r["iters"] == 4)
    assert np.array_equal(cpu["exitflag"], r["exitflag"])
    assert np.abs(r["z"] - cpu["z"]).max() <= 1e-9


def test_device_pointer_entry_matches_host_entry(rt):
    import torch
    sc = rt["make_scenario"]("cfg2", B=128, seed=8)
    s = rt["Solver"](sc.desc, max_batch=128)
    host = s.solve(sc.xinit, sc.x0, sc.params)
    dev = torch.device("cuda:0")
    tx = torch.from_numpy(sc.xinit).to(dev); t0 = torch.from_numpy(sc.x0).to(dev); tp = torch.from_numpy(sc.params).to(dev)
    z = torch.empty((128, sc.desc["N"], s.nvar), dtype=torch.float64, device=dev)
    ef = torch.empty(128, dtype=torch.int32, device=dev); it = torch.empty(128, dtype=torch.int32, device=dev)
    kkt = torch.empty(128, dtype=torch.float64, device=dev); obj = torch.empty(128, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    s.solve_device(128, tx, t0, tp, z, ef, it, kkt, obj, stream=st.cuda_stream)
    st.synchronize()
    assert np.array_equal(z.cpu().numpy(), host["z"]) and np.array_equal(ef.cpu().numpy(), host["exitflag"])
    s.close()


def test_planner_drop_in_closed_loop(rt, tmp_path):
    """makeSolver -> MPCPlanner -> computeAction in a closed loop with the ERK2
    plant; every step's plan equals the oracle's on the same inputs."""
    from robot_mpcs_amd.planner.mpcPlanner import MPCPlanner
    from robot_mpcs_amd.scenarios import CONFIG_DIR, build_model

    class Obst:
        def __init__(self, p, r): self._p, self._r = p, r
        def position(self): return self._p
        def radius(self): return self._r
        def dimension(self): return 3

    model, setup = build_model(os.path.join(CONFIG_DIR, "cfg1_pointRobotMpc.yaml"))
    model.generateSolver(location=str(tmp_path) + "/")
    planner = MPCPlanner("pointRobot", str(tmp_path) + "/", None, False, **setup["mpc"])
    planner.concretize(); planner.reset()
    planner.setGoalReaching([8.2, -0.2]); planner.setConstraintAvoidance()
    planner.setJointLimits(np.array([[-10, -10, -10], [10, 10, 10.0]]))
    planner.setInputLimits(np.array([[-1, -1, -15], [1, 1, 15.0]]))
    planner.setRadialConstraints([Obst([4.0, -0.5, 0.0], 1.0)], 0.3)
    planner.setSelfCollisionAvoidanceConstraints(0.3)
    o = rt["Oracle"](planner._descriptor)
    q, qdot = np.zeros(3), np.array([0.1, 0.0, 0.0])
    for step in range(25):
        action, output, exitflag = planner.computeAction(q, qdot)
        assert exitflag == 1 and set(output) == {"x%02d" % k for k in range(1, 11)}
        x0 = np.zeros((10, 9)); x0[:, :6] = np.concatenate([q, qdot])
        ref = o.solve(np.concatenate([q, qdot]), x0, planner._params)
        assert np.abs(action - ref["z"][0, 6:]).max() <= 1e-9
        xn = o.dynamics(np.concatenate([q, qdot]), action)
        q, qdot = xn[:3], xn[3:]
    assert q[0] > 0.25 and abs(action[0] - 1.0) < 1e-5  # accelerating towards the goal at the input limit


def test_closed_loop_matches_golden_trace(rt):
    """50 control steps of the reference example scenario with the HIP solver in the loop (B = 1, plant = the
    oracle's ERK2 map, current-state initialisation) against the committed oracle trace."""
    import json
    g = np.load(os.path.join(GOLDEN, "cfg1_closed_loop.npz"))
    desc = json.loads(str(g["desc"]))
    desc["lb"] = [float(v) for v in desc["lb"]]; desc["ub"] = [float(v) for v in desc["ub"]]
    o = rt["Oracle"](desc)
    s = rt["Solver"](desc, max_batch=1)
    nx, nv, N = o.nx, o.nv, o.N
    x = g["xs"][0].copy()
    for t in range(g["us"].shape[0]):
        x0 = np.zeros((1, N, nv))
        x0[0, :, :nx] = x
        r = s.solve(x[None], x0, g["params"][:1])
        assert r["exitflag"][0] == g["exitflag"][t]
        u = r["z"][0, 0, nv - o.nu:]
        assert np.abs(u - g["us"][t]).max() <= TOL * max(1.0, np.abs(g["us"][t]).max()), t
        x = o.dynamics(x, u)
        assert np.abs(x - g["xs"][t + 1]).max() <= TOL, t
    s.close()


# ---------------------------------------------------------------------------------------------------
# planner surface the shipped configs never exercise (reference mpcPlanner.py:265-281, 293-301)
# ---------------------------------------------------------------------------------------------------
class _Obst:
    def __init__(self, p, r): self._p, self._r = p, r
    def position(self): return self._p
    def radius(self): return self._r
    def dimension(self): return 3


def _point_planner(tmp_path, cls=None, batch=None, **over):
    from robot_mpcs_amd.planner.mpcPlanner import BatchedMPCPlanner, MPCPlanner
    from robot_mpcs_amd.scenarios import CONFIG_DIR, build_model
    model, setup = build_model(os.path.join(CONFIG_DIR, "cfg1_pointRobotMpc.yaml"), **over)
    model.generateSolver(location=str(tmp_path) + "/")
    if batch is None:
        planner = MPCPlanner("pointRobot", str(tmp_path) + "/", None, False, **setup["mpc"])
    else:
        planner = BatchedMPCPlanner("pointRobot", str(tmp_path) + "/", batch, None, False, **setup["mpc"])
    planner.concretize(); planner.reset()
    return planner


def _set_point_scene(planner, goal=(8.2, -0.2)):
    planner.setGoalReaching(list(goal)); planner.setConstraintAvoidance()
    planner.setJointLimits(np.array([[-10, -10, -10], [10, 10, 10.0]]))
    planner.setInputLimits(np.array([[-1, -1, -15], [1, 1, 15.0]]))
    planner.setRadialConstraints([_Obst([4.0, -0.5, 0.0], 1.0)], 0.3)
    planner.setSelfCollisionAvoidanceConstraints(0.3)


def test_planner_vel_mode_interval_and_previous_plan(rt, tmp_path):
    """control_mode "vel": action = velocity entries of the SECOND stage (mpcPlanner.py:275-276);
    interval 3: one solve every three calls, the action, output and exitflag of the last solve are
    returned in between (B7 fix, :293-301); initialization previous_plan: shifted warm start (:215-236)."""
    planner = _point_planner(tmp_path, control_mode="vel", interval=3, initialization="previous_plan")
    _set_point_scene(planner)
    o = rt["Oracle"](planner._descriptor)
    q, qdot = np.zeros(3), np.array([0.1, 0.0, 0.0])
    x0 = np.zeros((10, 9)); x0[:, :6] = np.concatenate([q, qdot])   # first call: current state repeated
    last = None
    for step in range(9):
        x = np.concatenate([q, qdot])
        action, output, exitflag = planner.computeAction(q, qdot)
        if step % 3 == 0:
            ref = o.solve(x, x0, planner._params)
            assert exitflag == ref["exitflag"] == 1
            assert np.abs(action - ref["z"][1, 3:6]).max() <= 1e-9       # qdot of stage 2
            assert np.abs(np.stack([output["x%02d" % k] for k in range(1, 11)]) - ref["z"]).max() <= 1e-9
            # the next solve starts from the shifted plan: x0[k] = z[k+1], last stage repeated
            x0 = np.concatenate([ref["z"][1:], ref["z"][-1:]])
            last = (action.copy(), {k: v.copy() for k, v in output.items()}, exitflag)
        else:
            assert np.array_equal(action, last[0]) and exitflag == last[2]
            assert all(np.array_equal(output[k], last[1][k]) for k in output)
        xn = o.dynamics(x, output["x01"][6:])
        q, qdot = xn[:3], xn[3:]


def test_planner_long_horizon_key_format(rt, tmp_path):
    """N = 100: three-digit stage keys on a real solve (mpcPlanner.py:265-273)."""
    planner = _point_planner(tmp_path, time_horizon=100)
    _set_point_scene(planner)
    action, output, exitflag = planner.computeAction(np.zeros(3), np.array([0.1, 0.0, 0.0]))
    assert exitflag in (1, 2) and set(output) == {"x%03d" % k for k in range(1, 101)}
    assert np.array_equal(action, output["x001"][-3:])
    o = rt["Oracle"](planner._descriptor)
    x0 = np.zeros((100, 9)); x0[:, :6] = [0, 0, 0, 0.1, 0, 0]
    ref = o.solve(x0[0, :6], x0, planner._params)
    assert ref["exitflag"] == exitflag and np.abs(action - ref["z"][0, 6:]).max() <= 1e-6


def test_batched_planner_equals_independent_planners(rt, tmp_path):
    """BatchedMPCPlanner(B) == B MPCPlanner objects stepped side by side (bit for bit: one instance's
    arithmetic does not depend on its batch), three control steps, previous_plan warm start."""
    B = 5
    goals = np.array([[8.2, -0.2], [6.0, 3.0], [-5.0, 4.0], [3.0, -6.0], [7.0, 1.5]])
    pb = _point_planner(tmp_path / "b", batch=B, initialization="previous_plan")
    pk = pb.packer
    pk.setGoalReaching(np.concatenate([goals, np.zeros((B, 1))], axis=1)); pk.setConstraintAvoidance()
    pk.setJointLimits(np.array([[-10, -10, -10], [10, 10, 10.0]])); pk.setInputLimits(np.array([[-1, -1, -15], [1, 1, 15.0]]))
    pk.setRadialConstraints(np.array([[[4.0, -0.5, 0.0]]]), np.array([[1.0]]), 0.3)
    singles = []
    for b in range(B):
        p1 = _point_planner(tmp_path / ("s%d" % b), initialization="previous_plan")
        _set_point_scene(p1, goal=goals[b])
        singles.append(p1)
    o = rt["Oracle"](pb._descriptor)
    q = np.zeros((B, 3)); qdot = np.tile([0.1, 0.0, 0.0], (B, 1))
    for step in range(3):
        ab, zb, eb = pb.computeAction(q, qdot)
        for b in range(B):
            a1, out1, e1 = singles[b].computeAction(q[b], qdot[b])
            assert e1 == eb[b] and np.array_equal(a1, ab[b])
            assert np.array_equal(np.stack([out1["x%02d" % k] for k in range(1, 11)]), zb[b])
        xn = np.stack([o.dynamics(np.concatenate([q[b], qdot[b]]), ab[b]) for b in range(B)])
        q, qdot = xn[:, :3], xn[:, 3:]


def test_device_entry_is_ordered_with_the_callers_stream(rt):
    """Inputs produced by torch ops on the current stream, outputs consumed by torch ops on the same stream,
    no device-wide synchronize in between (stream = None -> torch's current stream; NULL at the ABI = the legacy
    null stream).  The result must equal the host-pointer entry."""
    import torch
    sc = rt["make_scenario"]("cfg2", B=256, seed=18)
    s = rt["Solver"](sc.desc, max_batch=256)
    host = s.solve(sc.xinit, sc.x0, sc.params)
    dev = torch.device("cuda:0")
    big = torch.randn(4096, 4096, device=dev)
    for use_side_stream in (False, True):
        ctx = torch.cuda.stream(torch.cuda.Stream()) if use_side_stream else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            junk = (big @ big).sum()                      # keeps the stream busy ahead of the input ops
            tx = torch.from_numpy(sc.xinit).to(dev) * 2.0
            t0 = torch.from_numpy(sc.x0).to(dev) * 2.0
            tp = torch.from_numpy(sc.params).to(dev) * 2.0
            tx, t0, tp = tx * 0.5, t0 * 0.5, tp * 0.5       # exact: the inputs exist only once these ops have run
            z = torch.full((256, sc.desc["N"], s.nvar), float("nan"), dtype=torch.float64, device=dev)
            ef = torch.full((256,), -99, dtype=torch.int32, device=dev); it = torch.zeros(256, dtype=torch.int32, device=dev)
            kkt = torch.zeros(256, dtype=torch.float64, device=dev); obj = torch.zeros(256, dtype=torch.float64, device=dev)
            s.solve_device(256, tx, t0, tp, z, ef, it, kkt, obj)          # stream=None
            zsum = z.sum(dim=(1, 2))                        # consumer op on the same stream, no synchronize
            ef2 = ef + 0
            ef2_host = ef2.cpu().numpy()                    # (copy on the same stream: the solve call itself does not block)
        assert np.array_equal(ef2_host, host["exitflag"])
        torch.cuda.synchronize()
        assert np.array_equal(z.cpu().numpy(), host["z"])
        assert torch.equal(zsum, z.sum(dim=(1, 2)))       # the consumer op saw the finished plan, not the NaN fill
        del junk
    s.close()


def test_concurrent_handles_return_serial_results(rt):
    """Six handles (point robot, boxer, arm, a gantry, an arm on the pass kernels' sizes and a row-described model) driven
    by six host threads on six streams at once, five launches each, twice the chip's wavefronts in flight: every launch
    returns bit for bit what the same handle returns when it runs alone -- handles share no state on the device."""
    import threading
    import torch
    dev = torch.device("cuda:0")
    cases = [("cfg2", 4096, 91), ("cfg3", 2048, 92), ("cfg4", 1024, 93), ("chain2", 1024, 94), ("chain8", 256, 95), ("plug_panda", 512, 96)]
    work = []
    for name, B, seed in cases:
        sc = rt["make_scenario"](name, B=B, seed=seed)
        s = rt["Solver"](sc.desc, max_batch=B)
        ref = s.solve(sc.xinit, sc.x0, sc.params)
        N, nv = sc.desc["N"], s.nvar
        t = dict(x=torch.from_numpy(sc.xinit).to(dev), z0=torch.from_numpy(sc.x0).to(dev), p=torch.from_numpy(sc.params).to(dev),
                 z=[torch.empty((B, N, nv), dtype=torch.float64, device=dev) for _ in range(5)],
                 e=[torch.empty(B, dtype=torch.int32, device=dev) for _ in range(5)], i=[torch.empty(B, dtype=torch.int32, device=dev) for _ in range(5)],
                 k=torch.empty(B, dtype=torch.float64, device=dev), o=torch.empty(B, dtype=torch.float64, device=dev))
        work.append((s, B, ref, t, torch.cuda.Stream()))
    torch.cuda.synchronize()
    errors = []

    def run(w):
        s, B, ref, t, st = w
        try:
            for r in range(5):
                s.solve_device(B, t["x"], t["z0"], t["p"], t["z"][r], t["e"][r], t["i"][r], t["k"], t["o"], stream=st.cuda_stream)
        except Exception as ex:      # noqa: BLE001 -- reported by the main thread
            errors.append(ex)

    th = [threading.Thread(target=run, args=(w,)) for w in work]
    [x.start() for x in th]
    [x.join() for x in th]
    torch.cuda.synchronize()
    assert not errors, errors
    for (s, B, ref, t, st), (name, _, _) in zip(work, cases):
        for r in range(5):
            assert np.array_equal(t["z"][r].cpu().numpy(), ref["z"]), (name, r)
            assert np.array_equal(t["e"][r].cpu().numpy(), ref["exitflag"]) and np.array_equal(t["i"][r].cpu().numpy(), ref["iters"]), (name, r)
        s.close()


@pytest.mark.parametrize("name,B,budget", [("cfg4", 96, 13), ("cfg2", 128, 14), ("cfg3", 64, 24)])
def test_pass_budget_cuts_only_the_unfinished(rt, name, B, budget):
    """rmpc_set_pass_budget: instances that finish within the budget return exactly what they return without one;
    the others come back with exit flag 0 and their last accepted iterate; the instances cut are those the
    oracle needs more passes for (one per horizon evaluation: start point, trial points, recomputed steps)."""
    from oracle.oracle import lib as olib
    sc = rt["make_scenario"](name, B=B, seed=19)
    s = rt["Solver"](sc.desc, max_batch=B)
    free = s.solve(sc.xinit, sc.x0, sc.params)
    s.set_pass_budget(budget)
    cut = s.solve(sc.xinit, sc.x0, sc.params)
    if name == "cfg4":
        assert s.last_passes() <= budget
    s.set_pass_budget(0)
    again = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    assert np.array_equal(again["exitflag"], free["exitflag"]) and np.array_equal(again["z"], free["z"])
    done = cut["exitflag"] != 0
    assert 0 < done.sum() < B, (done.sum(), "choose a budget that cuts some instances and not all")
    assert np.array_equal(cut["exitflag"][done], free["exitflag"][done])
    assert np.array_equal(cut["iters"][done], free["iters"][done])
    assert np.array_equal(cut["z"][done], free["z"][done])
    assert np.all(np.isfinite(cut["z"][~done])) and np.all(cut["iters"][~done] < budget)
    o = rt["Oracle"](sc.desc)
    passes = np.zeros(B, dtype=int)
    for b in range(B):
        o.solve(sc.xinit[b], sc.x0[b], sc.params[b])
        passes[b] = olib().orc_last_passes()
    assert ((passes <= budget) == done).mean() >= 0.9, (passes, done)


def test_is_fused_tells_which_solves_need_no_host_look(rt, monkeypatch):
    """rmpc_is_fused: the point robot, the diff-drive base and (round 4: k_fused_arm) the arms with 5 .. 7 joints solve in
    one launch when N <= 32; arms with 4 or 8 joints, arms with the slack variable and long horizons go through the pass
    kernels; RMPC_NO_FUSED=1 (read at rmpc_create) sends everything through the pass kernels."""
    monkeypatch.delenv("RMPC_NO_FUSED", raising=False)
    want = {"cfg2": True, "cfg3": True, "cfg4": True, "chain2": True, "chain5": True, "chain6": True, "chain4": False, "chain8": False}
    for name, fused in want.items():
        sc = rt["make_scenario"](name, B=4, seed=1)
        s = rt["Solver"](sc.desc, max_batch=4)
        assert s.is_fused() == fused, name
        s.close()
    for name in ("cfg2", "cfg4"):
        sc = rt["make_scenario"](name, B=4, seed=1, time_horizon=40)
        s = rt["Solver"](sc.desc, max_batch=4)
        assert not s.is_fused(), name
        s.close()
    monkeypatch.setenv("RMPC_NO_FUSED", "1")
    for name in ("cfg2", "cfg4"):
        sc = rt["make_scenario"](name, B=4, seed=1)
        s = rt["Solver"](sc.desc, max_batch=4)
        assert not s.is_fused(), name
        s.close()


def test_budgeted_pass_kernel_solve_needs_no_host_look(rt, monkeypatch):
    """rmpc_is_async: under a pass budget a solve of the pass kernels (the arm with RMPC_NO_FUSED=1) is enqueued whole (no
    host look) and returns before the device has finished; its results are those of the host-driven loop under the same
    budget (bit for bit), the non-empty passes are counted on the device, and the call is ordered with the caller's stream."""
    import torch
    monkeypatch.setenv("RMPC_NO_FUSED", "1")
    B = 192
    sc = rt["make_scenario"]("cfg4", B=B, seed=11)
    s = rt["Solver"](sc.desc, max_batch=B)
    assert not s.is_fused() and not s.is_async()
    full = s.solve(sc.xinit, sc.x0, sc.params)
    need = s.last_passes()
    exact = None
    for budget in (need + 6, 9):
        s.set_pass_budget(budget)
        assert s.is_async()
        r = s.solve(sc.xinit, sc.x0, sc.params)
        if exact is None:
            exact = s.last_passes()   # (the host loop looks every few passes: its count is rounded up)
            assert need - 4 <= exact <= need and exact > 9
        else:
            assert s.last_passes() == budget
        if budget > need:
            for key in ("z", "exitflag", "iters", "obj", "kkt"):
                assert np.array_equal(r[key], full[key]), key
        else:
            cut = r["exitflag"] == 0
            assert cut.any() and np.array_equal(r["z"][~cut], full["z"][~cut])
    # stream order: outputs consumed by a later kernel on the same stream, no synchronisation in between
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xi, x0, pa = t(sc.xinit), t(sc.x0), t(sc.params)
    z = torch.zeros((B, s.N, s.nvar), dtype=torch.float64, device=dev)
    ef = torch.full((B,), -99, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
    kk = torch.zeros(B, dtype=torch.float64, device=dev); ob = torch.zeros(B, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        s.solve_device(B, xi, x0, pa, z, ef, it, kk, ob, stream=st.cuda_stream)
        zc = z.clone()
    torch.cuda.synchronize()
    assert np.array_equal(zc.cpu().numpy(), r["z"])
    s.close()
