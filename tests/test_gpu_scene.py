"""Next rows of the hot path on the device, through the C ABI:
scene -> all_parameters must be bit-identical to the host packer (which the CPU suite pins
against the reference's loops), the fused scene solve bit-identical to the plain solve, and
the closed-loop advance equal to the oracle's plant + the host warm-start shift."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import torch
    import __graft_entry__ as g
    g.build()
    from oracle.oracle import Oracle
    from robot_mpcs_amd._lib import Solver
    from robot_mpcs_amd.scenarios import (BOXER_LIMITS, BOXER_LIMITS_U, PANDA_LIMITS, PANDA_LIMITS_U, POINT_LIMITS,
                                          POINT_LIMITS_U, make_scenario)
    return dict(torch=torch, Oracle=Oracle, Solver=Solver, make_scenario=make_scenario,
                limits=dict(pointRobot=(POINT_LIMITS, POINT_LIMITS_U), boxer=(BOXER_LIMITS, BOXER_LIMITS_U),
                            panda=(PANDA_LIMITS, PANDA_LIMITS_U)))


def _scene_tensors(rt, sc):
    torch = rt["torch"]
    dev = torch.device("cuda:0")
    B = sc.B
    robot = sc.setup["mpc"]["model_name"]
    lim, limu = rt["limits"][robot]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    ten = dict(goal=t(sc.extra["goal"]), r_body=t(np.full(B, sc.extra["r_body"])),
               lower_limits=t(np.tile(lim[0], (B, 1))), upper_limits=t(np.tile(lim[1], (B, 1))),
               lower_limits_u=t(np.tile(limu[0], (B, 1))), upper_limits_u=t(np.tile(limu[1], (B, 1))))
    if "obst_dyn" in sc.extra:
        ten["obst_dyn"] = t(sc.extra["obst_dyn"])
    elif "obst_pos" in sc.extra:
        rad = sc.extra.get("obst_radius", np.full(sc.extra["obst_pos"].shape[:2], 0.1))
        ten["obst"] = t(np.concatenate([sc.extra["obst_pos"], rad[:, :, None]], axis=2))
    return ten


@pytest.mark.parametrize("name,B", [("cfg2", 300), ("cfg3", 130), ("cfg4", 70)])
def test_scene_packing_is_bit_identical_to_host_packer(rt, name, B):
    torch = rt["torch"]
    sc = rt["make_scenario"](name, B=B, seed=17)
    s = rt["Solver"](sc.desc, max_batch=B)
    ten = _scene_tensors(rt, sc)
    scene = s.make_scene(sc.setup["mpc"]["weights"], **ten)
    out = torch.full((B, sc.desc["N"] * sc.desc["npar"]), float("nan"), dtype=torch.float64, device="cuda:0")
    s.pack_scene_device(B, scene, out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), sc.params)
    s.close()


def test_linear_constraint_scene(rt):
    torch = rt["torch"]
    sc = rt["make_scenario"]("boxer", B=9, seed=2)
    N, nob = sc.desc["N"], sc.desc["nobst"]
    rng = np.random.default_rng(0)
    lin = rng.normal(size=(9, N, nob, 4))
    sc.packer.setLinearConstraints(lin, sc.extra["r_body"])
    s = rt["Solver"](sc.desc, max_batch=9)
    ten = _scene_tensors(rt, sc)
    ten["lin_constrs"] = torch.from_numpy(lin).to("cuda:0")
    scene = s.make_scene(sc.setup["mpc"]["weights"], **ten)
    out = torch.zeros((9, N * sc.desc["npar"]), dtype=torch.float64, device="cuda:0")
    s.pack_scene_device(9, scene, out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), sc.packer.params)
    s.close()


@pytest.mark.parametrize("name,B", [("cfg2", 256), ("cfg3", 96)])
def test_fused_scene_solve_equals_plain_solve(rt, name, B):
    torch = rt["torch"]
    sc = rt["make_scenario"](name, B=B, seed=23)
    s = rt["Solver"](sc.desc, max_batch=B)
    plain = s.solve(sc.xinit, sc.x0, sc.params)
    dev = "cuda:0"
    ten = _scene_tensors(rt, sc)
    scene = s.make_scene(sc.setup["mpc"]["weights"], **ten)
    N, nv = sc.desc["N"], s.nvar
    tx = torch.from_numpy(sc.xinit).to(dev); t0 = torch.from_numpy(sc.x0).to(dev)
    z = torch.empty((B, N, nv), dtype=torch.float64, device=dev)
    ef = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
    kkt = torch.empty(B, dtype=torch.float64, device=dev); obj = torch.empty(B, dtype=torch.float64, device=dev)
    s.solve_scene_device(B, scene, tx, t0, z, ef, it, kkt, obj)
    torch.cuda.synchronize()
    assert np.array_equal(z.cpu().numpy(), plain["z"]) and np.array_equal(ef.cpu().numpy(), plain["exitflag"])
    assert np.array_equal(it.cpu().numpy(), plain["iters"])
    # the two halves of the scene solve, for callers that pack the scenes of several handles before the first solve
    z2 = torch.full_like(z, float("nan")); ef2 = torch.zeros_like(ef); it2 = torch.zeros_like(it)
    with pytest.raises(Exception):    # nothing packed for this batch size yet
        s.solve(sc.xinit, sc.x0, sc.params)          # (a plain solve overwrites the workspace parameters)
        s.solve_packed_device(B, tx, t0, z2, ef2, it2, kkt, obj)
    s.pack_scene_workspace(B, scene)
    s.solve_packed_device(B, tx, t0, z2, ef2, it2, kkt, obj)
    torch.cuda.synchronize()
    assert np.array_equal(z2.cpu().numpy(), plain["z"]) and np.array_equal(ef2.cpu().numpy(), plain["exitflag"])
    assert np.array_equal(it2.cpu().numpy(), plain["iters"])
    s.close()


@pytest.mark.parametrize("name,prev", [("cfg2", True), ("cfg2", False), ("cfg3", True), ("cfg4", True)])
def test_advance_matches_oracle_plant_and_host_shift(rt, name, prev):
    torch = rt["torch"]
    B = 40
    sc = rt["make_scenario"](name, B=B, seed=31)
    o = rt["Oracle"](sc.desc)
    s = rt["Solver"](sc.desc, max_batch=B)
    r = s.solve(sc.xinit, sc.x0, sc.params)
    dev = "cuda:0"
    tz = torch.from_numpy(r["z"]).to(dev); tx = torch.from_numpy(sc.xinit.copy()).to(dev)
    t0 = torch.full((B, sc.desc["N"], s.nvar), float("nan"), dtype=torch.float64, device=dev)
    s.advance_device(B, tz, tx, t0, previous_plan=prev)
    torch.cuda.synchronize()
    nxs = sc.desc["nx"] + sc.desc["ns"]
    xn = np.stack([o.dynamics(sc.xinit[b], r["z"][b, 0, nxs:]) for b in range(B)])
    np.testing.assert_allclose(tx.cpu().numpy(), xn, rtol=0, atol=1e-14)
    pk = sc.packer
    pk.reset()
    pk._initial_step = False
    x0_host = pk.setX0(tx.cpu().numpy(), r["z"], "previous_plan" if prev else "current_state")
    assert np.array_equal(t0.cpu().numpy(), x0_host)
    s.close()


def test_device_closed_loop_reaches_goal(rt):
    """The reference's point-robot example scenario (pointRobot_example.py:31-65: goal (8.2, -0.2),
    obstacle (4, -0.5) r = 1), 64 replicas with perturbed start positions, 60 control steps (3 s) entirely
    on the device (scene solve + advance).  Every solve must succeed and every robot must head for the
    goal around the obstacle with the margin kept.  (Run much longer and the example's own constants
    make the NLP infeasible: 1 m/s^2 of braking cannot stop a robot that has accelerated for 4 s before
    the joint limit at x = 10, and a 1 s horizon does not see that coming.)"""
    torch = rt["torch"]
    B = 64
    sc = rt["make_scenario"]("pointRobot", B=B, seed=5)
    rng = np.random.default_rng(3)
    sc.xinit[:, 0:2] += rng.uniform(-0.5, 0.5, size=(B, 2))
    sc.x0[:, :, 0:6] = sc.xinit[:, None, :]
    s = rt["Solver"](sc.desc, max_batch=B)
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    lim, limu = rt["limits"]["pointRobot"]
    obst = np.tile(np.array([[[4.0, -0.5, 0.0, 1.0]]]), (B, 1, 1))
    scene = s.make_scene(sc.setup["mpc"]["weights"], goal=t(sc.extra["goal"]), r_body=t(np.full(B, 0.3)), obst=t(obst),
                         lower_limits=t(np.tile(lim[0], (B, 1))), upper_limits=t(np.tile(lim[1], (B, 1))),
                         lower_limits_u=t(np.tile(limu[0], (B, 1))), upper_limits_u=t(np.tile(limu[1], (B, 1))))
    N, nv = sc.desc["N"], s.nvar
    tx = t(sc.xinit); t0 = t(sc.x0)
    z = torch.empty((B, N, nv), dtype=torch.float64, device=dev)
    ef = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
    kkt = torch.empty(B, dtype=torch.float64, device=dev); obj = torch.empty(B, dtype=torch.float64, device=dev)
    goal = sc.extra["goal"][:, :2]
    d0 = np.linalg.norm(sc.xinit[:, :2] - goal, axis=1)
    min_clear = np.full(B, np.inf)
    for step in range(60):
        s.solve_scene_device(B, scene, tx, t0, z, ef, it, kkt, obj)
        assert int((ef < 0).sum().item()) == 0, (step, ef.cpu().numpy())
        s.advance_device(B, z, tx, t0, previous_plan=False)
        if step % 4 == 0:
            x = tx.cpu().numpy()
            c = np.linalg.norm(np.stack([x[:, 0] - 4.0, x[:, 1] + 0.5, np.full(B, 0.05)], 1), axis=1) - 1.0 - 0.3
            min_clear = np.minimum(min_clear, c)
    torch.cuda.synchronize()
    x = tx.cpu().numpy()
    d1 = np.linalg.norm(x[:, :2] - goal, axis=1)
    assert np.all(d1 < d0 - 3.0) and min_clear.min() > -1e-6
    s.close()


def test_free_space_decomposition_matches_reference_restatement(rt):
    """Lidar-like clouds (64 rays), seeds along a plan, K = 1 (shipped boxer config) and K = 5."""
    torch = rt["torch"]
    from oracle.fsd_numpy import free_space_decomposition
    from robot_mpcs_amd._lib import free_space_decomposition_device
    rng = np.random.default_rng(11)
    B, N, P = 37, 10, 64
    ang = np.linspace(-np.pi + np.pi / 8, -np.pi / 8, P)
    rngs = rng.uniform(0.5, 7.0, size=(B, P))
    centre = rng.uniform(-3, 3, size=(B, 1, 2))
    pts = np.concatenate([centre + rngs[:, :, None] * np.stack([np.cos(ang), np.sin(ang)], 1)[None], np.full((B, P, 1), 0.02)], axis=2)
    seeds = np.concatenate([centre + rng.normal(0, 0.4, size=(B, N, 2)), np.full((B, N, 1), 0.02)], axis=2)
    for K in (1, 5):
        out = torch.full((B, N, K, 4), float("nan"), dtype=torch.float64, device="cuda:0")
        free_space_decomposition_device(torch.from_numpy(pts).to("cuda:0"), torch.from_numpy(seeds).to("cuda:0"), out, 5.0)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        ref = np.stack([[free_space_decomposition(pts[b], seeds[b, k], K, 5.0) for k in range(N)] for b in range(B)])
        assert np.array_equal(got, ref)
    # a cloud entirely outside max_radius: only dummy planes
    out = torch.zeros((1, 1, 2, 4), dtype=torch.float64, device="cuda:0")
    far = np.array([[[100.0, 0, 0], [0, 100.0, 0]]])
    free_space_decomposition_device(torch.from_numpy(far).to("cuda:0"), torch.zeros((1, 1, 3), dtype=torch.float64, device="cuda:0"), out, 5.0)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy()[0, 0], free_space_decomposition(far[0], np.zeros(3), 2, 5.0))


def test_free_space_decomposition_class_interface(rt):
    from oracle.fsd_numpy import free_space_decomposition
    from robot_mpcs_amd.utils.free_space_decomposition import FreeSpaceDecomposition
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-4, 4, size=(64, 2)), np.full((64, 1), 0.02)], axis=1)
    fsd = FreeSpaceDecomposition(number_constraints=3, max_radius=5.0)
    fsd.set_position(np.array([0.3, -0.2, 0.02]))
    fsd.compute_constraints(pts)
    ref = free_space_decomposition(pts, np.array([0.3, -0.2, 0.02]), 3, 5.0)
    assert list(fsd.asdict()) == ["constraint_0", "constraint_1", "constraint_2"]
    assert np.array_equal(fsd.aslist(), ref)
    assert np.array_equal(np.array(list(fsd.asdict().values())), ref)
