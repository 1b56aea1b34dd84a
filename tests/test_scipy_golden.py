"""Independent full-horizon golden vectors (tests/golden/scipy_full_horizon.npz, written by
tests/golden/make_scipy_golden.py): scipy solutions of BASELINE configs[1..3] at their real horizons (pointRobot
N=30, boxer N=30 with slack, panda N=20), computed from the numpy restatement of the NLP with finite-difference
Jacobians -- neither rmpc_oracle.c nor the HIP kernels took part.

  * 64 seeded instances per robot through SLSQP (active-set SQP);
  * the first few again through trust-constr (trust-region interior point), where it converged within its
    iteration budget (point robot 6, panda 2; the boxer's two runs did not and are recorded as such);
  * one warm-started closed loop of 6 control steps per robot (plant = the numpy ERK2 map, every solve from the
    shifted previous plan);
  * the boxer's first 8 instances solved with Heun's rule instead of the explicit midpoint rule: how far the applied
    control moves if FORCES Pro's (unverifiable) ERK2 tableau is the other one.

CPU: the C oracle against them; GPU: the HIP solver against them.  Tolerance: applied control
|u_1 - u_1^scipy| <= 1e-4 and objective within 1e-6 relative when both land in the same basin."""
import hashlib
import os

import numpy as np
import pytest

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "scipy_full_horizon.npz"))
CASES = ["cfg2", "cfg3", "cfg4"]
B_GOLD = 64
# instances (of 64) on which the interior-point method and SLSQP must land in the same local solution.  The point
# robot's and the arm's NLPs behave: all of them.  The unicycle NLP is non-convex: from the same start the two
# methods settle in different local solutions for 14 of the 64 boxers (12 times SLSQP's is the better one, twice
# the interior-point method's) -- recorded, not hidden.
SAME_BASIN_MIN = {"cfg2": 64, "cfg3": 50, "cfg4": 64}
TC_CONVERGED = {"cfg2": 6, "cfg3": 0, "cfg4": 2}     # trust-constr runs that converged within their iteration budget
# measured sensitivity of the applied control to the ERK2 tableau (boxer, input scale 10): <= 2.6e-3 absolute, the
# objective moves by <= 1.6e-4 relative
HEUN_DU_BOUND, HEUN_OBJ_BOUND = 5e-3, 5e-4


def _inputs(name):
    from robot_mpcs_amd.scenarios import make_scenario
    sc = make_scenario(name, B=B_GOLD, seed=int(GOLD[name + "_seed"][0]))
    # the fixture's inputs are the scenario generator's (same seed): the stored hash pins that
    hsh = hashlib.sha256()
    for a in (sc.xinit, sc.x0, sc.params):
        hsh.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    assert hsh.hexdigest() == str(GOLD[name + "_inputs_sha256"])
    return sc


def _compare(name, z, obj, flags, nxs):
    Z1, fun = GOLD[name + "_z1"], GOLD[name + "_fun"]
    assert np.all(GOLD[name + "_viol"] <= 1e-8)              # scipy's own points are feasible
    assert np.all(np.isin(flags, (1, 2)))
    same = np.abs(obj - fun) <= 1e-6 * np.maximum(1.0, np.abs(fun))       # same basin
    du = np.abs(z[:, 0, nxs:] - Z1[:, nxs:]).max(axis=1)
    assert same.sum() >= SAME_BASIN_MIN[name], (name, int(same.sum()))
    assert np.all(du[same] <= 1e-4), (name, du[same].max())
    # second algorithm family: trust-constr, where it converged
    nt = TC_CONVERGED[name]
    if nt:
        assert np.all(GOLD[name + "_tc_viol"][:nt] <= 1e-8)
        assert np.all(np.abs(GOLD[name + "_tc_fun"][:nt] - fun[:nt]) <= 1e-6 * np.abs(fun[:nt]))      # SLSQP and trust-constr agree
        dut = np.abs(z[:nt, 0, nxs:] - GOLD[name + "_tc_z1"][:nt, nxs:]).max(axis=1)
        assert np.all(dut[same[:nt]] <= 1e-4), (name, dut)


def test_erk2_tableau_sensitivity_is_bounded():
    """Heun's rule instead of the explicit midpoint rule (the tableau FORCES Pro uses is not verifiable here): the
    boxer's applied control moves by at most HEUN_DU_BOUND, the objective by HEUN_OBJ_BOUND relative -- the size of the
    modelling uncertainty that stays [UNVERIFIED] (the double integrator of the other two robots is integrated
    exactly by every ERK2)."""
    nxs = 9
    n = GOLD["cfg3_heun_z1"].shape[0]
    assert np.all(GOLD["cfg3_heun_viol"] <= 1e-8)
    du = np.abs(GOLD["cfg3_heun_z1"][:, nxs:] - GOLD["cfg3_z1"][:n, nxs:]).max(axis=1)
    dobj = np.abs(GOLD["cfg3_heun_fun"] - GOLD["cfg3_fun"][:n]) / np.abs(GOLD["cfg3_fun"][:n])
    assert du.max() <= HEUN_DU_BOUND and dobj.max() <= HEUN_OBJ_BOUND, (du, dobj)
    assert du.max() > 1e-6     # (the line is not vacuous: the tableau does matter at this level)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_full_horizon_scipy(name, oracle_lib):
    from oracle.oracle import Oracle
    sc = _inputs(name)
    o = Oracle(sc.desc)
    r = o.solve_batch(sc.xinit, sc.x0, sc.params)
    _compare(name, r["z"], r["obj"], r["exitflag"], o.nx + o.ns)


@pytest.mark.parametrize("name", CASES)
def test_oracle_closed_loop_matches_scipy(name, oracle_lib):
    """Warm-started closed loop (shifted plan + multipliers of the previous solve, orc_solve_warm) against scipy's
    loop from the same start: the applied control of every control step within 1e-4."""
    from oracle.oracle import Oracle
    sc = _inputs(name)
    o = Oracle(sc.desc)
    nxs = o.nx + o.ns
    lx, lz1 = GOLD[name + "_loop_x"], GOLD[name + "_loop_z1"]
    x, x0, duals = sc.xinit[0].copy(), sc.x0[0].copy(), None
    for t in range(lx.shape[0]):
        assert np.abs(x - lx[t]).max() <= 1e-5, (name, t)        # the two loops see the same states
        r = o.solve_warm(x, x0, sc.params[0], duals)
        assert r["exitflag"] in (1, 2), (name, t, r["exitflag"])
        duals = r["duals"]
        assert np.abs(r["z"][0, nxs:] - lz1[t, nxs:]).max() <= 1e-4, (name, t)
        x = o.dynamics(x, r["z"][0, nxs:])
        x0 = np.vstack([r["z"][1:], r["z"][-1:]])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_solver_matches_full_horizon_scipy(name):
    import __graft_entry__ as g
    g.build()
    from robot_mpcs_amd._lib import Solver
    sc = _inputs(name)
    s = Solver(sc.desc, max_batch=B_GOLD)
    r = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    _compare(name, r["z"], r["obj"], r["exitflag"], sc.desc["nx"] + sc.desc["ns"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_closed_loop_matches_scipy(name):
    """The device-resident warm-started loop (rmpc_set_warm_start + rmpc_advance_device: plant step with the model's
    ERK2 map, shifted plan, multipliers of the previous solve) against scipy's loop: applied control of every control
    step within 1e-4, states within 1e-5."""
    import torch
    import __graft_entry__ as g
    g.build()
    from robot_mpcs_amd._lib import Solver
    sc = _inputs(name)
    d = sc.desc
    nxs, nx = d["nx"] + d["ns"], d["nx"]
    lx, lz1 = GOLD[name + "_loop_x"], GOLD[name + "_loop_z1"]
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    s = Solver(d, max_batch=1)
    s.set_warm_start(True)
    x, x0, pa = t(sc.xinit[:1]), t(sc.x0[:1]), t(sc.params[:1])
    z = torch.empty((1, d["N"], s.nvar), dtype=torch.float64, device=dev)
    ef = torch.empty(1, dtype=torch.int32, device=dev); it = torch.empty(1, dtype=torch.int32, device=dev)
    kk = torch.empty(1, dtype=torch.float64, device=dev); ob = torch.empty(1, dtype=torch.float64, device=dev)
    for step in range(lx.shape[0]):
        assert np.abs(x.cpu().numpy()[0, :nx] - lx[step]).max() <= 1e-5, (name, step)
        s.solve_device(1, x, x0, pa, z, ef, it, kk, ob)
        torch.cuda.synchronize()
        assert int(ef.item()) in (1, 2), (name, step, int(ef.item()))
        assert np.abs(z.cpu().numpy()[0, 0, nxs:] - lz1[step, nxs:]).max() <= 1e-4, (name, step)
        s.advance_device(1, z, x, x0, previous_plan=True, exitflag=ef)
    s.close()
