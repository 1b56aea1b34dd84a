"""Constraint plug-ins given as ROW DESCRIPTIONS (YAML block ``plugins``; include/rmpc.h ``RMPC_MOD_ROWS``): the
counterpart of a user class named in ``mpc.constraints`` (reference ``InequalityManager.py:17-21``) for variants of the
six built-in kinds -- other links, pairs, joints or inputs, parameter entries of their own -- lowered through the runtime
row tables of the oracle and of the HIP library, with no rebuild."""
import ctypes as C

import numpy as np
import pytest

from robot_mpcs_amd.scenarios import make_scenario


def _oracle(sc):
    from oracle.oracle import Oracle
    return Oracle(sc.desc).solve_batch(sc.xinit, sc.x0, sc.params)


def test_described_rows_reach_the_descriptor():
    sc = make_scenario("plug_point", B=2, seed=1)
    d = sc.desc
    assert d["module_kind"] == [6, 2, 6, 6]                     # RMPC_MOD_ROWS next to the built-in self-collision module
    rows = d["xrows"]
    assert len(rows) == 3 + 6 + 6 and d["nh"] == 15
    pm = sc.model._paramMap
    # spheres: links outer, spheres inner, 4 parameters each from the plug-in's own entry
    assert [r[4] for r in rows[:3]] == [pm["KeepOutSpheres"][0] + 4 * i for i in range(3)]
    # limits: lower then upper per index, on q (variables 0 .. 2) and on u (variables nx .. nx + 2)
    assert [(r[2], r[3]) for r in rows[3:9]] == [(0, 1), (0, -1), (1, 1), (1, -1), (2, 1), (2, -1)]
    assert [r[2] for r in rows[9:]] == [6, 6, 7, 7, 8, 8]
    assert pm["wconstr"] and len(pm["wconstr"]) == 4             # one inverse-barrier weight per module, described ones too
    # a name that is neither a class nor described fails like in the reference
    with pytest.raises(AttributeError):
        make_scenario("cfg2", B=1, constraints=["RadialConstraints", "NoSuchConstraints"])


def test_described_rows_survive_the_solver_directory(tmp_path):
    """generateSolver writes the rows into rmpc_model.yaml; the planner reads them back (mpcPlanner.py:73 counterpart)."""
    import os
    from robot_mpcs_amd.models.mpcModel import DESCRIPTOR_FILE, load_descriptor
    sc = make_scenario("plug_panda", B=1, seed=1)
    target = sc.model.generateSolver(location=str(tmp_path) + "/")
    back = load_descriptor(os.path.join(target, DESCRIPTOR_FILE))
    assert back["xrows"] == sc.desc["xrows"] and back["module_kind"] == [0, 6, 3, 5, 6, 6]
    import yaml
    pm = yaml.safe_load(open(os.path.join(target, "paramMap.yaml")))
    assert len(pm["KeepOut"]) == 4 and len(pm["WristSpeed_lower"]) == 2


@pytest.mark.parametrize("builtin,twin", [("cfg2", "plug_point"), ("wc_boxer", "plug_boxer")])
def test_described_twin_is_the_builtin_configuration_for_the_oracle(builtin, twin):
    """plug_pointRobotMpc.yaml describes cfg2's sphere, joint-limit and input-limit rows one by one, plug_boxerMpc.yaml the
    planes and the speed limits of wc_boxer (a diff-drive base; limits on the last two states): same rows, same order,
    same numbers through other parameter entries -- the restatement returns bit-identical plans."""
    a, b = make_scenario(builtin, B=24, seed=5), make_scenario(twin, B=24, seed=5)
    ra, rb = _oracle(a), _oracle(b)
    assert np.array_equal(ra["exitflag"], rb["exitflag"]) and np.array_equal(ra["iters"], rb["iters"])
    assert np.array_equal(ra["z"], rb["z"])


def test_described_variants_on_the_arm_hold_at_the_oracles_solution():
    """plug_pandaMpc.yaml: cfg4 with a keep-out sphere of its own for links 5 and 7, the self-collision pair as a
    described module and speed limits on the wrist joints.  Every instance converges; the described rows hold along the plan."""
    from robot_mpcs_amd.utils.urdf_chain import fk_positions
    sc = make_scenario("plug_panda", B=16, seed=5)
    r = _oracle(sc)
    assert np.all(r["exitflag"] >= 1)
    z = r["z"]
    pm, p3 = sc.model._paramMap, sc.packer.p3
    q, v = z[:, :, :7], z[:, :, 7:14]
    lo, hi = p3[:, :, pm["WristSpeed_lower"]], p3[:, :, pm["WristSpeed_upper"]]
    assert np.all(v[:, 1:, 5:7] >= lo[:, 1:] - 1e-7) and np.all(v[:, 1:, 5:7] <= hi[:, 1:] + 1e-7)
    assert np.abs(v[:, 1:, 5]).max() > 0.045          # (the limit of 0.05 rad/s is active: without it joint 6 reaches 0.08)
    keep = p3[:, 0, pm["KeepOut"]]
    rb = p3[:, 0, pm["r_body"][0]]
    rows = [x for x in sc.desc["xrows"] if x[1] == 0]
    for k in range(1, sc.desc["N"]):
        fk = fk_positions(sc.desc["joints"], q[:, k], [x[2] for x in rows])
        for x in rows:
            dist = np.linalg.norm(fk[x[2]] - keep[:, :3], axis=1) - keep[:, 3] - rb
            assert np.all(dist >= -1e-7)


def test_library_validates_row_descriptions():
    """build_model (host code of the library, no GPU needed): a module's rows must be all on states or all on inputs; a
    sphere must be addressable as an entry of the obstacle list; a descriptor of the 0.2.0 size is still accepted."""
    from robot_mpcs_amd import _lib
    L = _lib.load_library()
    L.rmpc_workspace_bytes.restype = C.c_int64
    sc = make_scenario("plug_point", B=1, seed=1)
    good = _lib.make_desc(sc.desc)
    assert L.rmpc_workspace_bytes(C.byref(good), 4) > 0
    mixed = dict(sc.desc)
    rows = [list(x) for x in sc.desc["xrows"]]
    rows[3][2] = 6                                   # one of the q-limit rows moved onto an input
    mixed["xrows"] = rows
    assert L.rmpc_workspace_bytes(C.byref(_lib.make_desc(mixed)), 4) < 0
    assert b"all be on states or all on inputs" in L.rmpc_last_error()
    off = dict(sc.desc)
    rows = [list(x) for x in sc.desc["xrows"]]
    rows[1][4] += 1                                  # a sphere that is no whole entry behind the first one
    off["xrows"] = rows
    assert L.rmpc_workspace_bytes(C.byref(_lib.make_desc(off)), 4) < 0
    assert b"multiple of 4" in L.rmpc_last_error()
    # the descriptor without the xrow arrays (ABI 0.2.0): accepted, no described rows
    old = _lib.make_desc(make_scenario("cfg2", B=1).desc)
    old.struct_size = _lib.RmpcDesc.n_xrows.offset
    assert L.rmpc_workspace_bytes(C.byref(old), 4) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,seed,fused", [("plug_point", 160, 11, True), ("plug_panda", 48, 12, False), ("plug_boxer", 96, 13, True)])
def test_described_rows_on_the_gpu_match_the_oracle(name, B, seed, fused):
    """HIP library vs restatement on the described plug-ins: flags, iteration counts, plans.  The point robot stays in
    k_fused; the arm's wrist speed limits break the uniform row structure k_fused_arm relies on, so it runs the pass kernels."""
    from robot_mpcs_amd._lib import Solver
    sc = make_scenario(name, B=B, seed=seed)
    cpu = _oracle(sc)
    s = Solver(sc.desc, max_batch=B)
    assert s.is_fused() == fused
    gpu = s.solve(sc.xinit, sc.x0, sc.params)
    s.close()
    same = gpu["exitflag"] == cpu["exitflag"]
    assert same.mean() >= 0.99
    ok = same & (cpu["exitflag"] >= 1)
    assert np.array_equal(gpu["iters"][ok], cpu["iters"][ok]) or np.mean(gpu["iters"][ok] == cpu["iters"][ok]) >= 0.98
    nxs = sc.desc["nx"] + sc.desc["ns"]
    du = np.abs(gpu["z"][ok][:, 0, nxs:] - cpu["z"][ok][:, 0, nxs:]).max()
    assert du <= 1e-6 * max(1.0, np.abs(cpu["z"][ok][:, 0, nxs:]).max()), du


@pytest.mark.gpu
def test_described_twin_of_wc_boxer_is_wc_boxer_on_the_gpu():
    """Both run k_fused over the runtime tables, and the tables come out equal entry for entry: bit-identical plans."""
    from robot_mpcs_amd._lib import Solver
    a, b = make_scenario("wc_boxer", B=128, seed=22), make_scenario("plug_boxer", B=128, seed=22)
    sa = Solver(a.desc, max_batch=128); ra = sa.solve(a.xinit, a.x0, a.params); sa.close()
    sb = Solver(b.desc, max_batch=128); rb = sb.solve(b.xinit, b.x0, b.params); sb.close()
    assert np.array_equal(ra["exitflag"], rb["exitflag"]) and np.array_equal(ra["iters"], rb["iters"])
    assert np.array_equal(ra["z"], rb["z"])


@pytest.mark.gpu
def test_described_twin_of_cfg2_is_cfg2_on_the_gpu():
    """The twin's row tables and parameter layout come out entry for entry as cfg2's, so rmpc_create gives it cfg2's
    generated view (a view is chosen by comparing tables, not names): bit-identical plans.  (Should the layouts ever
    differ, the twin runs over the runtime tables: flags and iteration counts agree, plans to 1e-8 as in test_spec_gen.)"""
    from robot_mpcs_amd._lib import Solver
    a, b = make_scenario("cfg2", B=256, seed=21), make_scenario("plug_point", B=256, seed=21)
    sa = Solver(a.desc, max_batch=256); ra = sa.solve(a.xinit, a.x0, a.params); va = sa.spec_name(); sa.close()
    sb = Solver(b.desc, max_batch=256); rb = sb.solve(b.xinit, b.x0, b.params); vb = sb.spec_name(); sb.close()
    if va == vb:
        assert np.array_equal(ra["exitflag"], rb["exitflag"]) and np.array_equal(ra["iters"], rb["iters"])
        assert np.array_equal(ra["z"], rb["z"])
        return
    assert np.mean(ra["exitflag"] == rb["exitflag"]) >= 0.99
    same = (ra["exitflag"] == rb["exitflag"]) & (ra["iters"] == rb["iters"]) & np.isin(ra["exitflag"], (1, 2))
    assert same.mean() >= 0.97
    assert np.abs(ra["z"][same] - rb["z"][same]).max() <= 1e-8
