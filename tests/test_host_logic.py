"""Host logic of the drop-in boundary: config surface, solver-directory
contract, parameter packing and warm start, checked against a per-element
restatement of the reference planner loops (mpcPlanner.py:83-236).  CPU only."""
import os

import numpy as np
import pytest
import yaml

from robot_mpcs_amd.models.mpcBase import MpcConfiguration
from robot_mpcs_amd.models.mpcModel import DESCRIPTOR_FILE, load_descriptor, normalise_descriptor
from robot_mpcs_amd.planner.mpcPlanner import _stage_key, solver_directory
from robot_mpcs_amd.planner.packing import ParamPacker
from robot_mpcs_amd.scenarios import CONFIG_DIR, build_model, make_scenario


def test_unknown_or_missing_config_keys_raise_type_error():
    _, setup = build_model(os.path.join(CONFIG_DIR, "pointRobotMpc.yaml"))
    bad = dict(setup["mpc"]); bad["not_a_key"] = 1
    with pytest.raises(TypeError):
        MpcConfiguration(**bad)
    bad = dict(setup["mpc"]); del bad["objectives"]
    with pytest.raises(TypeError):
        MpcConfiguration(**bad)


def test_unknown_constraint_class_name_raises_attribute_error():
    with pytest.raises(AttributeError):
        build_model(os.path.join(CONFIG_DIR, "pointRobotMpc.yaml"), constraints=["NoSuchConstraints"])


@pytest.mark.parametrize("cfg,dirname", [
    ("pointRobotMpc.yaml", "pointRobot_n3_005_H20_noSlack"),
    ("boxerMpc.yaml", "boxer_n3_01_H10_noSlack"),
    ("pandaMpc.yaml", "panda_n7_005_H20_noSlack"),
    ("cfg3_boxerMpc.yaml", "boxer_n3_01_H30"),
])
def test_solver_directory_contract(tmp_path, cfg, dirname):
    model, setup = build_model(os.path.join(CONFIG_DIR, cfg))
    target = model.generateSolver(location=str(tmp_path) + "/")
    assert os.path.basename(target) == dirname
    config = MpcConfiguration(**setup["mpc"])
    assert solver_directory(str(tmp_path) + "/", setup["mpc"]["model_name"], config) == target
    pm = yaml.safe_load(open(target + "/paramMap.yaml"))
    props = yaml.safe_load(open(target + "/properties.yaml"))
    assert pm == model._paramMap
    assert set(props) == {"nx", "nu", "npar", "ns", "m", "constraints"}
    assert props["npar"] == model._npar and props["constraints"] == setup["mpc"]["constraints"]
    d = load_descriptor(os.path.join(target, DESCRIPTOR_FILE))
    assert d == normalise_descriptor(model._model)
    assert np.isinf(d["ub"][d["nx"]]) if d["ns"] else True


def test_stage_keys_follow_horizon_digits():
    assert _stage_key(9, 1) == "x1" and _stage_key(10, 1) == "x01" and _stage_key(30, 30) == "x30"
    assert _stage_key(100, 7) == "x007"


def _reference_style_pack(pm, npar, N, cfg, calls):
    """Per-element loops exactly as the reference planner writes them."""
    p = np.zeros(npar * N)
    for i in range(N):
        if "wgoal" in pm:
            p[[npar * i + v for v in pm["wgoal"]]] = cfg.weights["w"]
        p[[npar * i + v for v in pm["wu"]]] = cfg.weights["wu"]
        if cfg.slack:
            p[[npar * i + v for v in pm["ws"]]] = cfg.weights["ws"]
    for name, args in calls:
        for i in range(N):
            if name == "radial":
                pos, rad, r_body = args
                p[npar * i + pm["r_body"][0]] = r_body
                for j in range(cfg.number_obstacles):
                    pj, rj = (pos[j], rad[j]) if j < len(pos) else ([-100, -100, -100], -100)
                    for m_i in range(3):
                        p[npar * i + pm["obst"][j * 4 + m_i]] = pj[m_i]
                    p[npar * i + pm["obst"][j * 4 + 3]] = rj
            elif name == "joint":
                for j in range(cfg.n):
                    p[npar * i + pm["lower_limits"][j]] = args[0][j]
                    p[npar * i + pm["upper_limits"][j]] = args[1][j]
            elif name == "input":
                for j in range(len(pm["lower_limits_u"])):
                    p[npar * i + pm["lower_limits_u"][j]] = args[0][j]
                    p[npar * i + pm["upper_limits_u"][j]] = args[1][j]
            elif name == "goal":
                for j in range(3):
                    p[npar * i + pm["goal"][j]] = args[j] if j < len(args) else 0
            elif name == "avoid":
                p[[npar * i + v for v in pm["wconstr"]]] = cfg.weights["wconstr"]
            elif name == "dyn":
                obst, dt, r = args
                nb = obst.size // 9
                for j in range(cfg.number_obstacles):
                    if j < nb:
                        pos, vel, acc = obst[9 * j: 9 * j + 3], obst[9 * j + 3: 9 * j + 6], obst[9 * j + 6: 9 * j + 9]
                    else:
                        pos, vel, acc = np.ones(3) * -100, np.zeros(3), np.zeros(3)
                    for m_i in range(3):
                        p[npar * i + pm["obst"][j * 4 + m_i]] = pos[m_i] + vel[m_i] * dt * i + 0.5 * (dt * i) ** 2 * acc[m_i]
                    p[npar * i + pm["obst"][j * 4 + 3]] = r
    return p


def test_packer_matches_reference_loops_point_robot():
    model, setup = build_model(os.path.join(CONFIG_DIR, "cfg2_pointRobotMpc.yaml"))
    cfg = MpcConfiguration(**setup["mpc"])
    props = {"nx": 6, "nu": 3, "ns": 0, "npar": model._npar, "m": 3}
    B = 3
    pk = ParamPacker(model._paramMap, props, cfg, batch=B)
    rng = np.random.default_rng(0)
    pos = rng.normal(size=(B, 2, 3)); rad = rng.uniform(0.2, 1, size=(B, 2))  # 2 given, third slot empty
    lim = np.array([[-10, -9, -8.0], [10, 9, 8]]); limu = np.array([[-1, -1, -15.0], [1, 1, 15]])
    goals = rng.normal(size=(B, 2))
    pk.setRadialConstraints(pos, rad, 0.3)
    pk.setJointLimits(lim); pk.setInputLimits(limu); pk.setGoalReaching(goals); pk.setConstraintAvoidance()
    for b in range(B):
        ref = _reference_style_pack(model._paramMap, model._npar, cfg.time_horizon, cfg, [
            ("radial", (pos[b], rad[b], 0.3)), ("joint", lim), ("input", limu), ("goal", goals[b]), ("avoid", None)])
        assert np.array_equal(pk.params[b], ref)


def test_packer_dynamic_obstacles_and_slack_weight():
    model, setup = build_model(os.path.join(CONFIG_DIR, "cfg3_boxerMpc.yaml"))
    cfg = MpcConfiguration(**setup["mpc"])
    props = {"nx": 8, "nu": 2, "ns": 1, "npar": model._npar, "m": 3}
    assert model._npar == 44 and "ws" in model._paramMap
    B = 2
    pk = ParamPacker(model._paramMap, props, cfg, batch=B)
    rng = np.random.default_rng(1)
    dyn = rng.normal(size=(B, 3 * 9))  # 3 of 5 obstacles given
    pk.setRadialConstraints(np.zeros((1, 0, 3)), np.zeros((1, 0)), 0.6)
    pk.updateDynamicObstacles(dyn)
    for b in range(B):
        ref = _reference_style_pack(model._paramMap, model._npar, cfg.time_horizon, cfg, [
            ("radial", ([], [], 0.6)), ("dyn", (dyn[b], cfg.time_step, 0.1))])
        np.testing.assert_allclose(pk.params[b], ref, rtol=0, atol=1e-15)
    assert np.all(pk.p3[:, :, model._paramMap["ws"][0]] == 1e10)


def test_warm_start_semantics():
    sc = make_scenario("cfg1", B=2)
    pk = sc.packer
    N, nv, nx = pk.N, pk.nvar, pk.nx
    xinit = np.arange(2 * nx, dtype=float).reshape(2, nx)
    pk.reset()
    x0 = pk.setX0(xinit, None, "current_state")
    assert np.array_equal(x0[:, :, :nx], np.repeat(xinit[:, None, :], N, axis=1)) and np.all(x0[:, :, nx:] == 0)
    # previous_plan: first call behaves like current_state, then shifts
    pk.reset()
    x0 = pk.setX0(xinit, None, "previous_plan").copy()
    assert np.array_equal(x0[:, :, :nx], np.repeat(xinit[:, None, :], N, axis=1))
    zprev = np.random.default_rng(0).normal(size=(2, N, nv))
    x0 = pk.setX0(xinit, zprev, "previous_plan")
    assert np.array_equal(x0[:, : N - 1], zprev[:, 1:]) and np.array_equal(x0[:, N - 1], zprev[:, N - 1])


def test_golden_fixtures_reproduce_from_oracle(oracle_lib):
    """The committed golden vectors are what the oracle computes today."""
    import json
    from oracle.oracle import Oracle
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    for name in ("cfg1", "cfg2", "cfg3", "cfg4", "boxer", "pointRobot", "panda"):
        g = np.load(os.path.join(gdir, name + ".npz"))
        desc = json.loads(str(g["desc"]))
        desc["lb"] = [float(v) for v in desc["lb"]]; desc["ub"] = [float(v) for v in desc["ub"]]
        o = Oracle(desc)
        r = o.solve_batch(g["xinit"], g["x0"], g["params"])
        assert np.array_equal(r["exitflag"], g["exitflag"])
        np.testing.assert_allclose(r["z"], g["z"], rtol=0, atol=1e-9)
        e = o.eval_stage(g["x0"][0, 0], g["params"][0].reshape(o.N, o.npar)[0])
        np.testing.assert_allclose(e["g"], g["stage_g"], atol=1e-13)
        np.testing.assert_allclose(e["H"], g["stage_H"], atol=1e-10)


def test_closed_loop_golden_trace_reproduces_from_oracle(oracle_lib):
    """tests/golden/cfg1_closed_loop.npz (50 control steps of the reference example scenario) is what the
    oracle-driven loop of tests/golden/make_golden.py computes today."""
    import importlib.util
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(gdir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = np.load(os.path.join(gdir, "cfg1_closed_loop.npz"))
    _, xs, us, flags = mg.closed_loop_trace()
    assert np.array_equal(flags, g["exitflag"])
    np.testing.assert_allclose(xs, g["xs"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(us, g["us"], rtol=0, atol=1e-9)
    assert xs[-1][0] > xs[0][0] + 2.0   # it does drive towards the goal


# ---------------------------------------------------------------------------------------------------
# tooling parity: makeSolver argv contract, ROS-style YAML, set_mpc_parameter dispatcher
# ---------------------------------------------------------------------------------------------------
def _load_make_solver():
    import importlib.util
    path = os.path.join(os.path.dirname(CONFIG_DIR), "makeSolver.py")
    spec = importlib.util.spec_from_file_location("makeSolver_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_make_solver_accepts_the_reference_argv_form(tmp_path):
    """reference examples/makeSolver.py:26-29: robot type = text between the last '/' and 'Mpc' (regex
    '\\/(\\S*)M'), file read = config/<type>Mpc.yaml"""
    ms = _load_make_solver()
    assert ms.setup_file_from_argv("config/pointRobotMpc.yaml") == "config/pointRobotMpc.yaml"
    assert ms.setup_file_from_argv("config/boxerMpc.yaml") == "config/boxerMpc.yaml"
    assert ms.setup_file_from_argv("x/pandaMpc.yaml") == "config/pandaMpc.yaml"   # as in the reference only the type counts
    absolute = os.path.join(CONFIG_DIR, "cfg2_pointRobotMpc.yaml")
    assert ms.setup_file_from_argv(absolute) == absolute


def test_ros_style_yaml_without_objectives_key(tmp_path):
    """ros_bridge/.../config/boxer_mpc_config.yaml:2-22 (from the fixture of the reference's files): no
    ``objectives`` key, no ``example`` block, obstacle weight under ``wobst``."""
    import json
    from robot_mpcs_amd.utils.utils import normalise_setup
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_pins.json")))
    ros = pins["configs"]["boxer_ros"]
    assert "objectives" not in ros["mpc"] and "example" not in ros
    with pytest.raises(TypeError):
        MpcConfiguration(**ros["mpc"])           # the strict schema still rejects it, like the reference
    norm = normalise_setup(ros)
    assert norm["mpc"]["objectives"] == ["GoalReaching", "ConstraintAvoidance"]
    assert norm["mpc"]["weights"]["wconstr"] == [0.5, 0.0, 0.0, 0.0] and norm["mpc"]["weights"]["ws"] == 1e10
    assert "objectives" not in ros["mpc"]        # input untouched
    yfile = tmp_path / "boxer_mpc_config.yaml"
    yfile.write_text(yaml.dump(ros))
    model, setup = build_model(str(yfile))
    target = model.generateSolver(location=str(tmp_path) + "/")
    assert os.path.basename(target) == "boxer_n3_01_H10_noSlack"
    assert model._paramMap["wconstr"] == [model._npar - 4 + i for i in range(4)] and model.number_inequalities == 1 + 0 + 6 + 4
    MpcConfiguration(**setup["mpc"])
    complete = normalise_setup(setup)
    assert complete["mpc"] == setup["mpc"]       # complete configs pass through unchanged


class _RecordingPlanner:
    def __init__(self):
        self.calls = []

    def concretize(self): self.calls.append(("concretize",))
    def reset(self): self.calls.append(("reset",))

    def __getattr__(self, name):
        if name.startswith("set"):
            return lambda *a: self.calls.append((name,) + a)
        raise AttributeError(name)


def test_set_mpc_parameter_dispatch_follows_the_reference(capsys):
    """reference examples/mpc_example.py:63-119: objectives first, then constraints, each in YAML order; the
    limits are handed over transposed; a missing scene attribute prints the reference's message and exits 1."""
    from robot_mpcs_amd.planner.harness import MpcHarness, SphereObstacle, StaticGoal
    rec = _RecordingPlanner()
    h = MpcHarness("config/wc_boxerMpc.yaml", planner=rec)
    assert h._robot_type == "wc_boxer" and rec.calls == [("concretize",), ("reset",)]
    h._goal = StaticGoal([7.2, -2.2])
    h._limits = np.array([[-10, 10], [-10, 10], [-10, 10.0]])
    h._limits_u = np.array([[-10, 10], [-10, 10.0]])
    h._limits_vel = np.array([[-1.2, 1.2], [-1.5, 1.5]])
    h._lin_constr = np.zeros((10, 2, 4)); h._r_body = 0.6
    h.set_mpc_parameter()
    names = [c[0] for c in rec.calls[2:]]
    assert names == ["setGoalReaching", "setConstraintAvoidance", "setLinearConstraints",
                     "setSelfCollisionAvoidanceConstraints", "setJointLimits", "setVelLimits", "setInputLimits"]
    assert rec.calls[2][1] == [7.2, -2.2]
    np.testing.assert_array_equal(rec.calls[6][1], h._limits.T)        # (2, n): limits[0] lower, limits[1] upper
    np.testing.assert_array_equal(rec.calls[7][1], h._limits_vel.T)
    # missing attribute
    h2 = MpcHarness("config/pointRobotMpc.yaml", planner=_RecordingPlanner())
    h2._goal = StaticGoal([8.2, -0.2]); h2._obstacles = [SphereObstacle([4.0, -0.5, 0.0], 1.0)]
    h2._limits = np.zeros((3, 2)); h2._limits_u = np.zeros((3, 2))   # _r_body is not set
    with pytest.raises(SystemExit) as ex:
        h2.set_mpc_parameter()
    assert ex.value.code == 1
    assert "The required attributes for setting RadialConstraints are not defined" in capsys.readouterr().out
    # unknown names
    h3 = MpcHarness("config/pointRobotMpc.yaml", planner=_RecordingPlanner())
    h3._config["mpc"]["objectives"] = ["NoSuchObjective"]
    with pytest.raises(SystemExit):
        h3.set_mpc_parameter()
    assert "No function to set the parameters for this objective is defined" in capsys.readouterr().out
