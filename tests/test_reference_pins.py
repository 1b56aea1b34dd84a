"""Pins the repo's re-typed inputs to the data the reference holds (CPU only).

``tests/golden/ref_pins.json`` is written in the build container by
``tests/golden/make_ref_pins.py`` from the reference's three YAML configs, three URDFs and the
scenario constants of its example programs (read as data, nothing imported).  The reference has no
golden vectors for the solve path, so parity of the SOLVER stays unpinned; what
can be pinned is every input of the model definition, and that is what this file does.
"""
import json
import math
import os

import numpy as np
import pytest
import yaml

from robot_mpcs_amd import scenarios
from robot_mpcs_amd.models.mpcModel import normalise_descriptor
from robot_mpcs_amd.utils.urdf_chain import parse_chain, rpy_to_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = json.load(open(os.path.join(HERE, "golden", "ref_pins.json")))
ROBOTS = ["pointRobot", "boxer", "panda"]


def _num(v):
    """PyYAML reads the reference's ``ws: 1e10`` as a string (YAML 1.1 floats need a dot)."""
    if isinstance(v, str):
        try:
            return float(v)
        except ValueError:
            return v
    if isinstance(v, dict):
        return {k: _num(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_num(x) for x in v]
    return v


def _urdf_from_pins(chain) -> str:
    """Skeleton URDF with exactly the joints of the reference URDF (names, types, origins, axes)."""
    out = ['<robot name="pinned">']
    out += ['<link name="%s"/>' % l for l in chain["links"]]
    for j in chain["joints"]:
        ax = '<axis xyz="%s"/>' % " ".join(repr(v) for v in j["axis"]) if j["axis"] is not None else ""
        out.append('<joint name="%s" type="%s"><parent link="%s"/><child link="%s"/>'
                   '<origin xyz="%s" rpy="%s"/>%s</joint>' % (
                       j["name"], j["type"], j["parent"], j["child"], " ".join(repr(v) for v in j["xyz"]),
                       " ".join(repr(v) for v in j["rpy"]), ax))
    out.append("</robot>")
    return "\n".join(out)


@pytest.mark.parametrize("robot", ROBOTS)
def test_shipped_yaml_equals_reference_yaml(robot):
    mine = yaml.safe_load(open(os.path.join(scenarios.CONFIG_DIR, robot + "Mpc.yaml")))
    ref = PINS["configs"][robot]
    assert _num(mine["mpc"]) == _num(ref["mpc"])
    assert mine["robot"] == ref["robot"]
    assert mine["example"] == ref["example"]


@pytest.mark.parametrize("robot", ROBOTS)
def test_skeleton_urdf_chain_equals_reference_urdf_chain(robot):
    """The repo ships kinematic skeletons of the reference URDFs: same chain root -> end link."""
    cfg = PINS["configs"][robot]["robot"]
    mine = parse_chain(open(os.path.join(scenarios.ASSET_DIR, robot, cfg["urdf_file"])).read(),
                       cfg["root_link"], cfg["end_link"])
    ref = parse_chain(_urdf_from_pins(PINS["chains"][robot]), cfg["root_link"], cfg["end_link"])
    assert [j.name for j in mine.joints] == [j.name for j in ref.joints]
    assert mine.n() == ref.n()
    for a, b in zip(mine.joints, ref.joints):
        assert (a.type, a.parent, a.child, a.dof) == (b.type, b.parent, b.child, b.dof)
        np.testing.assert_allclose(a.xyz, b.xyz, atol=0, rtol=0)
        np.testing.assert_allclose(a.rot, b.rot, atol=1e-15)
        np.testing.assert_allclose(a.axis, b.axis, atol=0, rtol=0)
    for link in cfg["collision_links"]:
        assert mine.frame_of(link) == ref.frame_of(link)


@pytest.mark.parametrize("robot", ROBOTS)
def test_descriptor_from_reference_files_equals_repo_descriptor(robot, tmp_path):
    """YAML + URDF data of the reference through the repo's loader: identical paramMap and
    descriptor (what rmpc_create consumes) as from the repo's own copies."""
    cfg = _num(PINS["configs"][robot])
    assets = tmp_path / "assets" / robot
    assets.mkdir(parents=True)
    (assets / cfg["robot"]["urdf_file"]).write_text(_urdf_from_pins(PINS["chains"][robot]))
    yfile = tmp_path / (robot + "Mpc.yaml")
    yfile.write_text(yaml.dump(cfg))
    ref_model, _ = scenarios.build_model(str(yfile), asset_dir=str(tmp_path / "assets"))
    my_model, _ = scenarios.build_model(os.path.join(scenarios.CONFIG_DIR, robot + "Mpc.yaml"))
    assert ref_model._paramMap == my_model._paramMap and ref_model._npar == my_model._npar
    assert normalise_descriptor(ref_model._model) == normalise_descriptor(my_model._model)


def test_scenario_constants_equal_reference_examples():
    s = PINS["scenes"]
    np.testing.assert_array_equal(scenarios.POINT_LIMITS, np.array(s["pointRobot"]["limits"], float).T)
    np.testing.assert_array_equal(scenarios.POINT_LIMITS_U, np.array(s["pointRobot"]["limits_u"], float).T)
    np.testing.assert_array_equal(scenarios.BOXER_LIMITS, np.array(s["boxer"]["limits"], float).T)
    np.testing.assert_array_equal(scenarios.BOXER_LIMITS_U, np.array(s["boxer"]["limits_u"], float).T)
    np.testing.assert_array_equal(scenarios.PANDA_LIMITS, np.array(s["panda"]["limits"], float).T)
    np.testing.assert_array_equal(scenarios.PANDA_LIMITS_U, np.array(s["panda"]["limits_u"], float).T)


def test_point_robot_example_scenario_is_the_reference_scenario():
    """cfg1 / pointRobot: start, goal, obstacle, radii, limits of pointRobot_example.py:31-65, as packed
    parameters (paramMap offsets from the reference's own YAML via the loader)."""
    s = PINS["scenes"]["pointRobot"]
    for name in ("cfg1", "pointRobot"):
        sc = scenarios.make_scenario(name, B=1)
        pm, p = sc.model._paramMap, sc.params[0].reshape(sc.desc["N"], -1)
        lim = np.array(s["limits"], float)
        np.testing.assert_array_equal(sc.xinit[0, :3], np.median(lim, axis=1))   # pos0 = median(limits)
        np.testing.assert_array_equal(sc.xinit[0, 3:], s["vel0"])
        for k in range(sc.desc["N"]):
            assert p[k, pm["r_body"][0]] == s["r_body"]
            np.testing.assert_array_equal(p[k, pm["obst"]], s["obstacles"][0]["position"] + [s["obstacles"][0]["radius"]])
            np.testing.assert_array_equal(p[k, pm["goal"]], s["goal"] + [0.0])
            np.testing.assert_array_equal(p[k, pm["lower_limits"]], lim[:, 0])
            np.testing.assert_array_equal(p[k, pm["upper_limits"]], lim[:, 1])
            np.testing.assert_array_equal(p[k, pm["lower_limits_u"]], np.array(s["limits_u"], float)[:, 0])
            np.testing.assert_array_equal(p[k, pm["upper_limits_u"]], np.array(s["limits_u"], float)[:, 1])
            np.testing.assert_array_equal(p[k, pm["wconstr"]], PINS["configs"]["pointRobot"]["mpc"]["weights"]["wconstr"])


def test_panda_example_scenario_is_the_reference_scenario():
    s = PINS["scenes"]["panda"]
    sc = scenarios.make_scenario("panda", B=1)
    pm, p = sc.model._paramMap, sc.params[0].reshape(sc.desc["N"], -1)
    lim = np.array(s["limits"], float)
    np.testing.assert_allclose(sc.xinit[0, :7], np.median(lim, axis=1), atol=1e-15)
    assert p[0, pm["r_body"][0]] == s["r_body"]
    np.testing.assert_array_equal(p[0, pm["obst"]], s["obstacles"][0]["position"] + [s["obstacles"][0]["radius"]])
    np.testing.assert_array_equal(p[0, pm["goal"]], s["goal"])
    np.testing.assert_array_equal(p[0, pm["lower_limits"]], lim[:, 0])
    np.testing.assert_array_equal(p[0, pm["upper_limits_u"]], np.array(s["limits_u"], float)[:, 1])


def test_panda_fk_from_reference_urdf_constants_against_sympy(oracle_lib):
    """Symbolic chain product built from the joint origins of the REFERENCE panda.urdf (fixture), not from
    typed-in constants, vs. the C oracle's kinematics on the repo's descriptor."""
    sp = pytest.importorskip("sympy")
    from oracle.oracle import Oracle
    cfg = PINS["configs"]["panda"]["robot"]
    chain = parse_chain(_urdf_from_pins(PINS["chains"]["panda"]), cfg["root_link"], cfg["end_link"])
    q = sp.symbols("q0:7")
    T = sp.eye(4)
    frames = []
    for j in chain.joints:
        R = sp.Matrix(3, 3, [sp.nsimplify(v, tolerance=1e-15, rational=False) if abs(v) in (0.0, 1.0) else v
                             for v in j.rot])
        H = sp.eye(4)
        H[:3, :3] = R
        H[:3, 3] = sp.Matrix(j.xyz)
        assert j.axis == [0.0, 0.0, 1.0]
        Rz = sp.Matrix([[sp.cos(q[j.dof]), -sp.sin(q[j.dof]), 0, 0], [sp.sin(q[j.dof]), sp.cos(q[j.dof]), 0, 0],
                        [0, 0, 1, 0], [0, 0, 0, 1]])
        T = T * H * Rz
        frames.append(T[:3, 3])
    sc = scenarios.make_scenario("cfg4", B=1)
    o = Oracle(sc.desc)
    rng = np.random.default_rng(5)
    for link in cfg["collision_links"]:
        fr = chain.frame_of(link)
        pos_fn = sp.lambdify(q, frames[fr], "numpy")
        jac_fn = sp.lambdify(q, frames[fr].jacobian(sp.Matrix(q)), "numpy")
        for _ in range(3):
            qv = rng.uniform(-1.5, 1.5, 7)
            pos, J = o.fk(qv, fr)
            np.testing.assert_allclose(pos, np.asarray(pos_fn(*qv), dtype=float).ravel(), atol=1e-12)
            np.testing.assert_allclose(J, np.asarray(jac_fn(*qv), dtype=float), atol=1e-12)


def test_rpy_convention():
    # URDF fixed-axis rpy: R = Rz(y) Ry(p) Rx(r); a pure roll of +pi/2 maps y -> z
    R = np.array(rpy_to_matrix([math.pi / 2, 0, 0])).reshape(3, 3)
    np.testing.assert_allclose(R @ np.array([0, 1, 0]), [0, 0, 1], atol=1e-15)
