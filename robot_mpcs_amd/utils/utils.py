"""Host-side helpers mirroring ``robotmpcs/utils/utils.py`` (reference
``utils.py:5-8`` parse_setup, ``:48-52`` point_to_plane).  The pybullet
debug-line drawing of the reference is out of scope (GUI)."""
import numpy as np
import yaml


def parse_setup(setup_file: str):
    with open(setup_file, "r") as setup_stream:
        setup = yaml.safe_load(setup_stream)
    return setup


def normalise_setup(setup: dict) -> dict:
    """Config variants of the reference that its own ``MpcConfiguration`` rejects.

    The ROS bridge ships ``ros_bridge/src/robotmpcs_ros/config/boxer_mpc_config.yaml:2-22`` without an
    ``objectives`` key and without an ``example`` block, with the obstacle weight under ``wobst`` (the
    vocabulary of the dead ``GoalMpcObjective``).  Intended semantics, implemented here for the tooling
    (``makeSolver``, ``MpcHarness``; ``MpcConfiguration`` itself stays strict): goal tracking + inverse-barrier
    obstacle avoidance on the first constraint module, i.e. ``objectives = [GoalReaching,
    ConstraintAvoidance]`` with ``wconstr = [wobst, 0, ...]``; ``example = {debug: False}``.  Numeric strings
    (PyYAML reads ``ws: 1e10`` as a string) become floats.  Returns a new dict; complete configs pass through."""
    out = {k: (dict(v) if isinstance(v, dict) else v) for k, v in setup.items()}
    mpc = out["mpc"]
    mpc["weights"] = dict(mpc.get("weights", {}))
    for k, v in list(mpc["weights"].items()):
        if isinstance(v, str):
            try:
                mpc["weights"][k] = float(v)
            except ValueError:
                pass
    if "objectives" not in mpc:
        mpc["objectives"] = ["GoalReaching", "ConstraintAvoidance"]
        if "wconstr" not in mpc["weights"]:
            nmod = len(mpc.get("constraints", []))
            mpc["weights"]["wconstr"] = [float(mpc["weights"].get("wobst", 0.0))] + [0.0] * max(0, nmod - 1)
    out.setdefault("example", {"debug": False})
    return out


def point_to_plane(point, plane) -> float:
    """|a.p + d| / ||a|| for plane [a(3), d] (reference ``utils.py:48-52``)."""
    point = np.asarray(point, dtype=float)
    plane = np.asarray(plane, dtype=float)
    return abs(float(np.dot(plane[0:3], point) + plane[3])) / float(np.linalg.norm(plane[0:3]))
