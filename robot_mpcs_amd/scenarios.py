"""Synthetic batched MPC scenarios for the BASELINE.json configs.

Inputs follow SURVEY.md section 8(d): seeded ``numpy.random.default_rng``,
fp64, scenario constants taken from the reference examples
(``examples/pointRobot_example.py:31-65``, ``examples/boxer_example.py:46-93``,
``examples/panda_example.py:53-83``).  Produces the three arrays of the solver
seam (``xinit``, ``x0``, ``all_parameters``; reference ``mpcPlanner.py:246-250``)
with a leading batch axis, packed through the same setters the planner uses.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

from robot_mpcs_amd.models.diff_drive_mpc_model import MpcDiffDriveModel
from robot_mpcs_amd.models.mpcBase import MpcConfiguration
from robot_mpcs_amd.models.mpcModel import MpcModel, normalise_descriptor
from robot_mpcs_amd.planner.packing import ParamPacker
from robot_mpcs_amd.utils.urdf_chain import fk_positions
from robot_mpcs_amd.utils.utils import normalise_setup, parse_setup

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIG_DIR = os.path.join(_ROOT, "examples", "config")
ASSET_DIR = os.path.join(_ROOT, "examples", "assets")

CONFIG_FILES = {
    "cfg1": "cfg1_pointRobotMpc.yaml",
    "cfg2": "cfg2_pointRobotMpc.yaml",
    "cfg3": "cfg3_boxerMpc.yaml",
    "cfg4": "cfg4_pandaMpc.yaml",
    "pointRobot": "pointRobotMpc.yaml",
    "boxer": "boxerMpc.yaml",
    "panda": "pandaMpc.yaml",
    # test configs: VelLimitConstraints and a ConstraintAvoidance weight on every module
    "wc_point": "wc_pointRobotMpc.yaml",
    "wc_boxer": "wc_boxerMpc.yaml",
    "wc_boxer_slack": "wc_boxerSlackMpc.yaml",
    "wc_panda": "wc_pandaMpc.yaml",
    # test configs: kernel variants beyond the three shipped robots (a 2-joint gantry = the point robot's URDF cut
    # after its prismatic joints, 4 / 5 / 6-joint arms = the panda's URDF cut after joint 4 / 5 / 6)
    "chain2": "t_chain2Mpc.yaml",
    "chain5": "t_chain5Mpc.yaml",
    "chain4": "t_chain4Mpc.yaml",
    "chain6": "t_chain6Mpc.yaml",
    "chain8": "t_chain8Mpc.yaml",
    # constraint plug-ins given as row descriptions (top-level YAML block `plugins`, include/rmpc.h RMPC_MOD_ROWS)
    "plug_point": "plug_pointRobotMpc.yaml",
    "plug_panda": "plug_pandaMpc.yaml",
    "plug_boxer": "plug_boxerMpc.yaml",
}
DEFAULT_BATCH = {"cfg1": 1, "cfg2": 4096, "cfg3": 4096, "cfg4": 1024}


def build_model(config_file: str, asset_dir: str = ASSET_DIR, **mpc_overrides):
    """YAML -> model object with descriptor assembled (counterpart of
    reference ``examples/makeSolver.py:14-22``)."""
    setup = normalise_setup(parse_setup(config_file))
    setup["mpc"].update(mpc_overrides)
    robot_type = setup["mpc"]["model_name"]
    urdf = setup["robot"]["urdf_file"]
    if not os.path.isabs(urdf):
        setup["robot"]["urdf_file"] = os.path.join(asset_dir, robot_type, urdf)
    if setup["robot"]["base_type"] == "holonomic":
        model = MpcModel(initParamMap=True, **setup)
    elif setup["robot"]["base_type"] == "diffdrive":
        model = MpcDiffDriveModel(initParamMap=True, **setup)
    else:
        raise ValueError("unknown base_type")
    model.setModel()
    model.setCodeoptions()
    return model, setup


@dataclass
class Scenario:
    name: str
    model: MpcModel
    setup: dict
    desc: dict
    packer: ParamPacker
    xinit: np.ndarray
    x0: np.ndarray
    params: np.ndarray
    extra: dict = field(default_factory=dict)

    @property
    def B(self):
        return self.xinit.shape[0]


def _packer_for(model, setup, B):
    cfg = MpcConfiguration(**setup["mpc"])
    props = {"nx": model._nx, "nu": model._nu, "ns": model._ns, "npar": model._npar, "m": model._m}
    return ParamPacker(model._paramMap, props, cfg, batch=B)


POINT_LIMITS = np.array([[-10.0, -10.0, -10.0], [10.0, 10.0, 10.0]])
POINT_LIMITS_U = np.array([[-1.0, -1.0, -15.0], [1.0, 1.0, 15.0]])
BOXER_LIMITS = np.array([[-10.0, -10.0, -10.0], [10.0, 10.0, 10.0]])
BOXER_LIMITS_U = np.array([[-10.0, -10.0], [10.0, 10.0]])
BOXER_LIMITS_VEL = np.array([[-1.2, -1.5], [1.2, 1.5]])   # (v, omega); no reference example sets these (setVelLimits is never called there)
PANDA_LIMITS = np.array([
    [-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973],
    [2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973],
])
PANDA_LIMITS_U = np.array([
    [-1.0, -1.0, -15.0, -15.0, -7.5, -10.0, -12.5],
    [1.0, 1.0, 15.0, 15.0, 7.5, 10.0, 12.5],
])
# (the extra joint of the test asset panda_tool8.urdf, configuration chain8: position and input limits)
TOOL8_LIMITS = np.array([[-2.0, -12.5], [2.0, 12.5]])


def make_scenario(name: str, B: int | None = None, seed: int = 0, **mpc_overrides) -> Scenario:
    if name not in CONFIG_FILES:
        raise KeyError(name)
    B = int(B if B is not None else DEFAULT_BATCH.get(name, 1))
    model, setup = build_model(os.path.join(CONFIG_DIR, CONFIG_FILES[name]), **mpc_overrides)
    desc = normalise_descriptor(model._model)
    pk = _packer_for(model, setup, B)
    rng = np.random.default_rng(seed)
    nx = model._nx
    xinit = np.zeros((B, nx))
    extra = {}
    robot = setup["mpc"]["model_name"]
    nob = int(setup["mpc"]["number_obstacles"])
    constraints = setup["mpc"]["constraints"]

    if robot == "pointRobot" and (name == "cfg1" or nob == 1):
        # exactly the pointRobot_example.py scenario, replicated B times
        xinit[:, 3] = 0.1
        r_body = 0.3
        goal = np.array([8.2, -0.2, 0.0])
        pk.setRadialConstraints(np.array([[[4.0, -0.5, 0.0]]]), np.array([[1.0]]), r_body)
        pk.setJointLimits(POINT_LIMITS)
        pk.setInputLimits(POINT_LIMITS_U)
        pk.setGoalReaching(goal)
        extra.update(goal=np.tile(goal, (B, 1)), r_body=r_body)
    elif robot == "pointRobot":
        r_body = 0.3
        start = np.zeros((B, 2)); goal = np.zeros((B, 2))
        opos = np.zeros((B, nob, 3)); orad = np.zeros((B, nob))
        todo = np.ones(B, dtype=bool)
        while todo.any():
            k = int(todo.sum())
            s = rng.uniform(-8, 8, size=(k, 2)); g = rng.uniform(-8, 8, size=(k, 2))
            dvec = g - s
            dist = np.linalg.norm(dvec, axis=1)
            ok = dist >= 4.0
            dirn = dvec / np.maximum(dist, 1e-9)[:, None]
            nrm = np.stack([-dirn[:, 1], dirn[:, 0]], axis=1)
            along = rng.uniform(0.0, 1.0, size=(k, nob))
            lat = rng.uniform(-1.5, 1.5, size=(k, nob))
            c = s[:, None, :] + along[:, :, None] * dvec[:, None, :] + lat[:, :, None] * nrm[:, None, :]
            r = rng.uniform(0.3, 1.0, size=(k, nob))
            ds = np.linalg.norm(c - s[:, None, :], axis=2)
            dg = np.linalg.norm(c - g[:, None, :], axis=2)
            ok &= np.all(ds > r + r_body + 0.2, axis=1) & np.all(dg > r + r_body + 0.2, axis=1)
            idx = np.flatnonzero(todo)[ok]
            start[idx] = s[ok]; goal[idx] = g[ok]
            opos[idx, :, :2] = c[ok]; orad[idx] = r[ok]
            todo[idx] = False
        xinit[:, 0:2] = start
        if name == "plug_point":
            # the same numbers through the parameter entries of the described plug-ins (cfg2's rows, see the YAML)
            pk.setEntry("r_body", r_body)
            pk.setEntry("KeepOutSpheres", np.concatenate([opos, orad[:, :, None]], axis=2).reshape(B, 1, 4 * nob))
            pk.setEntry("AxisLimits_lower", POINT_LIMITS[0]); pk.setEntry("AxisLimits_upper", POINT_LIMITS[1])
            pk.setEntry("ThrustLimits_lower", POINT_LIMITS_U[0]); pk.setEntry("ThrustLimits_upper", POINT_LIMITS_U[1])
        else:
            pk.setRadialConstraints(opos, orad, r_body)
            pk.setJointLimits(POINT_LIMITS[:, :model._n])
            pk.setInputLimits(POINT_LIMITS_U[:, :model._n])
        g3 = np.concatenate([goal, np.zeros((B, 1))], axis=1)
        pk.setGoalReaching(g3)
        extra.update(goal=g3, r_body=r_body, obst_pos=opos, obst_radius=orad)
    elif robot == "boxer":
        r_body = 0.6
        xinit[:, 0:2] = rng.uniform(-6, 6, size=(B, 2))
        xinit[:, 2] = rng.uniform(-np.pi, np.pi, size=B)
        ang = rng.uniform(-np.pi, np.pi, size=B)
        rad = rng.uniform(3.0, 6.0, size=B)
        goal = np.clip(xinit[:, 0:2] + rad[:, None] * np.stack([np.cos(ang), np.sin(ang)], 1), -9, 9)
        g3 = np.concatenate([goal, np.zeros((B, 1))], axis=1)
        pk.setJointLimits(BOXER_LIMITS)
        pk.setInputLimits(BOXER_LIMITS_U)
        pk.setGoalReaching(g3)
        extra.update(goal=g3, r_body=r_body)
        if "RadialConstraints" in constraints:
            ee = xinit[:, 0:2] + 0.4 * np.stack([np.cos(xinit[:, 2]), np.sin(xinit[:, 2])], 1)
            opos = np.zeros((B, nob, 3)); ovel = np.zeros((B, nob, 3))
            todo = np.ones((B, nob), dtype=bool)
            while todo.any():
                cand = rng.uniform(-8, 8, size=(B, nob, 2))
                far = np.linalg.norm(cand - ee[:, None, :], axis=2) > (0.1 + r_body + 1.0)
                upd = todo & far
                opos[:, :, :2][upd] = cand[upd]
                todo &= ~far
            ovel[:, :, :2] = rng.uniform(-0.5, 0.5, size=(B, nob, 2))
            dyn = np.concatenate([opos, ovel, np.zeros((B, nob, 3))], axis=2).reshape(B, nob * 9)
            pk.setRadialConstraints(np.zeros((1, 0, 3)), np.zeros((1, 0)), r_body)
            pk.updateDynamicObstacles(dyn)
            extra.update(obst_dyn=dyn)
        if "VelLimitConstraints" in constraints:
            pk.setVelLimits(BOXER_LIMITS_VEL)
        if "LinearConstraints" in constraints and nob == 1:
            lin = np.tile(np.array([1.0, 0.0, 0.0, -100.0]), (pk.N, nob, 1))
            pk.setLinearConstraints(lin, r_body)
        elif "LinearConstraints" in constraints or name == "plug_boxer":
            # planes at 2.0 .. 3.5 from the lidar point of the start pose, random normals, the robot on the free side
            ee = xinit[:, 0:2] + 0.4 * np.stack([np.cos(xinit[:, 2]), np.sin(xinit[:, 2])], 1)
            phi = rng.uniform(-np.pi, np.pi, size=(B, nob))
            nrm = np.stack([np.cos(phi), np.sin(phi), np.zeros_like(phi)], axis=2) * rng.uniform(0.5, 2.0, size=(B, nob, 1))
            dist = rng.uniform(2.0, 3.5, size=(B, nob))
            dcoef = dist * np.linalg.norm(nrm, axis=2) - np.einsum("bij,bj->bi", nrm[:, :, :2], ee)
            lin = np.concatenate([nrm, dcoef[:, :, None]], axis=2)            # a.p + d = dist * |a| at the start
            if name == "plug_boxer":
                # (wc_boxer's numbers through the entries of the described plug-ins, see the YAML)
                pk.setEntry("r_body", r_body)
                pk.setEntry("Walls", lin.reshape(B, 1, 4 * nob))
                pk.setEntry("SpeedLimits_lower", BOXER_LIMITS_VEL[0]); pk.setEntry("SpeedLimits_upper", BOXER_LIMITS_VEL[1])
            else:
                pk.setLinearConstraints(np.broadcast_to(lin[:, None], (B, pk.N, nob, 4)), r_body)
            extra.update(lin_constrs=lin)
    elif robot == "panda":
        r_body = 0.14
        nj = model._n   # (7 for the panda; the test configuration chain5 cuts the chain after joint 5)
        lim8 = np.concatenate([PANDA_LIMITS, TOOL8_LIMITS[:, 0:1]], axis=1)
        limu8 = np.concatenate([PANDA_LIMITS_U, TOOL8_LIMITS[:, 1:2]], axis=1)
        lim, limu = lim8[:, :nj], limu8[:, :nj]
        q0 = np.median(lim, axis=0)[None, :] + (rng.uniform(-0.3, 0.3, size=(B, max(7, nj)))[:, :nj] if B > 1 else 0.0)
        q0 = np.clip(q0, lim[0] + 0.05, lim[1] - 0.05)
        xinit[:, 0:nj] = q0
        jit = (lambda s, sz: rng.uniform(-s, s, size=sz)) if B > 1 else (lambda s, sz: np.zeros(sz))
        goal = np.array([0.1, -0.6, 0.4]) + jit(0.15, (B, 3))
        opos = (np.array([0.5, -0.3, 0.3]) + jit(0.1, (B, 3)))[:, None, :]
        # reject obstacles that touch a collision link at the initial configuration
        fks = fk_positions(desc["joints"], q0, desc["link_frame"])
        for _ in range(100):
            clear = np.min([np.linalg.norm(fks[f] - opos[:, 0, :], axis=1) for f in desc["link_frame"]], axis=0)
            badm = clear < 0.1 + r_body + 0.05
            if not badm.any():
                break
            opos[badm, 0, :] = np.array([0.5, -0.3, 0.3]) + rng.uniform(-0.1, 0.1, size=(int(badm.sum()), 3))
        for _ in range(20):  # leftovers: push the obstacle away from its nearest link
            dists = np.stack([np.linalg.norm(fks[f] - opos[:, 0, :], axis=1) for f in desc["link_frame"]], axis=0)
            badm = dists.min(axis=0) < 0.1 + r_body + 0.05
            if not badm.any():
                break
            near = np.stack([fks[f] for f in desc["link_frame"]], axis=0)[dists.argmin(axis=0), np.arange(B)]
            dirn = opos[:, 0, :] - near
            dirn /= np.maximum(np.linalg.norm(dirn, axis=1, keepdims=True), 1e-9)
            opos[badm, 0, :] = near[badm] + dirn[badm] * (0.1 + r_body + 0.06)
        pk.setRadialConstraints(opos, np.full((B, 1), 0.1), r_body)
        pk.setSelfCollisionAvoidanceConstraints(r_body)
        pk.setJointLimits(lim)
        pk.setInputLimits(limu)
        pk.setGoalReaching(goal)
        extra.update(goal=goal, r_body=r_body, obst_pos=opos)
        if name == "plug_panda":
            # a keep-out sphere above the base for links 5 and 7 (clear of the start poses), speed limits of the wrist
            keep = np.array([-0.35, 0.25, 0.75]) + jit(0.05, (B, 3))
            pk.setEntry("KeepOut", np.concatenate([keep, np.full((B, 1), 0.08)], axis=1)[:, None, :])
            pk.setEntry("WristSpeed_lower", np.array([-0.05, -0.05]))
            pk.setEntry("WristSpeed_upper", np.array([0.05, 0.05]))
            extra.update(keep_out=keep)
    else:
        raise KeyError(robot)
    if "ConstraintAvoidance" in setup["mpc"]["objectives"]:
        pk.setConstraintAvoidance()
    x0 = pk.setX0(xinit, None, "current_state").copy()
    return Scenario(name=name, model=model, setup=setup, desc=desc, packer=pk, xinit=xinit, x0=x0,
                    params=pk.params.copy(), extra=extra)
