"""MPC planner: drop-in for ``robotmpcs.planner.mpcPlanner.MPCPlanner``.

Same constructor, setters, ``solve(ob) -> (action, output, info, exitflag)``
and ``computeAction(*args) -> (action, output, exitflag)`` as the reference
(``robotmpcs/planner/mpcPlanner.py:32-301``).  The only native call of the
reference, ``forcespro.nlp.Solver.from_directory(dir)`` + ``.solve(problem)``
(``:73,262``), is replaced by the MI355X solver library through
``robot_mpcs_amd._lib.Solver`` (C ABI in ``include/rmpc.h``).

``BatchedMPCPlanner`` is the same planner with a leading batch axis ``B`` on
every input and output.

Reference bugs on this path are fixed with their intended semantics and marked
``FIX``.
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np
import yaml

from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.models.mpcBase import MpcConfiguration
from robot_mpcs_amd.models.mpcModel import DESCRIPTOR_FILE, load_descriptor
from robot_mpcs_amd.planner.packing import ParamPacker


class SolverDoesNotExistError(Exception):
    def __init__(self, solverName):
        super().__init__()
        self._solverName = solverName

    def __str__(self):
        return f"Solver with name {self._solverName} does not exist."


class EmptyObstacle():
    def position(self):
        return [-100, -100, -100]

    def radius(self):
        return -100

    def dim(self):
        return 3

    def dimension(self):  # FIX: the reference calls dimension() (mpcPlanner.py:129)
        return 3


class PlannerSettingIncomplete(Exception):
    pass


def solver_directory(solversDir, robotType, config: MpcConfiguration) -> str:
    """Directory name rule of reference ``mpcPlanner.py:43-54``."""
    dt_str = str(config.time_step).replace(".", "")
    name = solversDir + robotType + "_n" + str(config.n) + "_" + dt_str + "_H" + str(config.time_horizon)
    if not config.slack:
        name += "_noSlack"
    return name


def _stage_key(N: int, k: int) -> str:
    """1-based zero-padded stage keys of the FORCES output dict (reference :265-273)."""
    if N < 10:
        return "x%d" % k
    if N < 100:
        return "x%02d" % k
    return "x%03d" % k


class _PlannerCore(object):
    """Shared load-time contract: solver dir + two YAMLs + solver handle."""

    def _load(self, robotType, solversDir, mpc_model, debug, batch, device, kwargs):
        self._config = MpcConfiguration(**kwargs)
        self._robotType = robotType
        self._debug = debug
        self._solverFile = solver_directory(solversDir, robotType, self._config)
        if not os.path.isdir(self._solverFile):
            raise SolverDoesNotExistError(self._solverFile)
        with open(self._solverFile + "/paramMap.yaml", "r") as stream:
            self._paramMap = yaml.safe_load(stream)
        with open(self._solverFile + "/properties.yaml", "r") as stream:
            self._properties = yaml.safe_load(stream)
        self._nx = self._properties['nx']
        self._nu = self._properties['nu']
        self._ns = self._properties['ns']
        self._npar = self._properties['npar']
        desc_file = os.path.join(self._solverFile, DESCRIPTOR_FILE)
        if not os.path.isfile(desc_file):
            raise SolverDoesNotExistError(desc_file)
        self._descriptor = load_descriptor(desc_file)
        try:
            print("Loading solver %s" % self._solverFile)
            self._solver = Solver(self._descriptor, max_batch=batch, device=device)
        except Exception as e:
            print("FAILED TO LOAD SOLVER")
            raise e
        if self._debug:
            self._mpc_model = mpc_model
        self._packer = ParamPacker(self._paramMap, self._properties, self._config, batch=batch)
        self._actionCounter = self._config.interval

    def m(self):
        return self._properties['m']

    def dt(self):
        # FIX: referenced by the reference (mpcPlanner.py:158) but never defined there
        return self._config.time_step


class BatchedMPCPlanner(_PlannerCore):
    """B independent MPC instances of one model, solved in one call."""

    def __init__(self, robotType, solversDir, batch, mpc_model=None, debug=False, device=0, **kwargs):
        self._B = int(batch)
        self._load(robotType, solversDir, mpc_model, debug, self._B, device, kwargs)
        self._z = None
        self._exitflag = None
        self._action = None

    @property
    def packer(self) -> ParamPacker:
        return self._packer

    def reset(self):
        self._packer.reset()
        self._z = None

    def concretize(self):
        self._actionCounter = self._config.interval

    def solve(self, ob):
        """ob (B, nx [+ 9*k dynamic-obstacle state]) -> (action (B, nu), z (B, N, nvar), stats, exitflag (B,))"""
        ob = np.asarray(ob, dtype=np.float64).reshape(self._B, -1)
        xinit = ob[:, 0: self._nx]
        if ob.shape[1] > self._nx:
            self._packer.updateDynamicObstacles(ob[:, self._nx:])
        x0 = self._packer.setX0(xinit, self._z, self._config.initialization)
        res = self._solver.solve(xinit, x0, self._packer.params)
        self._z = res["z"]
        N, nu = self._config.time_horizon, self._nu
        if self._config.control_mode == "vel":
            action = self._z[:, 1, -nu - nu: -nu] if N > 1 else np.zeros((self._B, nu))
        elif self._config.control_mode == "acc":
            action = self._z[:, 0, -nu:]
        else:
            print("No valid control mode specified!")
            action = np.zeros((self._B, nu))
        return action.copy(), self._z, res, res["exitflag"]

    def computeAction(self, *args):
        ob = np.concatenate([np.asarray(a, dtype=np.float64).reshape(self._B, -1) for a in args[:3]], axis=1)
        if self._actionCounter >= self._config.interval:
            self._action, _, _, self._exitflag = self.solve(ob)
            self._actionCounter = 1
        else:
            self._actionCounter += 1
        return self._action, self._z, self._exitflag


class MPCPlanner(_PlannerCore):
    def __init__(self, robotType, solversDir, mpc_model=None, debug=False, **kwargs):
        self._initial_step = True
        self._load(robotType, solversDir, mpc_model, debug, 1, int(os.environ.get("RMPC_DEVICE", "0")), kwargs)
        self.output = None
        self._exitflag = 0
        self._action = np.zeros(self._nu)

    # -- state the reference keeps as attributes -------------------------------
    @property
    def _params(self):
        return self._packer.params[0]

    @property
    def _x0(self):
        return self._packer._x0[0]

    def reset(self):
        print("RESETTING PLANNER")
        self._packer.reset()
        self._xinit = np.zeros(self._nx)
        self._initial_step = True
        if self._config.slack:
            self._slack = 0.0
        self.output = None

    def dynamic(self):
        return False

    # -- setters (reference :120-210) ---------------------------------------------
    def setRadialConstraints(self, obsts, r_body):
        pos, rad = [], []
        for obst in list(obsts)[: self._config.number_obstacles]:
            p = list(np.asarray(obst.position(), dtype=float).ravel())
            pos.append((p + [0.0, 0.0, 0.0])[: self.m()])
            rad.append(float(obst.radius()))
        pos = np.asarray(pos, dtype=float).reshape(1, len(pos), self.m())
        rad = np.asarray(rad, dtype=float).reshape(1, len(rad))
        self._packer.setRadialConstraints(pos, rad, r_body)
        self._r = self._packer._r

    def setLinearConstraints(self, lin_constr, r_body):
        self._packer.setLinearConstraints(np.asarray(lin_constr, dtype=float), r_body)

    def updateDynamicObstacles(self, obstArray):
        self._packer.updateDynamicObstacles(np.asarray(obstArray, dtype=float).reshape(1, -1))

    def setSelfCollisionAvoidanceConstraints(self, r_body):
        self._packer.setSelfCollisionAvoidanceConstraints(r_body)

    def setJointLimits(self, limits):
        self._packer.setJointLimits(np.asarray(limits, dtype=float))

    def setVelLimits(self, limits_vel):
        self._packer.setVelLimits(np.asarray(limits_vel, dtype=float))

    def setInputLimits(self, limits_u):
        self._packer.setInputLimits(np.asarray(limits_u, dtype=float))

    def setGoalReaching(self, goal_position):
        self._packer.setGoalReaching(np.asarray(goal_position, dtype=float))

    def setConstraintAvoidance(self):
        self._packer.setConstraintAvoidance()

    def setEntry(self, name, value):
        """Parameter entry of a constraint plug-in given as a row description (YAML block ``plugins``; entries
        ``<name>``, ``<name>_lower`` / ``<name>_upper``): ``value`` is broadcast over the stages."""
        self._packer.setEntry(name, np.asarray(value, dtype=float))

    def concretize(self):
        self._actionCounter = self._config.interval

    # -- warm start (reference :215-236) ----------------------------------------------
    def shiftHorizon(self, output):
        N = self._config.time_horizon
        z_prev = np.stack([output[_stage_key(N, k + 1)] for k in range(N)])[None]
        self._packer.shiftHorizon(z_prev)

    def setX0(self, initialize_type="current_state", initial_step=True):
        z_prev = None
        if self.output is not None:
            N = self._config.time_horizon
            z_prev = np.stack([self.output[_stage_key(N, k + 1)] for k in range(N)])[None]
        self._packer._initial_step = bool(initial_step)
        self._packer.setX0(self._xinit[None, :], z_prev, initialize_type)
        self._initial_step = self._packer._initial_step

    # -- hot path (reference :240-301) -----------------------------------------------------
    def solve(self, ob):
        ob = np.asarray(ob, dtype=np.float64).ravel()
        self._xinit = ob[0: self._nx]
        if ob.size > self._nx:
            self.updateDynamicObstacles(ob[self._nx:])
        self.setX0(initialize_type=self._config.initialization, initial_step=self._initial_step)
        res = self._solver.solve(self._xinit[None, :], self._packer._x0, self._packer.params)
        N = self._config.time_horizon
        self.output = {_stage_key(N, k + 1): res["z"][0, k].copy() for k in range(N)}
        exitflag = int(res["exitflag"][0])
        info = {"it": int(res["iters"][0]), "res": float(res["kkt"][0]), "pobj": float(res["obj"][0])}
        if exitflag < 0:
            print(exitflag)
        key0, key1 = _stage_key(N, 1), _stage_key(N, 2)
        if self._config.control_mode == "vel":
            action = self.output[key1][-self._nu - self._nu: -self._nu]
        elif self._config.control_mode == "acc":
            action = self.output[key0][-self._nu:]
        else:
            print("No valid control mode specified!")
            action = np.zeros((self._nu))
        if self._config.slack:
            self._slack = self.output[key0][self._nx]
            if self._slack > 1e-3:
                print("slack : ", self._slack)
        return action, self.output, info, exitflag

    def computeAction(self, *args) -> Tuple[np.ndarray, dict, int]:
        ob = np.concatenate(args[:3])
        if self._actionCounter >= self._config.interval:
            self._action, _, _, self._exitflag = self.solve(ob)
            self._actionCounter = 1
        else:
            self._actionCounter += 1
        # FIX: output / exitflag are kept between solves (unbound in the reference, :299-301)
        return self._action, self.output, self._exitflag
