"""Holonomic MPC model: parameter map, NLP descriptor and the solver directory.

Mirror of reference ``robotmpcs/models/mpcModel.py``.  Where the reference
assembles a ``forcespro.nlp.SymbolicModel`` from CasADi callbacks
(``setModel``, ``:74-108``) and calls the FORCES Pro code generator
(``generateSolver``, ``:128-141``), this class assembles a plain numeric
*model descriptor* -- dimensions, module list, parameter offsets, chain
constants, bounds, integrator options -- that the MI355X solver library
(``robot_mpcs_amd/csrc``) consumes, and writes it next to the two YAML files
of the reference's on-disk contract:

    <location><model>_n<n>_<dt sans '.'>_H<N>[_noSlack]/
        paramMap.yaml      name -> index list   (reference ``:132-133``)
        properties.yaml    nx, nu, npar, ns, m, constraints (``:134-136``)
        rmpc_model.yaml    descriptor (this project)

Double-integrator dynamics (``continuous_dynamics``, ``:65-69``), ERK2 with
5 nodes and Ts = dt (``setCodeoptions``, ``:110-126``) are restated in
csrc/rmpc_model.hpp.
"""
import math
import os
from shutil import rmtree

import numpy as np
import yaml

from robot_mpcs_amd.models.mpcBase import ModelContext, ParamLayout
from robot_mpcs_amd.models.inequalities.InequalityManager import InequalityManager
from robot_mpcs_amd.models.objectives.ObjectiveManager import ObjectiveManager
from robot_mpcs_amd.utils.urdf_chain import JOINT_FIXED

ROBOT_CHAIN = 0
ROBOT_DIFFDRIVE = 1

MAX_JOINTS = 8
MAX_LINKS = 8
MAX_PAIRS = 4
MAX_MODULES = 8
NV_MAX = 24

DESCRIPTOR_FILE = "rmpc_model.yaml"

# default solver options (tolerances follow the FORCES Pro NLP defaults in
# spirit: TolStat / TolEq / TolIneq / TolComp)
DEFAULT_OPTIONS = {
    "max_iter": 200,
    "tol_stat": 1e-6,
    "tol_eq": 1e-8,
    "tol_ineq": 1e-8,
    "tol_comp": 1e-6,
    "mu0": 1.0,
    "acc_iters": 8,       # acceptable termination window (0 disables)
    "acc_obj_tol": 1e-8,  # relative objective change counted as stagnation
    "ls_max": 25,         # step halvings allowed in one line search (exitflag -8 beyond)
}


class MpcModel:
    def __init__(self, initParamMap=True, **kwargs):
        self._ctx = ctx = ModelContext(kwargs)
        self._kwargs = kwargs
        # names the reference's users (and makeSolver / the planner) read off the model object
        self._config, self._robot_config, self._debug, self._fk = ctx.config, ctx.robot, ctx.debug, ctx.chain
        self._n, self._nx, self._nu, self._ns, self._m, self._N = ctx.n, ctx.nx, ctx.nu, ctx.ns, ctx.m, ctx.N
        self._dt = ctx.dt
        self._modelName = ctx.config.model_name
        self._layout = ParamLayout()
        if initParamMap:
            # default box of the reference (mpcModel.py:23-27): x, u in [-100, 100], s in [0, inf)
            self._limits = {
                "x": {"low": np.full(self._nx, -100.0), "high": np.full(self._nx, 100.0)},
                "u": {"low": np.full(self._nu, -100.0), "high": np.full(self._nu, 100.0)},
                "s": {"low": np.zeros(1), "high": np.full(1, np.inf)},
            }
        self._inequality_manager = InequalityManager(ctx, self._layout)
        self.number_inequalities = self._inequality_manager.number_inequalities
        self._objective_manager = ObjectiveManager(ctx, self._layout)
        self._options = dict(DEFAULT_OPTIONS)

    # paramMap.yaml content and its length, under the reference's attribute names
    @property
    def _paramMap(self):
        return self._layout.entries

    @property
    def _npar(self):
        return self._layout.size

    def setLimits(self, limits):
        self._limits = limits

    def setDt(self, dt):
        self._dt = dt

    def setSolverOptions(self, **options):
        """Override solver tolerances / iteration cap (no reference analogue:
        FORCES code options beyond ``setCodeoptions`` are not exposed there)."""
        for k, v in options.items():
            if k not in DEFAULT_OPTIONS:
                raise KeyError(f"unknown solver option {k}")
            self._options[k] = v

    def robot_kind(self):
        return ROBOT_CHAIN

    # ------------------------------------------------------------------
    def setModel(self):
        """Assemble the NLP descriptor (counterpart of reference ``setModel``,
        ``mpcModel.py:74-108``): objective == objectiveN, 0 <= h <= inf,
        E = [I 0], bounds in order x, s, u, xinitidx = 0..nx-1."""
        chain = self._fk
        if len(chain.joints) > MAX_JOINTS:
            raise ValueError("kinematic chain longer than %d joints" % MAX_JOINTS)
        if self.robot_kind() == ROBOT_DIFFDRIVE and any(j.type != JOINT_FIXED for j in chain.joints):
            raise NotImplementedError("diff-drive base with an actuated arm (fk.n() > 0) is not supported")
        modules = self._inequality_manager.modules
        if len(modules) > MAX_MODULES:
            raise ValueError("more than %d constraint modules" % MAX_MODULES)
        links = [chain.frame_of(l) for l in self._robot_config.collision_links]
        pairs = [[chain.frame_of(a), chain.frame_of(b)] for a, b in self._robot_config.selfCollision['pairs']]
        if len(links) > MAX_LINKS or len(pairs) > MAX_PAIRS:
            raise ValueError("too many collision links / self-collision pairs")
        nvar = self._nx + self._ns + self._nu
        if nvar > NV_MAX:
            raise ValueError("nvar %d exceeds %d" % (nvar, NV_MAX))
        if self._ns > 0:
            lb = np.concatenate((self._limits["x"]["low"], self._limits["s"]["low"], self._limits["u"]["low"]))
            ub = np.concatenate((self._limits["x"]["high"], self._limits["s"]["high"], self._limits["u"]["high"]))
        else:
            lb = np.concatenate((self._limits["x"]["low"], self._limits["u"]["low"]))
            ub = np.concatenate((self._limits["x"]["high"], self._limits["u"]["high"]))

        off = self._layout.offset

        objectives = self._objective_manager.names
        self._model = {
            "robot": self.robot_kind(),
            "N": int(self._N),
            "n": int(self._n), "nx": int(self._nx), "nu": int(self._nu), "ns": int(self._ns),
            "npar": int(self._npar),
            "nh": int(self.number_inequalities),
            "dt": float(self._dt),
            "module_kind": [int(m.KIND) for m in modules],
            "nobst": int(self._config.number_obstacles),
            "link_frame": [int(v) for v in links],
            "pair_frame": [[int(a), int(b)] for a, b in pairs],
            "end_frame": int(chain.frame_of(self._robot_config.end_link)),
            "joints": [
                {"type": int(j.type), "dof": int(j.dof), "xyz": [float(v) for v in j.xyz],
                 "rot": [float(v) for v in j.rot], "axis": [float(v) for v in j.axis]}
                for j in chain.joints
            ],
            "off_r_body": off("r_body"), "off_obst": off("obst"), "off_lin": off("lin_constrs_0"),
            "off_lower": off("lower_limits"), "off_upper": off("upper_limits"),
            "off_lower_u": off("lower_limits_u"), "off_upper_u": off("upper_limits_u"),
            "off_lower_vel": off("lower_limits_vel"), "off_upper_vel": off("upper_limits_vel"),
            "off_wu": off("wu"), "off_goal": off("goal"), "off_wgoal": off("wgoal"),
            "off_wconstr": off("wconstr"), "off_ws": off("ws"),
            "has_goal": int("GoalReaching" in objectives),
            "has_avoid": int("ConstraintAvoidance" in objectives),
            "lb": [float(v) if math.isfinite(v) else (".inf" if v > 0 else "-.inf") for v in lb],
            "ub": [float(v) if math.isfinite(v) else (".inf" if v > 0 else "-.inf") for v in ub],
            "erk_nodes": 5,
            "options": dict(self._options),
        }
        # rows of the plug-ins given as row descriptions (include/rmpc.h RMPC_MOD_ROWS)
        xrows = []
        for mi, m in enumerate(modules):
            if hasattr(m, "rows"):
                xrows += [[int(v) for v in r] for r in m.rows(self._layout, mi)]
        if xrows:
            self._model["xrows"] = xrows
        return self._model

    def setCodeoptions(self, **kwargs):
        """Solver name rule of reference ``mpcModel.py:110-116``; integrator
        ERK2 / Ts = dt / nodes = 5 (``:118-120``) is fixed in the kernels."""
        solverName = self._modelName + "_n" + str(self._n) + "_" + str(self._dt).replace('.', '') + "_H" + str(self._N)
        if not self._config.slack:
            solverName += "_noSlack"
        if solverName in kwargs:
            solverName = kwargs.get('solverName')
        self._solverName = solverName
        self._codeoptions = {"name": solverName, "integrator": "ERK2", "Ts": self._dt, "nodes": 5,
                             "printlevel": 2 if self._config.debug else 0}

    def generateSolver(self, location="./"):
        """Write the solver directory (reference ``mpcModel.py:128-141``).
        Nothing is generated remotely: the HIP library is built once by
        ``__graft_entry__.build()`` and is model-generic."""
        if self._debug:
            location += 'debug/'
        target = location + self._solverName
        if os.path.exists(target) and os.path.isdir(target):
            rmtree(target)
        os.makedirs(target)
        with open(target + '/paramMap.yaml', 'w') as outfile:
            yaml.dump(self._paramMap, outfile, default_flow_style=False)
        properties = {"nx": self._nx, "nu": self._nu, "npar": self._npar, "ns": self._ns, "m": self._m,
                      "constraints": self._inequality_manager.names}
        with open(target + '/properties.yaml', 'w') as outfile:
            yaml.dump(properties, outfile, default_flow_style=False)
        with open(target + '/' + DESCRIPTOR_FILE, 'w') as outfile:
            yaml.dump(self._model, outfile, default_flow_style=None)
        return target


def load_descriptor(path):
    """Read ``rmpc_model.yaml`` back, restoring infinite bounds."""
    with open(path, "r") as f:
        desc = yaml.safe_load(f)

    def fix(v):
        if isinstance(v, str):
            return float(v.replace(".inf", "inf"))
        return float(v)

    desc["lb"] = [fix(v) for v in desc["lb"]]
    desc["ub"] = [fix(v) for v in desc["ub"]]
    return desc


def normalise_descriptor(desc):
    """In-memory descriptor (from ``setModel``) -> same form as ``load_descriptor``."""
    out = dict(desc)

    def fix(v):
        if isinstance(v, str):
            return float(v.replace(".inf", "inf"))
        return float(v)

    out["lb"] = [fix(v) for v in desc["lb"]]
    out["ub"] = [fix(v) for v in desc["ub"]]
    return out
