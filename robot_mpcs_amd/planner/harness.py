"""Headless scenario harness: the part of the reference's ``MpcExample``
(``examples/mpc_example.py:35-119``) that does not need the simulator.

``MpcHarness(config_file)`` builds the model and the planner the way
``MpcExample.__init__`` does (``:35-61``; robot type from the file name, solver
directory ``<examples>/solvers/``, URDF under ``<examples>/assets/<type>/``), and
``set_mpc_parameter()`` is the reference's name -> setter dispatch (``:63-119``):
for every objective and constraint listed in the YAML the matching planner setter
is called with the scene attribute the reference reads (``_goal``, ``_limits``,
``_limits_u``, ``_limits_vel``, ``_lin_constr``, ``_obstacles``, ``_r_body``); a
missing attribute prints the reference's message and exits with status 1, an
unknown name likewise.  The dispatch is a table, not an if-chain.
"""
from __future__ import annotations

import os
import re
import sys

import numpy as np

from robot_mpcs_amd.utils.utils import normalise_setup, parse_setup


class SphereObstacle:
    """Minimal stand-in for ``mpscenes`` sphere obstacles (position / radius / dimension)."""

    def __init__(self, position, radius):
        self._p, self._r = list(position), float(radius)

    def position(self):
        return self._p

    def radius(self):
        return self._r

    def dimension(self):
        return len(self._p)


class StaticGoal:
    """Minimal stand-in for ``GoalComposition``: ``primary_goal().position()`` and ``epsilon()``."""

    def __init__(self, position, epsilon=0.1):
        self._p, self._eps = list(position), float(epsilon)

    def primary_goal(self):
        return self

    def position(self):
        return self._p

    def epsilon(self):
        return self._eps


# YAML name -> (planner setter name, arguments read off the harness, error raised when they are missing)
_OBJECTIVES = {
    'GoalReaching': ('setGoalReaching', lambda h: (h._goal.primary_goal().position(),), AttributeError,
                     'are not defined'),
    'ConstraintAvoidance': ('setConstraintAvoidance', lambda h: (), KeyError, 'are not defined in the config file'),
}
_CONSTRAINTS = {
    'JointLimitConstraints': ('setJointLimits', lambda h: (np.transpose(h._limits),)),
    'VelLimitConstraints': ('setVelLimits', lambda h: (np.transpose(h._limits_vel),)),
    'InputLimitConstraints': ('setInputLimits', lambda h: (np.transpose(h._limits_u),)),
    'LinearConstraints': ('setLinearConstraints', lambda h: (h._lin_constr, h._r_body)),
    'RadialConstraints': ('setRadialConstraints', lambda h: (h._obstacles, h._r_body)),
    'SelfCollisionAvoidanceConstraints': ('setSelfCollisionAvoidanceConstraints', lambda h: (h._r_body,)),
}


class MpcHarness(object):
    def __init__(self, config_file_name: str, examples_dir: str | None = None, planner=None):
        """``planner``: inject a planner object (tests); by default the solver directory must exist
        (``python makeSolver.py <config>``) and an MI355X must be present."""
        here = examples_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(
            os.path.abspath(__file__)))), "examples")
        match = re.search(r'/([^/]+)Mpc\.yaml', "/" + config_file_name)     # mpc_example.py:38
        if match is None:
            raise ValueError("config file name must look like <dir>/<robot>Mpc.yaml")
        self._robot_type = match.group(1)
        self._solver_directory = os.path.join(here, "solvers") + "/"
        path = config_file_name if os.path.isabs(config_file_name) else os.path.join(here, config_file_name)
        self._config = normalise_setup(parse_setup(path))
        urdf = self._config['robot']['urdf_file']
        if not os.path.isabs(urdf):
            robot_dir = self._config['mpc']['model_name']
            self._config['robot']['urdf_file'] = os.path.join(here, "assets", robot_dir, urdf)
        if planner is None:
            from robot_mpcs_amd.planner.mpcPlanner import MPCPlanner
            planner = MPCPlanner(self._config['mpc']['model_name'], self._solver_directory, None,
                                 self._config['example']['debug'], **self._config['mpc'])
        self._planner = planner
        self._planner.concretize()
        self._planner.reset()
        self._n = self._config['mpc']['n']

    def set_mpc_parameter(self):
        for objective in self._config['mpc']['objectives']:
            if objective not in _OBJECTIVES:
                print('No function to set the parameters for this objective is defined')
                sys.exit(1)
            setter, args, error, text = _OBJECTIVES[objective]
            try:
                getattr(self._planner, setter)(*args(self))
            except error:
                print('The required attributes for setting ' + objective + ' ' + text)
                sys.exit(1)
        for constraint in self._config['mpc']['constraints']:
            if constraint not in _CONSTRAINTS:
                print('No function to set the parameters for this constraint type is defined')
                sys.exit(1)
            setter, args = _CONSTRAINTS[constraint]
            try:
                getattr(self._planner, setter)(*args(self))
            except AttributeError:
                print('The required attributes for setting ' + constraint + ' are not defined')
                sys.exit(1)
