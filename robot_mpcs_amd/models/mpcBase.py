"""Configuration dataclasses and dimension rules of the MPC model.

Mirror of reference ``robotmpcs/models/mpcBase.py``:
``MpcConfiguration`` / ``RobotConfiguration`` (``:7-31``, identical field
names, so unknown or missing YAML keys raise ``TypeError`` exactly as there),
dimension rules n / nx / nu (``:52-60``), ``addEntry2ParamMap`` (``:68-71``)
and the z layout ``[x(nx); s(ns); u(nu)]`` (``:76-80``).

Differences, both recorded in DESIGN.md: no CasADi / forwardkinematics objects
are created (the chain constants are read from the URDF by
``utils.urdf_chain``), and ``ns`` follows ``mpc.slack`` -- the reference hard
codes ``_ns = 0`` (``:62``) which leaves ``slack: True`` broken (SURVEY.md 8a
row A12); the intended design is implemented.
"""
from dataclasses import dataclass
from typing import Dict, List

from robot_mpcs_amd.utils.urdf_chain import UrdfChain, parse_chain


@dataclass
class MpcConfiguration:
    time_horizon: int
    time_step: float
    weights: dict
    slack: bool
    interval: int
    constraints: list
    objectives: list
    number_obstacles: int
    model_name: str
    initialization: str
    n: int
    control_mode: str
    name: str = 'mpc'
    debug: bool = False


@dataclass
class RobotConfiguration:
    collision_links: list
    selfCollision: dict
    urdf_file: str
    root_link: str
    end_link: str
    base_type: str


class MpcBase(object):
    _npar: int
    _N: int
    _pairs: List[int]
    _paramMap: Dict[str, List[int]]
    _modelName: str

    def __init__(self, **kwargs):
        self._config = MpcConfiguration(**kwargs['mpc'])
        self._debug = kwargs['example']['debug']
        self._robot_config = RobotConfiguration(**kwargs['robot'])
        with open(self._robot_config.urdf_file, 'r') as f:
            urdf = f.read()
        self._modelName = self._config.model_name
        self._fk: UrdfChain = parse_chain(
            urdf, self._robot_config.root_link, self._robot_config.end_link
        )
        self._m = 3
        self._dt = self._config.time_step
        if self._robot_config.base_type == 'holonomic':
            self._n = self._fk.n()
            self._nx = 2 * self._n
            self._nu = self._n
        elif self._robot_config.base_type == 'diffdrive':
            self._n = self._fk.n() + 3
            self._nx = 2 * self._n + 2
            self._nu = 2 + self._fk.n()
        else:
            raise ValueError(f"unknown base_type {self._robot_config.base_type}")
        self._ns = 1 if self._config.slack else 0
        self._n_obst = 0
        self._m_obst = 3
        self._pairs = []
        self._N = self._config.time_horizon

    def addEntry2ParamMap(self, name, n_par):
        if name not in self._paramMap:
            self._paramMap[name] = list(range(self._npar, self._npar + n_par))
            self._npar += n_par

    def get_velocity(self, z):
        return z[self._n: self._nx]

    def extractVariables(self, z):
        q = z[0: self._n]
        qdot = z[self._n: self._nx]
        qddot = z[self._nx + self._ns: self._nx + self._ns + self._nu]
        return q, qdot, qddot
