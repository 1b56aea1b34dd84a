"""Configuration schema and model dimensions of the MPC model.

The two dataclasses are the config SCHEMA of the reference
(``robotmpcs/models/mpcBase.py:7-31``): the YAML ``mpc`` / ``robot`` blocks are
splatted into them, so unknown or missing keys raise ``TypeError`` exactly as
there -- that is part of the drop-in contract and the field lists must match.

Everything else is this project's own: ``ModelContext`` reads the URDF chain
once (``utils.urdf_chain``; the reference re-creates a ``GenericURDFFk`` in every
plug-in module) and derives the dimensions n / nx / nu / ns by the rules of
``mpcBase.py:52-66``; ``ParamLayout`` is the append-only name -> index-list map
behind ``paramMap.yaml`` (``addEntry2ParamMap``, ``:68-71``: an existing name is
kept, not re-added).  ``ns`` follows ``mpc.slack`` -- the reference hard codes 0
(``:62``), which leaves ``slack: True`` broken.
"""
from dataclasses import dataclass
from functools import lru_cache
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


@lru_cache(maxsize=32)
def _chain_of(urdf_file: str, root_link: str, end_link: str) -> UrdfChain:
    with open(urdf_file, "r") as f:
        return parse_chain(f.read(), root_link, end_link)


class ParamLayout:
    """Ordered parameter map of one stage: name -> consecutive indices (``paramMap.yaml``)."""

    def __init__(self):
        self.entries: Dict[str, List[int]] = {}
        self.size = 0
        # where the sphere / plane list of the described plug-ins starts when no built-in module registered one
        # (inequalities/_modules.py DescribedRows: their entries are 4-aligned behind it)
        self.anchors: Dict[str, int] = {}

    def add(self, name: str, count: int) -> None:
        if name in self.entries:       # first registration wins (mpcBase.py:68-71)
            return
        self.entries[name] = list(range(self.size, self.size + int(count)))
        self.size += int(count)

    def offset(self, name: str) -> int:
        return self.entries[name][0] if name in self.entries else -1

    def pad_to(self, anchor: int, multiple: int) -> None:
        """Padding entries until ``size - anchor`` is a multiple of ``multiple`` (described sphere / plane rows address
        their parameters as entries of the obstacle / plane list: include/rmpc.h, ``xrow_poff``)."""
        while (self.size - anchor) % multiple:
            self.add("_pad%d" % self.size, 1)


class ModelContext:
    """What every plug-in needs to know about the model: config, chain, dimensions."""
    M_WORKSPACE = 3    # obstacle / goal dimension m (mpcBase.py:44)

    def __init__(self, setup: dict):
        self.config = MpcConfiguration(**setup['mpc'])
        self.debug = setup['example']['debug']
        self.robot = RobotConfiguration(**setup['robot'])
        # row descriptions of user-defined plug-ins: a top-level YAML block of this project (the reference reads the
        # keys it knows from the top-level dict and ignores the rest; its 'mpc' / 'robot' schemas stay untouched)
        self.plugins = dict(setup.get('plugins') or {})
        self.chain = _chain_of(self.robot.urdf_file, self.robot.root_link, self.robot.end_link)
        dof = self.chain.n()
        if self.robot.base_type == 'holonomic':          # mpcBase.py:52-55
            self.n, self.nx, self.nu = dof, 2 * dof, dof
        elif self.robot.base_type == 'diffdrive':        # mpcBase.py:56-60
            self.n = dof + 3
            self.nx, self.nu = 2 * self.n + 2, 2 + dof
        else:
            raise ValueError(f"unknown base_type {self.robot.base_type}")
        self.ns = 1 if self.config.slack else 0
        self.m = self.M_WORKSPACE
        self.N = int(self.config.time_horizon)
        self.dt = self.config.time_step
