"""Inequality plug-ins, table driven.

The reference resolves the YAML ``constraints`` list to classes by name
(``getattr(robotmpcs.models.inequalities, name)``, ``InequalityManager.py:17-21``)
and asks each for its row count and its ``paramMap`` entries.  Here every plug-in
is ONE row of ``SPECS``: the HIP kernel kind (``rmpc.h`` ``RMPC_MOD_*``; the
arithmetic lives in ``csrc/rmpc_kernels.hip``), the number of rows and the
parameter entries in registration order.  A class per YAML name is generated from
the table so that the lookup by name -- and its ``AttributeError`` for unknown
names -- stays what users of the reference expect.

Row semantics (reference file:line):
  RadialConstraints                 h = ||fk_l(q) - c_i|| - r_i - r_body, links outer, obstacles inner
                                    (RadialConstraints.py:6-23, mpcBase.py:82-101; FIX undefined ``j`` at :22)
  LinearConstraints                 h = |a.fk_l(q) + d| / ||a|| - r_body        (LinearConstraints.py:8-40)
  SelfCollisionAvoidanceConstraints h = ||fk_a(q) - fk_b(q)|| - 2 r_body        (SelfCollision...py:8-27)
  JointLimitConstraints             [q_j - lo_j, hi_j - q_j] interleaved        (JointLimitConstraints.py:8-31)
  VelLimitConstraints               the same on the last two velocity states; FIX: 4 rows, the reference
                                    declares ``_n_ineq = 2`` but returns 4      (VelLimitConstraints.py:8-31)
  InputLimitConstraints             [u_j - lo_j, hi_j - u_j] interleaved        (InputLimitConstraints.py:7-29)
"""
from typing import Callable, List, NamedTuple, Tuple

from robot_mpcs_amd.models.mpcBase import ModelContext, ParamLayout


class Spec(NamedTuple):
    kind: int                                                  # RMPC_MOD_* of include/rmpc.h
    rows: Callable[[ModelContext], int]
    params: Callable[[ModelContext], List[Tuple[str, int]]]     # (name, count) in registration order


def _fk_rows(c: ModelContext) -> int:
    return c.config.number_obstacles * len(c.robot.collision_links)


SPECS = {
    "RadialConstraints": Spec(0, _fk_rows, lambda c: [("r_body", 1), ("obst", 4 * c.config.number_obstacles)]),
    "LinearConstraints": Spec(1, _fk_rows, lambda c: [("r_body", 1)] + [("lin_constrs_%d" % i, 4) for i in
                                                                        range(c.config.number_obstacles)]),
    "SelfCollisionAvoidanceConstraints": Spec(2, lambda c: len(c.robot.selfCollision['pairs']),
                                              lambda c: [("r_body", 1)]),
    "JointLimitConstraints": Spec(3, lambda c: 2 * c.n, lambda c: [("lower_limits", c.n), ("upper_limits", c.n)]),
    "VelLimitConstraints": Spec(4, lambda c: 4, lambda c: [("lower_limits_vel", 2), ("upper_limits_vel", 2)]),
    "InputLimitConstraints": Spec(5, lambda c: 2 * c.nu,
                                  lambda c: [("lower_limits_u", c.nu), ("upper_limits_u", c.nu)]),
}


class InequalityModule:
    """One constraint plug-in bound to a model context."""
    NAME = ""

    def __init__(self, ctx: ModelContext):
        self.spec = SPECS[self.NAME]
        self.KIND = self.spec.kind
        self._n_ineq = int(self.spec.rows(ctx))
        self._params = self.spec.params(ctx)

    def register(self, layout: ParamLayout) -> None:
        for name, count in self._params:
            layout.add(name, count)


def _make(name):
    return type(name, (InequalityModule,), {"NAME": name, "__doc__": "constraint plug-in '%s' (see SPECS)" % name})


RadialConstraints = _make("RadialConstraints")
LinearConstraints = _make("LinearConstraints")
SelfCollisionAvoidanceConstraints = _make("SelfCollisionAvoidanceConstraints")
JointLimitConstraints = _make("JointLimitConstraints")
VelLimitConstraints = _make("VelLimitConstraints")
InputLimitConstraints = _make("InputLimitConstraints")
