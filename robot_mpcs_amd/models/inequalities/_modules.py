"""Inequality plug-in modules.

Mirror of reference ``robotmpcs/models/inequalities/*.py``: one class per YAML
name, each with ``_n_ineq`` and ``set_parameters(ParamMap, npar)`` that appends
its entries to the parameter map in the reference order.  The symbolic
``eval_constraint`` of the reference is replaced by the hand-written HIP
device function selected by ``KIND`` (robot_mpcs_amd/csrc/rmpc_model.hpp); the
row order inside each module is documented there and in DESIGN.md.
"""
from robot_mpcs_amd.models.mpcBase import MpcBase

KIND_RADIAL = 0
KIND_LINEAR = 1
KIND_SELFCOLLISION = 2
KIND_JOINTLIMIT = 3
KIND_VELLIMIT = 4
KIND_INPUTLIMIT = 5


class RadialConstraints(MpcBase):
    """h = ||fk_l(q) - c_i|| - r_i - r_body, links outer / obstacles inner
    (reference ``RadialConstraints.py:6-23`` + ``mpcBase.py:82-101``; the
    undefined ``j`` at ``RadialConstraints.py:22`` is a reference bug)."""
    KIND = KIND_RADIAL

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._n_ineq = self._config.number_obstacles * len(self._robot_config.collision_links)

    def set_parameters(self, ParamMap, npar):
        self._paramMap = ParamMap
        self._npar = npar
        self.addEntry2ParamMap("r_body", 1)
        self.addEntry2ParamMap("obst", 4 * self._config.number_obstacles)
        return self._paramMap, self._npar


class LinearConstraints(MpcBase):
    """h = |a.fk_l(q) + d| / ||a|| - r_body for planes [a, d]
    (reference ``LinearConstraints.py:8-40``)."""
    KIND = KIND_LINEAR

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._n_ineq = self._config.number_obstacles * len(self._robot_config.collision_links)

    def set_parameters(self, ParamMap, npar):
        self._paramMap = ParamMap
        self._npar = npar
        self.addEntry2ParamMap("r_body", 1)
        for i in range(self._config.number_obstacles):
            self.addEntry2ParamMap("lin_constrs_" + str(i), 4)
        return self._paramMap, self._npar


class SelfCollisionAvoidanceConstraints(MpcBase):
    """h = ||fk_a(q) - fk_b(q)|| - 2 r_body per pair
    (reference ``SelfCollisionAvoidanceConstraints.py:8-27``)."""
    KIND = KIND_SELFCOLLISION

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._n_ineq = len(self._robot_config.selfCollision['pairs'])

    def set_parameters(self, ParamMap, npar):
        self._paramMap = ParamMap
        self._npar = npar
        self.addEntry2ParamMap("r_body", 1)
        return self._paramMap, self._npar


class JointLimitConstraints(MpcBase):
    """interleaved [q_j - lo_j, hi_j - q_j] (reference
    ``JointLimitConstraints.py:8-31``)."""
    KIND = KIND_JOINTLIMIT

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._n_ineq = self._n * 2

    def set_parameters(self, ParamMap, npar):
        self._paramMap = ParamMap
        self._npar = npar
        self.addEntry2ParamMap("lower_limits", self._n)
        self.addEntry2ParamMap("upper_limits", self._n)
        return self._paramMap, self._npar


class VelLimitConstraints(MpcBase):
    """interleaved lower/upper limits on the last two velocity states
    (reference ``VelLimitConstraints.py:8-31``; the reference declares
    ``_n_ineq = 2`` but returns 4 rows -- 4 is used)."""
    KIND = KIND_VELLIMIT

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._n_ineq = 4

    def set_parameters(self, ParamMap, npar):
        self._paramMap = ParamMap
        self._npar = npar
        self.addEntry2ParamMap("lower_limits_vel", 2)
        self.addEntry2ParamMap("upper_limits_vel", 2)
        return self._paramMap, self._npar


class InputLimitConstraints(MpcBase):
    """interleaved [u_j - lo_j, hi_j - u_j] (reference
    ``InputLimitConstraints.py:7-29``)."""
    KIND = KIND_INPUTLIMIT

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._n_ineq = self._nu * 2

    def set_parameters(self, ParamMap, npar):
        self._paramMap = ParamMap
        self._npar = npar
        self.addEntry2ParamMap("lower_limits_u", self._nu)
        self.addEntry2ParamMap("upper_limits_u", self._nu)
        return self._paramMap, self._npar
