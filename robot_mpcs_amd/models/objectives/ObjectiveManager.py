"""Adds ``wu`` (and ``ws`` when ``slack``) then the YAML ``objectives`` in
order (reference ``robotmpcs/models/objectives/ObjectiveManager.py:6-26``).
Stage cost = sum of modules + u^T diag(wu) u + ws s^2, terminal cost identical
(``:28-46``)."""
from robot_mpcs_amd.models.mpcBase import MpcBase


class ObjectiveManager(MpcBase):
    def __init__(self, ParamMap={}, npar=0, ineq_modules=[], **kwargs):
        super().__init__(**kwargs)
        self._paramMap = ParamMap
        self._npar = npar
        self._kwargs = kwargs
        self.objective_modules = []
        self.objective_modules_strs = self._kwargs['mpc']['objectives']
        self.addEntry2ParamMap("wu", self._nu)
        if self._ns > 0:
            # the planner writes paramMap["ws"] when slack (reference mpcPlanner.py:101-104)
            self.addEntry2ParamMap("ws", 1)
        self._ineq_modules = ineq_modules

    def set_objectives(self):
        import robot_mpcs_amd.models.objectives as registry
        self.objective_modules = []
        for class_name in self.objective_modules_strs:
            class_ = getattr(registry, class_name)
            self.objective_modules.append(class_(self._ineq_modules, **self._kwargs))
            self._paramMap, self._npar = self.objective_modules[-1].set_parameters(self._paramMap, self._npar)
        return self._paramMap, self._npar
