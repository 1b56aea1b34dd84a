"""Resolves the YAML ``constraints`` list to plug-in classes by name and
builds the parameter map in YAML order (reference
``robotmpcs/models/inequalities/InequalityManager.py:15-23``)."""
from robot_mpcs_amd.models.mpcBase import MpcBase


class InequalityManager(MpcBase):

    def __init__(self, ParamMap={}, npar=0, **kwargs):
        super().__init__(**kwargs)
        self._paramMap = ParamMap
        self._npar = npar
        self._kwargs = kwargs
        self.inequality_modules = []
        self.inequality_modules_strs = self._kwargs['mpc']['constraints']

    def set_constraints(self):
        import robot_mpcs_amd.models.inequalities as registry
        self.inequality_modules = []
        for class_name in self.inequality_modules_strs:
            class_ = getattr(registry, class_name)
            self.inequality_modules.append(class_(**self._kwargs))
            self._paramMap, self._npar = self.inequality_modules[-1].set_parameters(self._paramMap, self._npar)
        return self._paramMap, self._npar
