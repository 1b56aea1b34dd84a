"""Instantiates the YAML ``constraints`` list, in YAML order, and lets every plug-in register its
parameters (counterpart of reference ``InequalityManager.py:15-23``; the concatenation of the rows
and their softening, ``:25-33``, happen in the kernels)."""
from robot_mpcs_amd.models.mpcBase import ModelContext, ParamLayout


class InequalityManager:
    def __init__(self, ctx: ModelContext, layout: ParamLayout):
        import robot_mpcs_amd.models.inequalities as registry
        self.names = list(ctx.config.constraints)
        self.modules = []
        described = dict((ctx.plugins.get("constraints") or {}))
        for name in self.names:
            if not hasattr(registry, name) and name in described:
                # a user-defined plug-in given as a row description (plugins: constraints: <name>: ...)
                module = registry.DescribedRows(ctx, name, described[name])
            else:
                module = getattr(registry, name)(ctx)     # unknown name: AttributeError, as in the reference
            module.register(layout)
            self.modules.append(module)

    @property
    def number_inequalities(self) -> int:
        return sum(m._n_ineq for m in self.modules)
