"""Registers ``wu`` (and ``ws`` when the model has a slack variable), then the YAML ``objectives`` in
order (parameter order of reference ``ObjectiveManager.py:6-26``; the planner writes ``paramMap["ws"]``
when ``slack``, ``mpcPlanner.py:101-104``, so the entry must exist).  Stage cost = sum of the plug-ins
+ u^T diag(wu) u + ws s^2, terminal cost identical (``:28-46``)."""
from robot_mpcs_amd.models.mpcBase import ModelContext, ParamLayout


class ObjectiveManager:
    def __init__(self, ctx: ModelContext, layout: ParamLayout):
        import robot_mpcs_amd.models.objectives as registry
        layout.add("wu", ctx.nu)
        if ctx.ns > 0:
            layout.add("ws", 1)
        self.names = list(ctx.config.objectives)
        self.modules = []
        for name in self.names:
            module = getattr(registry, name)(ctx)
            module.register(layout)
            self.modules.append(module)
