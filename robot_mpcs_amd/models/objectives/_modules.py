"""Objective plug-ins, table driven (reference ``robotmpcs/models/objectives``).

  GoalReaching         (fk_end(q) - goal)^T diag(wgoal) (fk_end(q) - goal)            goal_reaching.py:19-33
  ConstraintAvoidance  N * sum_i wconstr_i / h_i[0] over the first row of every non-empty inequality
                       module; the redundant ``for j in range(N)`` of the reference multiplies the term
                       by N -- kept                                                    constraint_avoidance.py:22-31

Each plug-in only contributes parameter entries here; the arithmetic is in csrc/rmpc_kernels.hip.
"""
from robot_mpcs_amd.models.mpcBase import ModelContext, ParamLayout

SPECS = {
    "GoalReaching": lambda c: [("goal", c.m), ("wgoal", c.m)],
    "ConstraintAvoidance": lambda c: [("wconstr", len(c.config.constraints))],
}


class ObjectiveModule:
    NAME = ""

    def __init__(self, ctx: ModelContext):
        self._params = SPECS[self.NAME](ctx)

    def register(self, layout: ParamLayout) -> None:
        for name, count in self._params:
            layout.add(name, count)


GoalReaching = type("GoalReaching", (ObjectiveModule,), {"NAME": "GoalReaching"})
ConstraintAvoidance = type("ConstraintAvoidance", (ObjectiveModule,), {"NAME": "ConstraintAvoidance"})
