from robot_mpcs_amd.models.objectives._modules import GoalReaching, ConstraintAvoidance
