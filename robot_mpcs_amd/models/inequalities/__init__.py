from robot_mpcs_amd.models.inequalities._modules import (
    SelfCollisionAvoidanceConstraints,
    JointLimitConstraints,
    VelLimitConstraints,
    InputLimitConstraints,
    RadialConstraints,
    LinearConstraints,
    DescribedRows,
)
