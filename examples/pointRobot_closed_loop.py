#!/usr/bin/env python3
"""Headless closed loop of the point-robot example (scenario constants of the
reference's ``examples/pointRobot_example.py:31-65``): the plant is the same
ERK2 double integrator the MPC model uses, because the pybullet simulator of
the reference (urdfenvs) is not part of this repository.  Needs an MI355X.

    python makeSolver.py config/pointRobotMpc.yaml
    python pointRobot_closed_loop.py config/pointRobotMpc.yaml
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from robot_mpcs_amd.planner.mpcPlanner import MPCPlanner  # noqa: E402
from robot_mpcs_amd.utils.utils import parse_setup  # noqa: E402


class Sphere:
    def __init__(self, position, radius):
        self._p, self._r = position, radius

    def position(self): return self._p
    def radius(self): return self._r
    def dimension(self): return 3


def main(config_file):
    setup = parse_setup(os.path.join(HERE, config_file))
    planner = MPCPlanner("pointRobot", os.path.join(HERE, "solvers") + "/", None, False, **setup["mpc"])
    planner.concretize()
    planner.reset()
    planner.setGoalReaching([8.2, -0.2])
    planner.setConstraintAvoidance()
    planner.setJointLimits(np.array([[-10, -10, -10], [10, 10, 10.0]]))
    planner.setInputLimits(np.array([[-1, -1, -15], [1, 1, 15.0]]))
    planner.setRadialConstraints([Sphere([4.0, -0.5, 0.0], 1.0)], 0.3)
    planner.setSelfCollisionAvoidanceConstraints(0.3)
    dt = setup["mpc"]["time_step"]
    q, qdot = np.zeros(3), np.array([0.1, 0.0, 0.0])
    for step in range(400):
        action, output, exitflag = planner.computeAction(q, qdot)
        q = q + dt * qdot + 0.5 * dt * dt * action
        qdot = qdot + dt * action
        if step % 20 == 0:
            print(f"step {step:3d}  q = ({q[0]:6.3f}, {q[1]:6.3f})  |goal error| = {np.hypot(q[0]-8.2, q[1]+0.2):6.3f}  exitflag {exitflag}")
        if np.hypot(q[0] - 8.2, q[1] + 0.2) < 0.1:
            print("goal reached at step", step)
            break


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "config/pointRobotMpc.yaml")
