#!/usr/bin/env python3
"""Write the solver directory for a config (counterpart of the reference's
``examples/makeSolver.py``).

    python makeSolver.py config/pointRobotMpc.yaml

produces ``examples/solvers/<model>_n<n>_<dt>_H<N>[_noSlack]/`` with
``paramMap.yaml``, ``properties.yaml`` (the reference's on-disk contract,
``mpcModel.py:132-136``) and ``rmpc_model.yaml`` (the numeric descriptor the
MI355X solver library loads).  No code generation and no licence server are
involved: the HIP library is model-generic and built once by
``__graft_entry__.build()``.
"""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from robot_mpcs_amd.scenarios import build_model  # noqa: E402


def main(setup_file: str):
    if not os.path.isabs(setup_file):
        setup_file = os.path.join(HERE, setup_file)
    model, _ = build_model(setup_file, asset_dir=os.path.join(HERE, "assets"))
    path_to_solvers = os.path.join(HERE, "solvers") + "/"
    target = model.generateSolver(location=path_to_solvers)
    print("solver directory:", target)
    return target


def setup_file_from_argv(arg: str) -> str:
    """The reference's argv contract (``examples/makeSolver.py:26-29``): the robot type is whatever stands
    between the last '/' and 'Mpc' -- ``re.findall('\\/(\\S*)M', arg)[0]`` -- and the file read is
    ``config/<type>Mpc.yaml``.  ``python makeSolver.py config/pointRobotMpc.yaml`` therefore behaves as there;
    any other existing path (absolute, or a config outside ``config/``) is taken as is."""
    found = re.findall(r'\/(\S*)M', arg)
    if found:
        candidate = 'config/' + str(found[0]) + "Mpc.yaml"
        if os.path.isfile(os.path.join(HERE, candidate)) and not os.path.isabs(arg):
            return candidate
    return arg


if __name__ == "__main__":
    if len(sys.argv) < 2:
        print("Please provide a config file for solver generation.")
        sys.exit(1)
    main(setup_file_from_argv(sys.argv[1]))
