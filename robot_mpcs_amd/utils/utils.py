"""Host-side helpers mirroring ``robotmpcs/utils/utils.py`` (reference
``utils.py:5-8`` parse_setup, ``:48-52`` point_to_plane).  The pybullet
debug-line drawing of the reference is out of scope (GUI)."""
import numpy as np
import yaml


def parse_setup(setup_file: str):
    with open(setup_file, "r") as setup_stream:
        setup = yaml.safe_load(setup_stream)
    return setup


def point_to_plane(point, plane) -> float:
    """|a.p + d| / ||a|| for plane [a(3), d] (reference ``utils.py:48-52``)."""
    point = np.asarray(point, dtype=float)
    plane = np.asarray(plane, dtype=float)
    return abs(float(np.dot(plane[0:3], point) + plane[3])) / float(np.linalg.norm(plane[0:3]))
