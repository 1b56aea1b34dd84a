"""Numpy restatement of the reference's free-space decomposition (TEST INFRASTRUCTURE).

Follows ``robotmpcs/utils/free_space_decomposition.py:61-116`` line by line
(``HalfPlane.__init__`` ``:11-14``, ``compute_constraints`` ``:79-97``, ``asdict``
``:103-116``); the module itself cannot be imported here because of its ``urdfenvs``
import (``:4``), which the algorithm does not use.  One call = one seed position and
one point cloud; the GPU kernel does B x N of them at once.
"""
import numpy as np


def dot3(a, b):
    """np.dot of two 3-vectors with the summation order written out.  The reference calls np.dot
    (:14, :17), whose rounding for n = 3 depends on the BLAS build (plain or fused multiply-add);
    the restatement fixes ((a0 b0 + a1 b1) + a2 b2) without fusing so that it is reproducible
    anywhere; tests/test_fsd_oracle.py bounds the difference to np.dot at a few ulp."""
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def half_plane(point, position):
    """[normal(3), constant] of HalfPlane(point, position): normal = position - point,
    constant = -normal . point (reference :11-14, :53-54)."""
    normal = position - point
    return np.concatenate((normal, np.array([-dot3(normal, point)])))


def free_space_decomposition(points, position, number_constraints, max_radius):
    points = np.asarray(points, dtype=float)
    position = np.asarray(position, dtype=float)
    constraints = []
    dists = np.linalg.norm(points - position, axis=1)
    idx = np.argsort(dists, kind="stable")
    points = points[idx]
    points = points[dists[idx] < max_radius]
    while points.size > 0 and len(constraints) < number_constraints:
        point = points[0]
        c = half_plane(point, position)
        constraints.append(c)
        infront = np.array([not (dot3(c[:3], q) + c[3] <= 0) for q in points])   # point_infront_plane (:16-20)
        points = points[infront]
    out = np.zeros((number_constraints, 4))
    for i in range(number_constraints):
        if i < len(constraints):
            out[i] = constraints[i]
        else:
            out[i] = half_plane(position + np.array([20.0, 20.0, 0.0]), position)   # asdict dummy (:110-114)
    return out
