"""CPU checks of the free-space-decomposition restatement (oracle/fsd_numpy.py) on hand-computable
cases of the reference rule (free_space_decomposition.py:79-116)."""
import numpy as np

from oracle.fsd_numpy import free_space_decomposition, half_plane


def test_half_plane_definition():
    c = half_plane(np.array([2.0, 0.0, 0.0]), np.zeros(3))
    assert np.array_equal(c, [-2.0, 0.0, 0.0, 4.0])            # normal = position - point, constant = -n.p
    assert c[:3] @ np.zeros(3) + c[3] > 0                      # the seed is in front of its planes


def test_greedy_selection_and_dummy_planes():
    pos = np.zeros(3)
    pts = np.array([[2.0, 0, 0], [3.0, 0.1, 0], [0, -1.0, 0], [0.5, 4.0, 0], [9.0, 9.0, 0]])
    out = free_space_decomposition(pts, pos, 4, 5.0)
    # closest first: (0,-1,0); it does not hide (2,0,0); (2,0,0) hides (3,0.1,0); (0.5,4,0) remains; (9,9,0) is too far
    assert np.array_equal(out[0], half_plane(pts[2], pos))
    assert np.array_equal(out[1], half_plane(pts[0], pos))
    assert np.array_equal(out[2], half_plane(pts[3], pos))
    assert np.array_equal(out[3], half_plane(pos + np.array([20.0, 20.0, 0.0]), pos))   # dummy
    assert np.all(out[:, :3] @ pos + out[:, 3] > 0)


def test_against_np_dot_transcription():
    """Same rule written with np.dot / np.apply_along_axis as the reference does: identical plane
    selection, constants equal to a few ulp (np.dot's rounding is BLAS dependent)."""
    rng = np.random.default_rng(5)
    for _ in range(50):
        pts = np.concatenate([rng.uniform(-6, 6, size=(64, 2)), np.full((64, 1), 0.02)], axis=1)
        pos = np.array([*rng.uniform(-1, 1, size=2), 0.02])
        K = 6
        cons = []
        d = np.linalg.norm(pts - pos, axis=1)
        idx = np.argsort(d, kind="stable")
        rem = pts[idx][d[idx] < 5.0]
        while rem.size > 0 and len(cons) < K:
            n = pos - rem[0]
            c = -np.dot(n, rem[0])
            cons.append(np.concatenate((n, [c])))
            rem = rem[np.apply_along_axis(lambda q: not (np.dot(n, q) + c <= 0), 1, rem)]
        out = free_space_decomposition(pts, pos, K, 5.0)
        assert len(cons) >= 2
        np.testing.assert_allclose(out[: len(cons)], np.array(cons), rtol=1e-14, atol=1e-14)
