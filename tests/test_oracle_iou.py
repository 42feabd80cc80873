"""The rotated-IoU oracle (oracle/warp_oracle.c, a float64 Sutherland-Hodgman clip in world coordinates) against 50-digit arithmetic
(tests/exact_iou.py): how far the yardstick itself can be trusted.  The arithmetic of the reference's call (d3d.box.box2d_iou,
rbox_tracker.py:87-92) is not under /root/reference: parity unpinned; this pins the oracle to the GEOMETRIC quantity instead."""
import numpy as np
import pytest

from oracle import cpu_oracle as co

mp = pytest.importorskip("mpmath")
from tests.exact_iou import iou as exact  # noqa: E402


def _boxes(rng, n, span, sizes=((1.6, 2.2), (3.5, 6.0))):
    return np.column_stack([rng.uniform(0, span, (n, 2)), rng.uniform(*sizes[0], n), rng.uniform(*sizes[1], n), rng.uniform(-np.pi, np.pi, n)])


def test_oracle_matches_exact_arithmetic_for_vehicle_sized_boxes():
    """Config 5's boxes (vehicles in a 100 m square, and a dense 12 m square where most pairs overlap): the oracle is exact to 1e-13."""
    rng = np.random.default_rng(3)
    for span in (100.0, 12.0):
        a, b = _boxes(rng, 40, span), _boxes(rng, 40, span)
        got = co.rbox_iou(a, b)
        ii, jj = np.nonzero(got > 0)
        assert len(ii) > (5 if span > 50 else 200)
        for i, j in list(zip(ii, jj))[:300]:
            assert abs(got[i, j] - exact(a[i], b[j])) <= 1e-13, (i, j)
        for i, j in zip(*np.nonzero(got == 0)):
            if (i + j) % 37 == 0:
                assert exact(a[i], b[j]) == 0.0


def test_oracle_error_grows_with_distance_over_size():
    """Centimetre boxes twenty metres from the origin: the world-coordinate clip loses ulp(20) / 0.01 -- the reason the GPU tests compare the
    kernel (which works in the clip box's own frame) with the oracle at 2e-10 for such pairs and at 1e-12 otherwise."""
    rng = np.random.default_rng(77)
    n = 60
    a = np.column_stack([rng.uniform(15, 20, (n, 2)), 10 ** rng.uniform(-2, -1.5, n), 10 ** rng.uniform(-2, -1.5, n), rng.uniform(-7, 7, n)])
    b = a + np.array([1e-9, 0, 0, 0, 0])
    got = np.diag(co.rbox_iou(a, b))
    err = np.array([abs(got[i] - exact(a[i], b[i])) for i in range(n)])
    assert err.max() < 5e-9  # (7e-10 with this seed)
    assert err.max() > 1e-13  # (if this ever fails the oracle got better: tighten the GPU test)


def test_known_answers_in_exact_arithmetic():
    sq = [0.0, 0.0, 1.0, 1.0, 0.0]
    assert exact(sq, sq) == 1.0
    inter = 2 * (np.sqrt(2) - 1)
    assert abs(exact(sq, [0, 0, 1, 1, np.pi / 4]) - inter / (2 - inter)) < 1e-15
    assert exact(sq, [3.0, 0, 1, 1, 0.3]) == 0.0
    assert abs(exact([0, 0, 2, 4, 0.0], [1.0, 0, 2, 4, 0.0]) - (3 * 2) / (16 - 6)) < 1e-15  # length h = 4 lies along x at yaw 0
