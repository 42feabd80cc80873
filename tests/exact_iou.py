"""Rotated-box IoU in 50-digit arithmetic (mpmath): an INDEPENDENT statement of the quantity the oracle's float64 clip and the HIP kernel's
contour sum both approximate -- intersection area of two rectangles ([x, y, w, h, yaw], length h along the heading: rbox.py:87-95) over
the area of their union.  Test infrastructure only; slow (about a millisecond per pair)."""
import mpmath as mp


def _quad(box):
    cx, cy, w, h, r = [mp.mpf(float(v)) for v in box[:5]]
    c, s = mp.cos(r), mp.sin(r)
    return [(c * x - s * y + cx, s * x + c * y + cy) for x, y in ((-h / 2, -w / 2), (h / 2, -w / 2), (h / 2, w / 2), (-h / 2, w / 2))]


def iou(a, b, digits=50):
    """IoU of box a with box b (positive sizes) as a float, every step in `digits`-digit arithmetic."""
    with mp.workdps(digits):
        P, Q = _quad(a), _quad(b)
        for e in range(4):
            (bx, by), (nx, ny) = Q[e], Q[(e + 1) % 4]
            ex, ey = nx - bx, ny - by
            out = []
            for i in range(len(P)):
                (xi, yi), (xj, yj) = P[i], P[(i + 1) % len(P)]
                di, dj = ex * (yi - by) - ey * (xi - bx), ex * (yj - by) - ey * (xj - bx)
                if di >= 0:
                    out.append((xi, yi))
                if (di >= 0) != (dj >= 0):
                    t = di / (di - dj)
                    out.append((xi + t * (xj - xi), yi + t * (yj - yi)))
            P = out
            if not P:
                break
        twice = mp.mpf(0)
        for i in range(len(P)):
            (xi, yi), (xj, yj) = P[i], P[(i + 1) % len(P)]
            twice += xi * yj - xj * yi
        inter = abs(twice) / 2
        area = abs(mp.mpf(float(a[2])) * mp.mpf(float(a[3]))) + abs(mp.mpf(float(b[2])) * mp.mpf(float(b[3])))
        return float(inter / (area - inter)) if area - inter > 0 else 0.0
