"""Seeded synthetic workloads of SURVEY.md §8(d), shared by tests and bench.py (no oracle, no GPU)."""
import numpy as np

import bev
from bev.homo import compose_H_bev_img, homo_from_pts


def frame(idx, h, w, dtype, c=3):
    rng = np.random.default_rng(1234 + idx)
    if np.dtype(dtype) == np.uint8:
        return rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    return rng.random((h, w, c), dtype=np.float32)


def synth_brno_H(src_w, src_h, dst_w, dst_h):
    """Parity homography: BrnoCompSpeed-like vanishing-point calibration + a 64 m x 64 m BEV window,
    composed exactly as vis_homo.py:61-63 does."""
    calib = bev.Calib(vp1=np.array([1200.0, -300.0]), vp2=np.array([-2500.0, -150.0]), pp=np.array([959.5, 539.5]),
                      height=8, u_size=1920, v_size=1080)
    if (src_w, src_h) != (1920, 1080):
        calib = calib.scale(align_corners=False, new_u=src_w, new_v=src_h)
    center = calib.gen_center_in_world()
    bspec = bev.BEVWorldSpec(u_size=dst_w, v_size=dst_h, u_axis="y", v_axis="-x", x_size=64, y_size=64,
                             x_min=center[0] - 20, y_min=center[1] - 32)
    return compose_H_bev_img(bspec.gen_H_world_bev(), calib.gen_H_world_img())


def keystone_H(src_w, src_h, dst_w, dst_h):
    """Roofline homography: the whole BEV samples inside the frame, footprint ~ 80 % of it.
    Returns the FORWARD matrix (src px -> dst px) like every warpPerspective caller passes."""
    dst = np.array([[0, 0], [dst_w - 1, 0], [dst_w - 1, dst_h - 1], [0, dst_h - 1]], dtype=np.float64)
    src = np.array([[0.2 * (src_w - 1), 0], [0.8 * (src_w - 1), 0], [src_w - 1, src_h - 1], [0, src_h - 1]], dtype=np.float64)
    return homo_from_pts(src, dst)


def keystone_inset_H(src_w, src_h, dst_w, dst_h, inset=8.0):
    """The keystone footprint pulled `inset` pixels inside the frame on every side: no tap of any frame of a jittered batch
    touches the frame's edge (measurement aid: the all-interior case)."""
    dst = np.array([[0, 0], [dst_w - 1, 0], [dst_w - 1, dst_h - 1], [0, dst_h - 1]], dtype=np.float64)
    src = np.array([[0.2 * (src_w - 1), inset], [0.8 * (src_w - 1), inset], [src_w - 1 - inset, src_h - 1 - inset], [inset, src_h - 1 - inset]],
                   dtype=np.float64)
    return homo_from_pts(src, dst)


def rotated_H(src_w, src_h, dst_w, dst_h, degrees, zoom=1.2):
    """A similarity footprint: the BEV window is the frame's centre region turned by `degrees` (source pixels per BEV pixel =
    `zoom`), small enough to stay inside the frame for every angle.  Measurement aid: row segments of the BEV cross
    256 * zoom * sin(angle) source rows."""
    t = np.deg2rad(degrees)
    c, s = np.cos(t) * zoom, np.sin(t) * zoom
    # dst -> src: rotate about the BEV centre, land on the frame centre
    A = np.array([[c, -s, 0.0], [s, c, 0.0], [0, 0, 1.0]])
    T0 = np.array([[1, 0, -(dst_w - 1) / 2.0], [0, 1, -(dst_h - 1) / 2.0], [0, 0, 1.0]])
    T1 = np.array([[1, 0, (src_w - 1) / 2.0], [0, 1, (src_h - 1) / 2.0], [0, 0, 1.0]])
    return np.linalg.inv(T1 @ A @ T0)  # forward map (src -> dst), as warpPerspective callers pass it


def jitter_H(H, idx, px=2.0):
    """Per-frame variant: pre-multiply by a seeded +-px translation of the destination."""
    rng = np.random.default_rng(99 + idx)
    tx, ty = rng.uniform(-px, px, 2) if idx > 0 else (0.0, 0.0)
    T = np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1.0]])
    return T @ H
