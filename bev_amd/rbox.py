"""Rotated-box and point transforms between BEV pixels, the world plane and the image (numpy).

Mirrors the names of /root/reference/bev/rbox.py:20-314.  Conventions (rbox.py:5-18):
  * BEV:   u right, v down; yaw 0 points along +v, yaw = atan2(du, dv).
  * world: right-handed x/y; yaw 0 points along +x, yaw = atan2(dy, dx), CCW positive.
  * rbox = [x, y, w (width), h (length), yaw]; the length runs along the yaw direction.
The bulk point projection `pts_world_bev` has a device twin: bev_amd.points.project_points.

Deviation: `xywhr2xyxy(..., external_aa=True)` raises IndexError in the reference (rbox.py:71,83
write columns 4..7 of a 4-column array); here it returns the intended [x_min, y_min, x_max, y_max].
"""
import numpy as np

from .homo import homo_from_KRt

_MODES = ("bev", "world")


def _other(frame):
    return "world" if frame == "bev" else "bev"


def v2yaw(x, mode):
    assert mode in _MODES
    a, b = (x[:, 0], x[:, 1]) if mode == "bev" else (x[:, 1], x[:, 0])
    return np.arctan2(a, b)


def yaw2v(x, mode):
    assert mode in _MODES
    s, c = np.sin(x), np.cos(x)
    return np.stack((s, c) if mode == "bev" else (c, s), axis=1)


def yaw2mat(x, mode):
    """n x 2 x 2 rotation taking box-local offsets to frame offsets."""
    assert mode in _MODES
    x = x.reshape(-1, 1)
    s, c = np.sin(x), np.cos(x)
    cols = [c, s, -s, c] if mode == "bev" else [c, -s, s, c]
    return np.concatenate(cols, axis=1).reshape(-1, 2, 2)


# corner sign pattern (x, y per corner) times (half extent along frame x, along frame y)
_SIGNS = {"bev": np.array([-1, -1, -1, 1, 1, 1, 1, -1.0]),      # TL, BL, BR, TR in the u-right/v-down raster
          "world": np.array([-1, -1, 1, -1, 1, 1, -1, 1.0])}


def xywhr2xyxy(x, mode, external_aa=False):
    """n x 5 [x, y, w, h, yaw] -> n x 8 corner coordinates (or n x 4 axis-aligned hull).
    "bev": w spans u, h spans v at yaw 0; "world": h spans x, w spans y at yaw 0 (rbox.py:65-112)."""
    assert mode in _MODES
    half_x, half_y = (x[:, 2], x[:, 3]) if mode == "bev" else (x[:, 3], x[:, 2])
    local = np.zeros((x.shape[0], 8), dtype=x.dtype)
    local[:, 0::2] = _SIGNS[mode][0::2] * half_x[:, None] / 2
    local[:, 1::2] = _SIGNS[mode][1::2] * half_y[:, None] / 2
    pts = np.matmul(yaw2mat(x[:, 4], mode), local.reshape(-1, 4, 2).swapaxes(1, 2))  # n x 2 x 4
    y = pts.swapaxes(1, 2).reshape(-1, 8)
    y += x[:, [0, 1, 0, 1, 0, 1, 0, 1]]
    if not external_aa:
        return y
    xs, ys = y[:, 0::2], y[:, 1::2]
    return np.stack([xs.min(axis=1), ys.min(axis=1), xs.max(axis=1), ys.max(axis=1)], axis=1)


def xy82xywhr(xy8, mode):
    assert mode in _MODES
    top_left, bot_left, top_right = xy8[:, 0:2], xy8[:, 2:4], xy8[:, 6:8]
    w = np.sqrt(((top_right - top_left) ** 2).sum(1, keepdims=True))
    h = np.sqrt(((bot_left - top_left) ** 2).sum(1, keepdims=True))
    xy = 0.5 * (bot_left + top_right)
    r = v2yaw(top_left - bot_left, mode).reshape(-1, 1)
    return np.concatenate([xy, w, h, r], axis=1)


def xywhr2xyvec(xywhr, mode):
    """[x_start, y_start, x_end, y_end] of the heading vector scaled by the box length."""
    assert mode in _MODES
    vs = yaw2v(xywhr[:, 4], mode) * xywhr[:, 3:4]
    xs, ys = xywhr[:, 0], xywhr[:, 1]
    return np.stack([xs, ys, xs + vs[:, 0], ys + vs[:, 1]], axis=1)


def xy82xyvec(xy8):
    vs = xy8[:, 2:4] - xy8[:, :2]
    xs = 0.5 * (xy8[:, 0] + xy8[:, 4])
    ys = 0.5 * (xy8[:, 1] + xy8[:, 5])
    return np.stack([xs, ys, xs + vs[:, 0], ys + vs[:, 1]], axis=1)


def pts_world_bev(pts_src, H):
    """Project points through a homography and dehomogenise (rbox.py:136-151).
    N x 2 in -> N x 2 out; N x 3 (homogeneous) in -> N x 3 out; a single 1-D point is accepted."""
    pts_src = np.array(pts_src)
    if pts_src.ndim == 1:
        pts_src = pts_src[None, :]
    homogeneous_in = pts_src.shape[1] != 2
    if not homogeneous_in:
        pts_src = np.concatenate([pts_src, np.ones_like(pts_src[:, [0]])], axis=1)
    assert pts_src.shape[1] == 3
    pts_tgt = H.dot(pts_src.T).T
    pts_tgt = pts_tgt / pts_tgt[:, [2]]
    return pts_tgt if homogeneous_in else pts_tgt[:, :2]


def dist_world_bev(dist_src, H):
    """Scale lengths by the similarity's scale (column norms; both columns must agree)."""
    scale = np.sqrt(H[0, 0] ** 2 + H[1, 0] ** 2)
    scale_1 = np.sqrt(H[0, 1] ** 2 + H[1, 1] ** 2)
    assert np.abs(scale - scale_1) < 1e-5
    return scale * dist_src


def angle_world_bev(angle_src, H, src):
    assert src in _MODES
    angle = np.array(angle_src).reshape(-1)
    v_src = np.concatenate([yaw2v(angle, src), np.zeros_like(angle)[..., None]], axis=1)  # directions: w = 0
    v_tgt = H.dot(v_src.T).T[:, :2]
    return v2yaw(v_tgt, _other(src))


def _normalised_similarity(H):
    H = H / H[2, 2]
    assert np.abs(H[2, 0]) + np.abs(H[2, 1]) < 1e-5
    return H


def rbox_world_bev(rbox_src, H, src):
    """n x 5 rboxes from frame `src` ("bev" | "world") to the other frame through a similarity H
    (translation, rotation, reflection, uniform scale), e.g. H_world_bev (rbox.py:173-219)."""
    assert src in _MODES
    H = _normalised_similarity(H)
    if len(rbox_src) == 0:
        return rbox_src
    r_tgt = angle_world_bev(rbox_src[:, 4], H, src)
    xy_tgt = pts_world_bev(rbox_src[:, :2], H)
    wh_tgt = dist_world_bev(rbox_src[:, 2:4], H)
    return np.concatenate([xy_tgt, wh_tgt, r_tgt[..., None]], axis=1)


def rbox_world_img(rbox_world, H_img_world):
    """Image pixel of each box centre (rbox.py:221-226)."""
    return pts_world_bev(rbox_world[:, :2], H_img_world)


def _ground_shadow(xyz, K, Rt, H_world_cam):
    """World plane point seen through the same pixel as the 3-D point xyz (3 x N)."""
    cam = Rt[:3, :3].dot(xyz) + Rt[:3, [3]]
    uvd = K.dot(cam)
    uv1 = uvd / np.clip(uvd[2], a_min=1e-2, a_max=None)
    xy1 = H_world_cam.dot(uv1)
    xy1 = xy1 / xy1[2]
    assert (xy1[2] == 1).all(), "{}".format(xy1)
    return xy1


def rbox_zt2tt_world(rboxzt, K, Rt):
    """[x,y,w,h,r,z,t(all)] -> [x',y',w,h,r,dx,dy] in the world plane: the box bottom centre and the
    offset to where its top centre appears on the ground from this camera (rbox.py:228-256)."""
    H_world_cam = np.linalg.inv(homo_from_KRt(K=K, Rt_homo=Rt))
    low = _ground_shadow(rboxzt[:, [0, 1, 5]].T, K, Rt, H_world_cam)
    xyz_high = rboxzt[:, [0, 1, 5]].T
    xyz_high[2] = xyz_high[2] + rboxzt[:, 6].T
    high = _ground_shadow(xyz_high, K, Rt, H_world_cam)
    return np.concatenate([low[:2].T, rboxzt[:, 2:5], (high[:2] - low[:2]).T], axis=1)


def rboxtt_world_bev(rbox_src, H, src):
    """n x 7 [x,y,w,h,r,dx,dy] between frames; the tail offset transforms as a vector (rbox.py:258-288)."""
    assert src in _MODES
    if len(rbox_src) == 0:
        return rbox_src
    H = _normalised_similarity(H)
    assert rbox_src.shape[1] == 7
    ones = np.ones((rbox_src.shape[0], 1))
    start = H.dot(np.concatenate([rbox_src[:, :2], ones], axis=1).T)
    assert (start[2] == 1).all(), "{}".format(start)
    end = H.dot(np.concatenate([rbox_src[:, :2] + rbox_src[:, 5:], ones], axis=1).T)
    assert (end[2] == 1).all(), "{}".format(end)
    return np.concatenate([rbox_world_bev(rbox_src[:, :5], H, src), (end - start)[:2].T], axis=1)


def rboxzt_world_bev(rbox_src, H, K, Rt, src):
    """xywhr + (z, height) in the world -> xywhr + tail offset in the BEV (rbox.py:291-314)."""
    assert src in _MODES
    if len(rbox_src) == 0:
        return rbox_src
    H = _normalised_similarity(H)
    assert rbox_src.shape[1] == 7
    if src != "world":
        raise NotImplementedError("rboxzt_world_bev only supports converting from world to bev")
    return rboxtt_world_bev(rbox_zt2tt_world(rbox_src, K, Rt), H, src)
