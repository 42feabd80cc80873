"""3x3 homography algebra -- host side of the BEV warp path (float64 numpy).

Mirrors the public names of /root/reference/bev/homo.py:6-135 (same signatures, defaults,
assertions and the `H_<target>_<source>` convention: pt_target ~ H @ pt_source) without
OpenCV: `homo_from_pts` is an in-repo normalised DLT (+ Gauss-Newton polish for n > 4)
instead of cv2.findHomography, `Rt_from_homo_K` uses numpy's SVD instead of cv2.SVDecomp.
"""
import math

import numpy as np

__all__ = ["homo_from_KRt", "homo_from_pts", "get_focal", "get_K_from_f_pp", "get_K_from_vps", "homo_from_vps",
           "get_vps_from_homo", "get_KRt_from_homo", "Rt_from_homo_K", "Rt_from_pts_K_dist", "compose_H_bev_img"]

_PLANE_COLS = [0, 1, 3]  # world plane z = 0: drop the z column of [R|t]


def homo_from_KRt(K, R=None, t=None, Rt_homo=None):
    """H_img_world = K[:, :3] @ [r1 r2 t]  for the world plane z = 0 (reference homo.py:6-26).

    Give either `Rt_homo` (4x4 or 3x4) or both `R` (3x3) and `t` (3,) / (3,1).  K may be 3x4.
    The result is NOT normalised."""
    K = np.asarray(K)
    if K.shape[1] == 4:
        K = K[:, :3]
    if Rt_homo is not None:
        assert R is None and t is None
        plane = np.asarray(Rt_homo)[:3][:, _PLANE_COLS]
    else:
        assert R is not None and t is not None
        plane = np.concatenate((np.asarray(R)[:, :2], np.asarray(t).reshape(-1, 1)), axis=1)
    return K.dot(plane)


def _dlt_normalisation(pts):
    """Similarity that moves the centroid to 0 and the mean |deviation| per axis to 1."""
    c = pts.mean(axis=0)
    dev = np.abs(pts - c).mean(axis=0)
    s = np.where(dev > np.finfo(float).eps, 1.0 / np.where(dev > 0, dev, 1.0), 1.0)
    return np.array([[s[0], 0.0, -c[0] * s[0]], [0.0, s[1], -c[1] * s[1]], [0.0, 0.0, 1.0]])


def _apply(H, pts):
    q = np.concatenate([pts, np.ones((len(pts), 1))], axis=1) @ H.T
    return q[:, :2] / q[:, 2:3]


def homo_from_pts(pts_src, pts_tgt):
    """H with pts_tgt ~ H @ pts_src, least squares, h33 = 1 (reference homo.py:29-38 ->
    cv2.findHomography(method=0)).  pts are n x 2 arrays, n >= 4.

    n == 4 has a unique solution; for n > 4 the algebraic DLT solution is refined on the
    reprojection error, as OpenCV does, so results agree to solver tolerance (not bitwise)."""
    pts_src = np.asarray(pts_src)
    pts_tgt = np.asarray(pts_tgt)
    assert pts_src.ndim == 2 and pts_src.shape[1] == 2, pts_src.shape
    assert pts_tgt.ndim == 2 and pts_tgt.shape[1] == 2, pts_tgt.shape
    assert pts_src.shape[0] == pts_tgt.shape[0] and pts_src.shape[0] >= 4, (pts_src.shape, pts_tgt.shape)
    a = pts_src.astype(np.float64)
    b = pts_tgt.astype(np.float64)
    Ta, Tb = _dlt_normalisation(a), _dlt_normalisation(b)
    an, bn = _apply(Ta, a), _apply(Tb, b)
    n = len(a)
    A = np.zeros((2 * n, 9))
    A[0::2, 0:2], A[0::2, 2] = an, 1.0
    A[0::2, 6:8], A[0::2, 8] = -bn[:, :1] * an, -bn[:, 0]
    A[1::2, 3:5], A[1::2, 5] = an, 1.0
    A[1::2, 6:8], A[1::2, 8] = -bn[:, 1:] * an, -bn[:, 1]
    h = np.linalg.svd(A)[2][-1].reshape(3, 3)
    H = np.linalg.inv(Tb) @ h @ Ta
    if abs(H[2, 2]) < 1e-300:
        return H
    H = H / H[2, 2]
    if n > 4:
        H = _polish_reprojection(H, a, b)
    return H


def _polish_reprojection(H, a, b, iters=10):
    """Gauss-Newton on sum |proj(H, a) - b|^2 over the 8 free entries (h33 fixed to 1)."""
    h = H.ravel()[:8].copy()
    for _ in range(iters):
        Hc = np.append(h, 1.0).reshape(3, 3)
        q = np.concatenate([a, np.ones((len(a), 1))], axis=1) @ Hc.T
        w = 1.0 / q[:, 2]
        px, py = q[:, 0] * w, q[:, 1] * w
        r = np.concatenate([px - b[:, 0], py - b[:, 1]])
        J = np.zeros((2 * len(a), 8))
        n = len(a)
        J[:n, 0], J[:n, 1], J[:n, 2] = a[:, 0] * w, a[:, 1] * w, w
        J[:n, 6], J[:n, 7] = -a[:, 0] * px * w, -a[:, 1] * px * w
        J[n:, 3], J[n:, 4], J[n:, 5] = a[:, 0] * w, a[:, 1] * w, w
        J[n:, 6], J[n:, 7] = -a[:, 0] * py * w, -a[:, 1] * py * w
        step = np.linalg.lstsq(J, -r, rcond=None)[0]
        h = h + step
        if np.linalg.norm(step) <= 1e-14 * max(1.0, np.linalg.norm(h)):
            break
    return np.append(h, 1.0).reshape(3, 3)


def get_focal(vp1, vp2, pp):
    """f = sqrt(-(vp1 - pp) . (vp2 - pp)) for two orthogonal vanishing points (homo.py:40-41)."""
    return math.sqrt(-np.dot(vp1[0:2] - pp[0:2], vp2[0:2] - pp[0:2]))


def get_K_from_f_pp(focal, pp):
    return np.array([[focal, 0, pp[0]], [0, focal, pp[1]], [0, 0, 1]])


def get_K_from_vps(vp1, vp2, pp):
    focal = get_focal(vp1, vp2, pp)
    return get_K_from_f_pp(focal, pp), focal


def homo_from_vps(vp1, vp2, height, u_size, v_size, pp=None):
    """H_img_world from two vanishing points and the camera height (reference homo.py:52-96;
    Dubska et al. 2015).  vp1, vp2, pp are (u, v); pp defaults to the image centre
    ((u_size-1)/2, (v_size-1)/2).  World x / y run along the vp1 / vp2 directions, the road
    plane is z = 0 and the camera sits `height` above it."""
    if pp is None:
        pp = np.array([(u_size - 1) * 0.5, (v_size - 1) * 0.5])
    K, focal = get_K_from_vps(vp1, vp2, pp)

    def lift(p, z):
        return np.concatenate((p, [z]))

    pp_w = lift(pp, 0)
    d1 = lift(vp1, focal) - pp_w
    d2 = lift(vp2, focal) - pp_w
    n = np.cross(d1, d2)
    # third vanishing point re-projected onto the image plane z = focal, then as a direction
    vp3 = n[0:2] / n[2] * focal + pp
    d3 = lift(vp3, focal) - pp_w
    d1, d2, d3 = (d / np.linalg.norm(d) for d in (d1, d2, d3))

    M = np.stack((lift(d1, 0), lift(d2, 0), lift(d3, -1 * height), [0, 0, 0, 1]), axis=0)
    M_inv_43 = np.linalg.inv(M)[:, _PLANE_COLS]
    K_34 = np.concatenate((K, np.zeros((3, 1), dtype=int)), axis=1)
    return np.dot(K_34, M_inv_43)


def get_vps_from_homo(H_img_world):
    """Vanishing points of the world x and y axes: columns 0 and 1 dehomogenised (homo.py:98-102)."""
    H = H_img_world
    return (np.array([H[0, 0] / H[2, 0], H[1, 0] / H[2, 0]]), np.array([H[0, 1] / H[2, 1], H[1, 1] / H[2, 1]]))


def get_KRt_from_homo(H_img_world, pp):
    vp1, vp2 = get_vps_from_homo(H_img_world)
    K, focal = get_K_from_vps(vp1, vp2, pp)
    R, t = Rt_from_homo_K(H_img_world, K)
    return K, focal, R, t


def Rt_from_homo_K(H_img_world, K):
    """Pose from a plane homography (reference homo.py:111-128): K^-1 H scaled so |r1| = 1,
    r3 = r1 x r2, then the nearest rotation via SVD (numpy instead of cv2.SVDecomp)."""
    G = np.linalg.inv(K).dot(H_img_world)
    G = G / np.sqrt((G[:, 0] ** 2).sum())
    r1, r2, tvec = G[:, 0], G[:, 1], G[:, 2]
    u, _, vt = np.linalg.svd(np.stack((r1, r2, np.cross(r1, r2)), axis=1))
    return np.matmul(u, vt), tvec


def Rt_from_pts_K_dist(pts_world, pts_img, K, dist_coeffs):
    """Reference homo.py:130-135 calls cv2.solvePnP + cv2.Rodrigues.  Without OpenCV only the
    case this package needs is supported: coplanar world points (z == 0) and no lens distortion,
    solved through the plane homography.  Anything else raises NotImplementedError."""
    pts_world = np.asarray(pts_world, dtype=np.float64)
    if dist_coeffs is not None and np.any(np.asarray(dist_coeffs) != 0):
        raise NotImplementedError("lens distortion needs an iterative PnP solver (OpenCV), out of scope")
    if pts_world.shape[1] == 3 and np.any(pts_world[:, 2] != 0):
        raise NotImplementedError("non-coplanar PnP needs OpenCV, out of scope")
    H_img_world = homo_from_pts(pts_world[:, :2], np.asarray(pts_img, dtype=np.float64))
    R, tvec = Rt_from_homo_K(H_img_world, np.asarray(K, dtype=np.float64)[:, :3])
    if tvec[2] < 0:  # keep the points in front of the camera
        R, tvec = R * np.array([-1, -1, 1]), -tvec
    return R, tvec.reshape(3, 1)


def compose_H_bev_img(H_world_bev, H_world_img):
    """H_bev_img = inv(H_world_bev) @ H_world_img -- the forward matrix handed to warpPerspective
    (reference vis_homo.py:61-63, :77-78)."""
    return np.linalg.inv(H_world_bev).dot(H_world_img)
