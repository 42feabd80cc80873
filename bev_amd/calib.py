"""Calib: the camera -> road-plane model that yields H_world_img (float64 numpy, host).

Mirrors /root/reference/bev/calib.py:7-267: same keyword set, attribute names, `.mode`
selection and method signatures.  Three ways to pin the image<->world homography:

  from_KRt  intrinsics K (or fx, fy, cx, cy) + extrinsics T (or R, t)   -- full geometry
  from_vps  two vanishing points + principal point + camera height        -- K, R; scale from height
  from_pts  point correspondences image <-> world plane                   -- H only

scale / pad / flip return NEW Calib objects describing the resized / padded / mirrored image.
Known defect of the reference NOT reproduced: calib.py:76 (R, t without T) raises TypeError
there; here T is assembled from R and t.
"""
import numpy as np

from .frozen_class import FrozenClass
from .homo import homo_from_KRt, homo_from_pts, homo_from_vps

_MODES = ("from_KRt", "from_pts", "from_vps")


def _stack_T(R, t):
    top = np.concatenate((R, np.asarray(t).reshape(3, 1)), axis=1)
    return np.concatenate((top, np.array([[0, 0, 0, 1]])), axis=0).astype(np.float32)


class Calib(FrozenClass):
    def __init__(self, **kwargs):
        # intrinsics
        self.K = None
        self.fx = self.fy = self.cx = self.cy = 0
        self.dist_coeff = None
        # extrinsics (world -> camera)
        self.R = self.t = self.T = None
        # plane correspondences
        self.pts_world = self.pts_image = None
        self.H_world_img = self.H_img_world = None
        # vanishing-point model
        self.vp1 = self.vp2 = self.pp = None
        self.height = None
        self.u_size = self.v_size = None
        self.mode = None
        self._freeze()
        self.__dict__.update(kwargs)

        if self.pts_image is not None and self.pts_world is not None:
            self.mode = "from_pts"
        elif self.vp1 is not None and self.vp2 is not None:
            self.mode = "from_vps"
        else:
            self.mode = "from_KRt"
        self.update()
        self.check_validity()

    def _krt_missing(self):
        return [self.K is None, self.R is None, self.t is None, self.T is None]

    def update(self):
        """Complete K / R / t / T from whichever of them (or fx, fy, cx, cy) were given."""
        missing = self._krt_missing()
        if all(missing):
            pass
        elif any(missing):
            assert self.K is not None or all(v is not None for v in (self.fx, self.fy, self.cx, self.cy))
            if self.K is None:
                # float32 on purpose: the reference builds K this way (calib.py:74) and its numbers carry it
                self.K = np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1]], dtype=np.float32)
            if self.T is None and self.R is not None and self.t is not None:
                self.T = _stack_T(self.R, self.t)
            elif self.T is not None and self.R is None and self.t is None:
                self.R = self.T[:3, :3]
                self.t = self.T[:3, 3]
            else:
                raise ValueError("R,t,T not valid", self.R, self.t, self.T)
        else:
            assert np.allclose(self.T, _stack_T(self.R, self.t)), "{} {} {}".format(self.R, self.t, self.T)

        if self.mode == "from_vps" and self.pp is None:
            self.pp = np.zeros_like(self.vp1)
            self.pp[0] = (self.u_size - 1) * 0.5
            self.pp[1] = (self.v_size - 1) * 0.5

    def check_validity(self):
        missing = self._krt_missing()
        assert all(missing) or not any(missing)
        if not any(missing):
            assert np.allclose(self.T, _stack_T(self.R, self.t)), "{} {} {}".format(self.R, self.t, self.T)
        if self.mode == "from_pts":
            assert self.pts_image is not None and self.pts_world is not None
        elif self.mode == "from_vps":
            assert all(v is not None for v in (self.vp1, self.vp2, self.pp, self.height, self.u_size, self.v_size))

    def gen_H_world_img(self, mode=None):
        """3x3 with pt_world ~ H @ pt_img (calib.py:109-127)."""
        self.check_validity()
        mode = self.mode if mode is None else mode
        assert mode in _MODES, mode
        if mode == "from_pts":
            assert self.pts_image is not None and self.pts_world is not None
            return homo_from_pts(self.pts_image, self.pts_world[:, :2])
        if mode == "from_vps":
            H_img_world = homo_from_vps(self.vp1, self.vp2, self.height, self.u_size, self.v_size, self.pp)
        else:
            assert self.R is not None and self.t is not None
            H_img_world = homo_from_KRt(self.K, Rt_homo=self.T)
        return np.linalg.inv(H_img_world)

    def gen_center_in_world(self):
        """The image centre pixel ((u-1)/2, (v-1)/2) dropped onto the road plane: (x, y, 1)."""
        H_world_img = self.gen_H_world_img()
        p = H_world_img.dot(np.array([(self.u_size - 1) / 2, (self.v_size - 1) / 2, 1]).reshape(3))
        return (p / p[2]).reshape(-1)

    # ---- geometric augmentations: each returns a new Calib ----------------------------------------
    def _rebuild(self, u_size, v_size, map_u, map_v, K_edit):
        """map_u / map_v move pixel coordinates of vps / pp / image points; K_edit edits a K copy."""
        if self.mode == "from_KRt":
            K = self.K.copy()
            K_edit(K)
            return Calib(K=K, T=self.T.copy(), u_size=u_size, v_size=v_size)
        if self.mode == "from_vps":
            moved = {}
            for name in ("vp1", "vp2", "pp"):
                p = getattr(self, name).copy()
                p[0], p[1] = map_u(p[0]), map_v(p[1])
                moved[name] = p
            return Calib(height=self.height, u_size=u_size, v_size=v_size, **moved)
        pts_image = self.pts_image.copy()
        pts_image[:, 0] = map_u(pts_image[:, 0])
        pts_image[:, 1] = map_v(pts_image[:, 1])
        return Calib(pts_image=pts_image, pts_world=self.pts_world.copy(), u_size=u_size, v_size=v_size)

    def scale(self, align_corners, new_u=None, new_v=None, scale_ratio_u=None, scale_ratio_v=None):
        """Calib of the resized image (calib.py:142-198); conventions as BEVWorldSpec.scale."""
        if scale_ratio_u is None and scale_ratio_v is None:
            assert new_u is not None and new_v is not None
            if align_corners:
                scale_ratio_u, scale_ratio_v = (new_u - 1) / (self.u_size - 1), (new_v - 1) / (self.v_size - 1)
            else:
                scale_ratio_u, scale_ratio_v = new_u / self.u_size, new_v / self.v_size
        elif align_corners:
            new_u, new_v = scale_ratio_u * (self.u_size - 1) + 1, scale_ratio_v * (self.v_size - 1) + 1
        else:
            new_u, new_v = scale_ratio_u * self.u_size, scale_ratio_v * self.v_size

        def mover(ratio):
            return (lambda p: p * ratio) if align_corners else (lambda p: (p + 0.5) * ratio - 0.5)

        map_u, map_v = mover(scale_ratio_u), mover(scale_ratio_v)

        def K_edit(K):
            K[0, 0] = K[0, 0] * scale_ratio_u
            K[1, 1] = K[1, 1] * scale_ratio_v
            K[0, 2] = map_u(K[0, 2])
            K[1, 2] = map_v(K[1, 2])

        return self._rebuild(new_u, new_v, map_u, map_v, K_edit)

    def pad(self, pad_left, pad_top, pad_right, pad_bottom):
        """Calib of the padded (or, with negative values, cropped) image (calib.py:200-229)."""
        def K_edit(K):
            K[0, 2] = K[0, 2] + pad_left
            K[1, 2] = K[1, 2] + pad_top

        return self._rebuild(self.u_size + pad_left + pad_right, self.v_size + pad_top + pad_bottom,
                             lambda p: p + pad_left, lambda p: p + pad_top, K_edit)

    def flip(self, lr=False, tb=False):
        """Calib of the mirrored image (calib.py:231-267).  In from_KRt mode the flip lives entirely
        in K: the principal point is reflected and fx / fy change sign."""
        u_last, v_last = self.u_size - 1, self.v_size - 1

        def K_edit(K):
            if lr:
                K[0, 2] = u_last - K[0, 2]
                K[0, 0] = -K[0, 0]
            if tb:
                K[1, 2] = v_last - K[1, 2]
                K[1, 1] = -K[1, 1]

        return self._rebuild(self.u_size, self.v_size,
                             (lambda p: u_last - p) if lr else (lambda p: p),
                             (lambda p: v_last - p) if tb else (lambda p: p), K_edit)
