"""BEVWorldSpec: the BEV raster <-> world rectangle correspondence (float64 numpy, host).

Mirrors /root/reference/bev/bev.py:6-175: same constructor keywords, attribute names,
defaults (u_axis "-x", v_axis "y"), assertions and method signatures, so callers such as
vis_homo.py:59-62 keep working.  The default pixel correspondence uses (0,0)..(u_size,v_size)
-- size, not size-1 (bev.py:71); after scale()/pad() the explicit u_min..v_max are used.
"""
import numpy as np

from .frozen_class import FrozenClass
from .homo import homo_from_pts

# gen_bev_corners_in_world lists BEV corners as top-left, bottom-left, bottom-right, top-right.
# World rectangle corners are numbered 0:(x_min,y_min) 1:(x_min,y_max) 2:(x_max,y_max) 3:(x_max,y_min);
# each legal (u_axis, v_axis) picks the world corner under each BEV corner (bev.py:85-100).
_CORNER_ORDER = {
    ("x", "y"): (0, 1, 2, 3),
    ("x", "-y"): (1, 0, 3, 2),
    ("-x", "-y"): (2, 3, 0, 1),
    ("-x", "y"): (3, 2, 1, 0),
    ("y", "x"): (0, 3, 2, 1),
    ("y", "-x"): (3, 0, 1, 2),
    ("-y", "-x"): (2, 1, 0, 3),
    ("-y", "x"): (1, 2, 3, 0),
}
_AXES = ("x", "y", "-x", "-y")


def _negate_axis(name):
    return name[1:] if name.startswith("-") else "-" + name


class BEVWorldSpec(FrozenClass):
    def __init__(self, u_size, v_size, **kwargs):
        self.u_size = u_size
        self.v_size = v_size
        self.u_axis = "-x"
        self.v_axis = "y"
        for name in ("x_size", "y_size", "x_min", "x_max", "y_min", "y_max",
                     # pixel coordinates of the world rectangle's edges once the raster was scaled / padded
                     "u_min", "u_max", "v_min", "v_max"):
            setattr(self, name, None)
        self._freeze()
        self.__dict__.update(kwargs)
        self.update()
        self.check_validity()

    def set_keep(self, **kwargs):
        """Overwrite fields; to change one of (min, max, size) pass a related one as None as well."""
        self.__dict__.update(kwargs)
        self.update()

    def _complete_axis(self, a):
        lo, hi, size = (getattr(self, "%s_%s" % (a, k)) for k in ("min", "max", "size"))
        missing = [v is None for v in (lo, hi, size)]
        if any(missing):
            assert sum(missing) == 1, np.array([lo, hi, size])
            if lo is None:
                lo = hi - size
            elif hi is None:
                hi = lo + size
            else:
                size = hi - lo
            setattr(self, a + "_min", lo)
            setattr(self, a + "_max", hi)
            setattr(self, a + "_size", size)
        else:
            assert size == hi - lo

    def update(self):
        """Fill in whichever of (min, max, size) is None per world axis; if none is, they must agree."""
        self._complete_axis("x")
        self._complete_axis("y")

    def check_validity(self):
        for a in ("x", "y"):
            lo, hi, size = (getattr(self, "%s_%s" % (a, k)) for k in ("min", "max", "size"))
            assert lo is not None and hi is not None and size is not None
            assert np.isclose(size, hi - lo)
        assert self.u_axis in _AXES
        assert self.v_axis in _AXES
        assert ("x" in self.u_axis and "y" in self.v_axis) or ("y" in self.u_axis and "x" in self.v_axis)

    def _pixel_extent(self):
        if self.u_min is None:
            return 0, self.u_size, 0, self.v_size
        return self.u_min, self.u_max, self.v_min, self.v_max

    def gen_H_world_bev(self):
        """3x3 with pt_world ~ H @ pt_bev from the four corner correspondences (bev.py:67-79)."""
        self.check_validity()
        u0, u1, v0, v1 = self._pixel_extent()
        pts_bev = np.array([[u0, v0], [u0, v1], [u1, v1], [u1, v0]], dtype=float)
        return homo_from_pts(pts_bev, self.gen_bev_corners_in_world())

    def gen_bev_corners_in_world(self):
        """World (x, y) under the BEV's top-left, bottom-left, bottom-right, top-right corners."""
        key = (self.u_axis, self.v_axis)
        if key not in _CORNER_ORDER:
            raise ValueError("illegal u_axis and v_axis combo", self.u_axis, self.v_axis)
        rect = np.array([[self.x_min, self.y_min], [self.x_min, self.y_max],
                         [self.x_max, self.y_max], [self.x_max, self.y_min]], dtype=float)
        return rect[list(_CORNER_ORDER[key])]

    def _derive(self, u_size, v_size, extent, u_axis=None, v_axis=None):
        u0, u1, v0, v1 = extent
        return BEVWorldSpec(u_size=u_size, v_size=v_size, u_axis=u_axis or self.u_axis, v_axis=v_axis or self.v_axis,
                            x_size=self.x_size, y_size=self.y_size, x_min=self.x_min, y_min=self.y_min,
                            u_min=u0, v_min=v0, u_max=u1, v_max=v1)

    def scale(self, align_corners, new_u=None, new_v=None, scale_ratio_u=None, scale_ratio_v=None):
        """New spec for a resized raster (bev.py:107-140).  align_corners=True aligns the centres of
        the corner pixels (ratio = (new-1)/(old-1)); False aligns their outer corners (ratio = new/old,
        pixel coordinate p -> (p + 0.5) * ratio - 0.5).  Pass sizes or ratios, consistently."""
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

        def remap(p, ratio):
            return p * ratio if align_corners else (p + 0.5) * ratio - 0.5

        u0, u1, v0, v1 = self._pixel_extent()
        return self._derive(new_u, new_v, (remap(u0, scale_ratio_u), remap(u1, scale_ratio_u),
                                           remap(v0, scale_ratio_v), remap(v1, scale_ratio_v)))

    def pad(self, pad_left, pad_top, pad_right, pad_bottom):
        u0, u1, v0, v1 = self._pixel_extent()
        return self._derive(self.u_size + pad_left + pad_right, self.v_size + pad_top + pad_bottom,
                            (u0 + pad_left, u1 + pad_left, v0 + pad_top, v1 + pad_top))

    def flip(self, lr=False, tb=False):
        """Mirror the raster by negating the world axis it runs along (bev.py:162-175)."""
        return self._derive(self.u_size, self.v_size, (self.u_min, self.u_max, self.v_min, self.v_max),
                            u_axis=_negate_axis(self.u_axis) if lr else self.u_axis,
                            v_axis=_negate_axis(self.v_axis) if tb else self.v_axis)
