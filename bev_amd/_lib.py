"""ctypes binding of bev_amd/csrc/libbevwarp.so (C ABI: include/bevwarp.h).

There is no CPU fallback: if the shared library is missing or a call fails, this raises.
`build()` compiles the library in-tree with hipcc for gfx950 (cross-compiles without a GPU).
"""
import ctypes
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("BEVWARP_LIB") or os.path.join(_CSRC, "libbevwarp.so")  # override = A/B builds

U8, F32, F64 = 0, 1, 2
INTER_NEAREST, INTER_LINEAR = 0, 1
ABI_VERSION = 7

# every symbol include/bevwarp.h declares: (name, restype, argtypes)
_c = ctypes
SYMBOLS = {
    "bevwarp_version": (_c.c_int, []),
    "bevwarp_strerror": (_c.c_char_p, [_c.c_int]),
    "bevwarp_last_hip_error": (_c.c_char_p, []),
    "bevwarp_invert_homography": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int]),
    "bevwarp_warp": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                _c.c_void_p, _c.c_void_p]),
    "bevwarp_warp_classes": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                        _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                        _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p]),
    "bevwarp_tile_classes_bytes": (_c.c_int64, [_c.c_int] * 8),
    "bevwarp_warp_planar": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                       _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                       _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "bevwarp_composite": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p]),
    "bevwarp_warp_composite": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int64,
                                          _c.c_int64, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int64, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p]),
    "bevwarp_resize": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int64,
                                  _c.c_int64, _c.c_int, _c.c_int, _c.c_void_p]),
    "bevwarp_footprint": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int,
                                     _c.c_int, _c.c_void_p]),
    "bevwarp_project_points": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p]),
    "bevwarp_rbox_iou": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int,
                                    _c.c_void_p]),
    "bevwarp_rbox_transform": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p]),
    "bevwarp_tracker_step": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p,
                                        _c.c_double, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p]),
}

_lib = None


class BevWarpError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """hipcc the HIP sources into csrc/libbevwarp.so (gfx950)."""
    cmd = ["make", "-C", _CSRC, "-j8"] + (["-B"] if force else [])
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(cmd, stdout=out)
    return LIB_PATH


def load():
    """The loaded library with typed entry points.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BevWarpError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(or `make -C bev_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and this table ever drift apart
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.bevwarp_version() != ABI_VERSION:
            raise BevWarpError("libbevwarp ABI %d, expected %d" % (lib.bevwarp_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(status):
    if status != 0:
        lib = load()
        msg = lib.bevwarp_strerror(status).decode()
        if status == -5:
            raise BevWarpError("%s: %s" % (msg, lib.bevwarp_last_hip_error().decode()))
        raise ValueError("libbevwarp: " + msg)
