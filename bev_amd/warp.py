"""Device warp: the HIP replacement for `cv2.warpPerspective(img, H_bev_img, (u_size, v_size))`
(reference call sites vis_homo.py:89, :91; bev/tool/compo.py:38, :46, :47).

    dst = warp_perspective(src, M, (u_size, v_size))            # torch tensors on the GPU, batched
    dst = warpPerspective(img, M, (u_size, v_size))             # numpy in / numpy out, cv2 call shape

M is the FORWARD map (src px -> dst px) exactly as callers hand it to OpenCV; it is inverted on the
host with OpenCV's closed form and the inverse is cached on the device (calibrations are static per
camera).  All pixel work happens in bev_amd/csrc (HIP, gfx950) through the C ABI; nothing here
falls back to the CPU.
"""
import collections
import ctypes

import numpy as np
import torch

from . import _lib

INTER_NEAREST = _lib.INTER_NEAREST
INTER_LINEAR = _lib.INTER_LINEAR
WARP_INVERSE_MAP = 16  # cv2 flag value
BORDER_CONSTANT = 0

_DTYPES = {torch.uint8: _lib.U8, torch.float32: _lib.F32}


def invert_homography(M):
    """Host float64 inverse(s) of 3x3 matrices, OpenCV evaluation order.  (..., 3, 3) -> same shape."""
    M = np.ascontiguousarray(M, dtype=np.float64)
    assert M.shape[-2:] == (3, 3), M.shape
    out = np.empty_like(M)
    _lib.check(_lib.load().bevwarp_invert_homography(M.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
                                                     int(M.size // 9)))
    return out


_MINV_CACHE_MAX = 256
_MINV_PINNED_MAX = 4096
_minv_cache = collections.OrderedDict()  # (matrix bytes, device, inverse_given) -> device tensor, least recently used first
_minv_pinned = {}                        # entries whose address a hipGraph capture has seen: never evicted


def device_inverse(M, device, inverse_given=False):
    """(n, 3, 3) float64 device tensor of inverse matrices for forward matrices M (cached by value).

    Lifetime rules (launches only ever receive the tensor's raw address): every use records the current stream on the
    tensor, so an evicted entry's memory is not handed out again before the launches that read it have run; an entry
    that is looked up while the current stream is being captured into a graph is pinned for the life of the process
    (the graph replays its address; 72 bytes per matrix, at most _MINV_PINNED_MAX entries -- re-capturing with ever new
    matrices beyond that raises: pass `M_inv_device`), and a cache MISS during capture raises -- upload the matrices before capturing
    (or pass `M_inv_device`, which the caller owns)."""
    if isinstance(M, torch.Tensor) and M.is_cuda and inverse_given:
        return M.to(torch.float64).reshape(-1, 3, 3).contiguous()
    device = torch.device(device)
    Mh = np.ascontiguousarray(M.detach().cpu().numpy() if isinstance(M, torch.Tensor) else M, dtype=np.float64).reshape(-1, 3, 3)
    key = (Mh.tobytes(), str(device), bool(inverse_given))
    capturing = False
    if device.type == "cuda" and torch.cuda.is_available():
        with torch.cuda.device(device):  # the capture state of THIS device's current stream, not of the default device's
            capturing = torch.cuda.is_current_stream_capturing()
    hit = _minv_pinned.get(key)
    if hit is not None:
        return hit
    hit = _minv_cache.get(key)
    if hit is None:
        if capturing:
            raise RuntimeError("device_inverse: homography not resident while a graph is being captured; call the step once "
                               "before capturing, or pass M_inv_device")
        inv = Mh if inverse_given else invert_homography(Mh)
        hit = torch.from_numpy(inv).to(device)
        hit._bevwarp_owned = True  # (cached by value, never written again: its tile verdicts may be cached too, _tile_classes)
        _minv_cache[key] = hit
        while len(_minv_cache) > _MINV_CACHE_MAX:
            _minv_cache.popitem(last=False)  # (safe: every use recorded its stream, see below)
    else:
        _minv_cache.move_to_end(key)
    if capturing:
        if len(_minv_pinned) >= _MINV_PINNED_MAX:
            raise RuntimeError("device_inverse: %d homography sets are already pinned by graph captures; pass M_inv_device (a tensor the "
                               "caller owns) when capturing graphs with ever new matrices" % _MINV_PINNED_MAX)
        _minv_pinned[key] = _minv_cache.pop(key)
    elif hit.is_cuda:
        hit.record_stream(torch.cuda.current_stream(hit.device))
    return hit


try:  # the current stream's raw hipStream_t without building a torch.cuda.Stream object (~0.2 us instead of ~1.5)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover - older / CPU-only builds
    def _raw_stream(dev_index):
        return torch.cuda.current_stream(dev_index).cuda_stream

_PLANS_MAX = 1024
_plans = {}  # validated launches by (addresses, shapes, strides, dtypes, dsize, flags) -> (entry point, bound arguments, device index)


def _check_minv(M_inv_device, device, B):
    """A caller-supplied matrix tensor goes to the kernel as a raw address: it must be exactly what the ABI reads."""
    t = M_inv_device
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float64 or not t.is_contiguous() or t.device != device:
        raise ValueError("M_inv_device must be a contiguous float64 tensor on %s" % (device,))
    if t.dim() < 2 or tuple(t.shape[-2:]) != (3, 3) or t.numel() % 9:
        raise ValueError("M_inv_device must be (n, 3, 3), got %s" % (tuple(t.shape),))
    n_m = t.numel() // 9
    if n_m not in (1, B):
        raise ValueError("got %d homographies for a batch of %d" % (n_m, B))
    return n_m


def _check_out(out, dtype, device, numel):
    """A caller-supplied destination is written through its raw address with the SOURCE's element size."""
    if not isinstance(out, torch.Tensor) or out.dtype != dtype or out.device != device or out.numel() != numel:
        raise ValueError("out must be a %s tensor of %d elements on %s" % (dtype, numel, device))


def _border(border_value, C):
    if border_value is None:
        return None
    return np.ascontiguousarray(np.broadcast_to(np.asarray(border_value, dtype=np.float64), (C,)))


_CLASSES_MAX = 64
_class_tables = collections.OrderedDict()  # (matrix tensor address, n matrices, batch, sizes, format) -> (verdict table, the matrix tensor)


def _tile_classes(M_inv_device, n_m, call_args, stream):
    """The per-tile verdict table (include/bevwarp.h, bevwarp_warp_classes) of a launch whose matrices are owned by device_inverse --
    cached by value and never written again, so verdicts derived from them once hold for every later launch with the same geometry:
    the camera loop of vis_homo.py:85-91 warps every frame of a video through one H_bev_img.  Filled on first use (one launch that
    writes no pixel); None for matrices the caller owns (their contents may change under the same address), while a graph is being
    captured (the fill would allocate), and for launches the library keeps no table for."""
    if not getattr(M_inv_device, "_bevwarp_owned", False):
        return None
    (_, _, B, H, W, dh, dw, C, _, _, _, _, _, _, dtype, interp, _) = call_args
    key = (M_inv_device.data_ptr(), n_m, B, H, W, dh, dw, C, dtype, interp)
    with torch.cuda.device(M_inv_device.device):
        if torch.cuda.is_current_stream_capturing():  # (a graph would replay the table's raw address: captured launches classify for themselves)
            return None
        hit = _class_tables.get(key)
        if hit is not None:
            _class_tables.move_to_end(key)
            return hit[0]
        lib = _lib.load()
        nbytes = lib.bevwarp_tile_classes_bytes(B, H, W, dh, dw, C, dtype, interp)
        if nbytes <= 0:
            return None
        table = torch.zeros(nbytes // 4, dtype=torch.int32, device=M_inv_device.device)
        _lib.check(lib.bevwarp_warp_classes(*call_args, table.data_ptr(), 1, ctypes.c_void_p(stream)))
    _class_tables[key] = (table, M_inv_device)  # (holding the matrix tensor keeps its address from being handed out again)
    while len(_class_tables) > _CLASSES_MAX:
        _class_tables.popitem(last=False)
    return table


def warp_perspective(src, M, dsize, flags=INTER_LINEAR, border_value=None, out=None, M_inv_device=None):
    """Batched perspective warp on the GPU.

    src      (B, H, W, C) or (H, W, C) or (H, W) uint8 / float32 CUDA tensor, channels-last, rows contiguous.
    M        (3, 3) shared or (B, 3, 3) per-frame forward homography (numpy or tensor); with
             flags | WARP_INVERSE_MAP it is taken as the dst -> src map instead.
    dsize    (width, height) = (u_size, v_size), as OpenCV.
    flags    INTER_LINEAR (default) or INTER_NEAREST, optionally | WARP_INVERSE_MAP.
    out      optional preallocated result; M_inv_device optional (n, 3, 3) f64 CUDA tensor to skip the cache.
    Returns a tensor shaped like src with (height, width) replaced.  Asynchronous on the current stream."""
    if out is not None and M_inv_device is not None and border_value is None:
        # Steady-state call of a camera loop: same buffers, same geometry as a call that has already been validated.  The
        # whole Python layer is then one dictionary lookup and the C call with its argument tuple bound (a single 720p -> 512^2
        # frame is a 10-us kernel: the general path below costs more host time than that).
        try:
            # (addresses can be recycled by the allocator: everything the slow path validates is part of the key -- dtypes and
            # devices of all three tensors included, or a float32 / CPU matrix tensor at a recycled address would skip _check_minv)
            key = (src.data_ptr(), out.data_ptr(), M_inv_device.data_ptr(), src.shape, src.stride(), out.shape, out.stride(), M_inv_device.shape,
                   M_inv_device.stride(), src.dtype, out.dtype, M_inv_device.dtype, src.device, out.device, M_inv_device.device, dsize[0], dsize[1], flags)
            plan = _plans.get(key)
        except (AttributeError, TypeError, IndexError):
            plan = None
        if plan is not None:
            fn, args, dev_index = plan[:3]
            if torch.cuda.current_device() == dev_index:
                st = fn(*args, _raw_stream(dev_index))
            else:
                with torch.cuda.device(dev_index):
                    st = fn(*args, _raw_stream(dev_index))
            if st:
                _lib.check(st)
            return out
    else:
        key = None
    if not isinstance(src, torch.Tensor) or not src.is_cuda:
        raise ValueError("warp_perspective needs a CUDA (HIP) tensor; use warpPerspective for numpy images")
    if src.dtype not in _DTYPES:
        raise ValueError("unsupported dtype %s (uint8 / float32)" % src.dtype)
    interp = int(flags) & 7
    if interp not in (INTER_NEAREST, INTER_LINEAR):
        raise ValueError("unsupported interpolation flag %d" % interp)
    shape = tuple(src.shape)
    s4 = src
    if src.dim() == 2:
        s4 = src[None, :, :, None]
    elif src.dim() == 3:
        s4 = src[None]
    elif src.dim() != 4:
        raise ValueError("src must be (B,H,W,C), (H,W,C) or (H,W)")
    B, H, W, C = s4.shape
    copied = s4.stride(3) != 1 or s4.stride(2) != C
    if copied:
        s4 = s4.contiguous()
    dw, dh = int(dsize[0]), int(dsize[1])
    esz = s4.element_size()
    if M_inv_device is None:
        M_inv_device = device_inverse(M, s4.device, inverse_given=bool(int(flags) & WARP_INVERSE_MAP))
    n_m = _check_minv(M_inv_device, s4.device, B)
    if out is None:
        d4 = torch.empty((B, dh, dw, C), dtype=s4.dtype, device=s4.device)
    else:
        _check_out(out, s4.dtype, s4.device, B * dh * dw * C)
        d4 = out.reshape(B, dh, dw, C)
        if d4.data_ptr() != out.data_ptr() or d4.stride(3) != 1 or d4.stride(2) != C:
            raise ValueError("out must be a contiguous-row channels-last tensor")
    bv = _border(border_value, C)
    stream = torch.cuda.current_stream(s4.device).cuda_stream
    args = (s4.data_ptr(), d4.data_ptr(), B, H, W, dh, dw, C, s4.stride(0) * esz, s4.stride(1) * esz, d4.stride(0) * esz, d4.stride(1) * esz,
            M_inv_device.data_ptr(), n_m, _DTYPES[s4.dtype], interp, None if bv is None else bv.ctypes.data_as(ctypes.c_void_p))
    fn = _lib.load().bevwarp_warp
    table = _tile_classes(M_inv_device, n_m, args, stream)
    if table is not None:  # verdicts of these very matrices and this geometry: the kernel reads them instead of deriving them
        table.record_stream(torch.cuda.current_stream(s4.device))
        fn, args = _lib.load().bevwarp_warp_classes, args + (table.data_ptr(), 0)
    with torch.cuda.device(s4.device):
        st = fn(*args, ctypes.c_void_p(stream))
    _lib.check(st)
    if key is not None and not copied:  # validated and launched: the next call with these very buffers skips the checks
        if len(_plans) >= _PLANS_MAX:
            _plans.clear()
        _plans[key] = (fn, args, s4.device.index if s4.device.index is not None else torch.cuda.current_device(), table,
                       M_inv_device if table is not None else None)  # (a plan with a verdict table keeps the table AND the matrices it belongs to alive:
        #                                                          their address must not be handed out again while the plan can be hit)
    if out is not None:
        return out
    if len(shape) == 2:
        return d4[0, :, :, 0]
    if len(shape) == 3:
        return d4[0]
    return d4


def warp_to_planar(src, M, dsize, scale=1.0 / 255.0, bias=0.0, flags=INTER_LINEAR, border_value=None, out=None, M_inv_device=None):
    """Warp uint8 (or float32) frames and write them as normalised float32 channel planes in the same pass (SURVEY.md 8(f2):
    the layout a detector takes, `(B, C, v_size, u_size)`), without materialising the interleaved BEV frame:

        out[b, c] = warp_perspective(src, M, dsize)[b, :, :, c].float() * scale[c] + bias[c]      (float32 mul, then add)

    src (B, H, W, C) / (H, W, C) / (H, W) uint8 or float32 CUDA tensor; scale / bias scalars or per-channel sequences (e.g.
    1 / (255 * std) and -mean / std); other arguments as warp_perspective.  Returns (B, C, h, w), or (C, h, w) for a
    single frame.  Asynchronous on the current stream."""
    if not isinstance(src, torch.Tensor) or not src.is_cuda or src.dtype not in _DTYPES:
        raise ValueError("warp_to_planar needs a uint8 or float32 CUDA (HIP) tensor")
    interp = int(flags) & 7
    if interp not in (INTER_NEAREST, INTER_LINEAR):
        raise ValueError("unsupported interpolation flag %d" % interp)
    if src.dim() == 2:
        s4 = src[None, :, :, None]
    elif src.dim() == 3:
        s4 = src[None]
    elif src.dim() == 4:
        s4 = src
    else:
        raise ValueError("src must be (B,H,W,C), (H,W,C) or (H,W)")
    B, H, W, C = s4.shape
    if s4.stride(3) != 1 or s4.stride(2) != C:
        s4 = s4.contiguous()
    dw, dh = int(dsize[0]), int(dsize[1])
    if M_inv_device is None:
        M_inv_device = device_inverse(M, s4.device, inverse_given=bool(int(flags) & WARP_INVERSE_MAP))
    n_m = _check_minv(M_inv_device, s4.device, B)
    if out is None:
        d4 = torch.empty((B, C, dh, dw), dtype=torch.float32, device=s4.device)
    else:
        _check_out(out, torch.float32, s4.device, B * C * dh * dw)
        d4 = out.reshape(B, C, dh, dw)
        if d4.data_ptr() != out.data_ptr() or d4.stride(3) != 1:
            raise ValueError("out must be a float32 (B, C, h, w) tensor with contiguous rows")
    def per_channel(v):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64), (C,)))
    sc, bi = per_channel(scale), per_channel(bias)
    bv = None if border_value is None else per_channel(border_value)
    stream = torch.cuda.current_stream(s4.device).cuda_stream
    with torch.cuda.device(s4.device):
        st = _lib.load().bevwarp_warp_planar(
            s4.data_ptr(), d4.data_ptr(), B, H, W, dh, dw, C, s4.stride(0) * s4.element_size(), s4.stride(1) * s4.element_size(),
            d4.stride(0) * 4, d4.stride(1) * 4, d4.stride(2) * 4, M_inv_device.data_ptr(), n_m, _DTYPES[s4.dtype], interp,
            None if bv is None else bv.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p),
            bi.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(stream))
    _lib.check(st)
    if out is not None:
        return out
    return d4 if src.dim() == 4 else d4[0]


def footprint(src_hw, M, dsize, batch=None, flags=INTER_LINEAR, device="cuda"):
    """Exact count of distinct in-bounds source pixels the warp reads, per frame.
    Returns (counts int64 tensor [n], touched uint8 tensor [n, H, W])."""
    H, W = int(src_hw[0]), int(src_hw[1])
    minv = device_inverse(M, torch.device(device), inverse_given=bool(int(flags) & WARP_INVERSE_MAP))
    n = minv.shape[0] if batch is None else int(batch)
    touched = torch.zeros((n, H, W), dtype=torch.uint8, device=minv.device)
    stream = torch.cuda.current_stream(minv.device).cuda_stream
    with torch.cuda.device(minv.device):
        st = _lib.load().bevwarp_footprint(touched.data_ptr(), n, H, W, int(dsize[1]), int(dsize[0]), minv.data_ptr(),
                                           minv.shape[0], int(flags) & 7, ctypes.c_void_p(stream))
    _lib.check(st)
    return touched.reshape(n, -1).sum(dim=1, dtype=torch.int64), touched


def scalar_border(borderValue, channels):
    """cv2 turns `borderValue` into a cv::Scalar: a bare number v means (v, 0, 0, 0) -- channel 0 only -- and a sequence
    fills the leading channels, the rest stay 0."""
    v = np.atleast_1d(np.asarray(borderValue, dtype=np.float64)).ravel()[:4]
    out = np.zeros(channels, dtype=np.float64)
    out[:min(len(v), channels)] = v[:channels]
    return out


def warpPerspective(src, M, dsize, dst=None, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0, device="cuda"):
    """cv2.warpPerspective call shape for numpy images: uploads, warps on the GPU, downloads.
    (The per-frame PCIe round trip dominates here; batch frames with warp_perspective for throughput.)"""
    if borderMode != BORDER_CONSTANT:
        raise ValueError("only BORDER_CONSTANT is implemented (the reference never passes another mode)")
    img = np.asarray(src)
    if img.dtype not in (np.uint8, np.float32):
        raise ValueError("unsupported dtype %s (uint8 / float32)" % img.dtype)
    t = torch.from_numpy(np.ascontiguousarray(img)).to(device)
    bv = scalar_border(borderValue, 1 if img.ndim == 2 else img.shape[2])
    res = warp_perspective(t, np.asarray(M, dtype=np.float64), dsize, flags=flags, border_value=bv).cpu().numpy()
    if dst is not None:
        dst[...] = res
        return dst
    return res


def resize_matrix(src_wh, new_wh, align_corners=False):
    """3x3 map from pixels of a (w, h) image to pixels of its resized (new_w, new_h) version, in the convention of
    the reference's Calib.scale (bev/calib.py:142-198): x' = (x + 0.5) * r - 0.5 (align_corners=False, what
    cv2.resize does) or x' = x * (new - 1) / (old - 1)."""
    (w, h), (nw, nh) = src_wh, new_wh
    if align_corners:
        ru, rv = (nw - 1) / (w - 1), (nh - 1) / (h - 1)
        return np.array([[ru, 0, 0], [0, rv, 0], [0, 0, 1.0]])
    ru, rv = nw / w, nh / h
    return np.array([[ru, 0, 0.5 * ru - 0.5], [0, rv, 0.5 * rv - 0.5], [0, 0, 1.0]])


def warp_perspective_resized(src, M_resized, dsize, new_wh, align_corners=False, **kw):
    """The "small" branch of vis_homo.py:73-78,90-91 -- cv2.resize(img, new_wh) followed by
    cv2.warpPerspective(img_small, H_bev_img_small, dsize) -- with the resize folded into the homography: the
    full-resolution frame is sampled once through M_resized @ resize_matrix(...), no intermediate image exists.
    `M_resized` maps pixels of the RESIZED image to the destination (H_bev_img_small).  The result is the warp of the
    full-resolution frame (sharper than the two-step path, which low-passes through the resize); it is bit-identical
    to warp_perspective(src, M_resized @ S, dsize).  Keyword arguments as for warp_perspective."""
    h, w = (src.shape[-3], src.shape[-2]) if src.dim() >= 3 else src.shape
    S = resize_matrix((w, h), new_wh, align_corners)
    return warp_perspective(src, np.asarray(M_resized, dtype=np.float64) @ S, dsize, **kw)

