"""Batched tracker geometry as ONE device launch.

The reference's per-frame tracking loop (bev/tool/rbox_tracking_BrnoCompSpeed.py:88-109 with the association front-end of
bev/tracker/rbox_tracker.py:383-405) moves detections from the BEV raster to the world plane (`rbox_world_bev`), scores
them against the trackers' predicted boxes (`iou_batch_rbox` -> d3d), thresholds the scores, and projects box centres into
the image (`rbox_world_img`): tens of numpy calls per frame.  Here that is `bevwarp_tracker_step` -- one HIP launch on
device tensors, no host synchronisation, nothing computed by torch ops.  The Hungarian assignment and the per-track Kalman
filters stay on the host (out of scope)."""
import ctypes

import numpy as np
import torch

from . import _lib

_DTYPES = {torch.float32: _lib.F32, torch.float64: _lib.F64}


def _host9(H):
    return np.ascontiguousarray(np.asarray(H, dtype=np.float64).reshape(3, 3))


def _dev(x, device):
    return x.to(device) if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x)).to(device)


def _boxes(t, what):
    if t.dim() != 2 or t.shape[1] < 5 or t.dtype not in _DTYPES:
        raise ValueError("%s must be an (n, >=5) float32 / float64 tensor, got %s %s" % (what, tuple(t.shape), t.dtype))
    return t.contiguous()


def rbox_world_bev_device(rbox_src, H, src):
    """(n, >=5) CUDA tensor of [x, y, w, h, yaw] between the BEV ("bev": yaw = atan2(u, v)) and the world ("world":
    yaw = atan2(y, x)) through the similarity H -- rbox.py:173-219 / rbox_torch.py:123-168 in one launch
    (bevwarp_rbox_transform).  Raises ValueError when H is not a similarity (the reference's assertions).  Returns (n, 5)."""
    assert src in ("bev", "world")
    if not isinstance(rbox_src, torch.Tensor) or not rbox_src.is_cuda:
        raise ValueError("rbox_world_bev_device needs a CUDA (HIP) tensor; bev.rbox.rbox_world_bev is the host version")
    b = _boxes(rbox_src, "rbox_src")
    out = torch.empty((b.shape[0], 5), dtype=b.dtype, device=b.device)
    Hh = _host9(H)
    stream = torch.cuda.current_stream(b.device).cuda_stream
    with torch.cuda.device(b.device):
        st = _lib.load().bevwarp_rbox_transform(b.data_ptr(), b.shape[0], b.shape[1], Hh.ctypes.data_as(ctypes.c_void_p), int(src == "bev"),
                                                out.data_ptr(), _DTYPES[b.dtype], ctypes.c_void_p(stream))
    _lib.check(st)
    return out


def _check_out_dict(out, n, m, dtype, device, want_img):
    """A reused result dict goes to the kernel as raw addresses: every tensor must be exactly what the launch writes."""
    need = {"dets_world": ((n, 5), dtype), "iou": ((n, m), dtype), "candidates": ((n, m), torch.bool)}
    if want_img:
        need["dets_img"] = ((n, 2), dtype)
    for key, (shape, dt) in need.items():
        t = out.get(key) if isinstance(out, dict) else None
        if not isinstance(t, torch.Tensor) or tuple(t.shape) != shape or t.dtype != dt or t.device != device or not t.is_contiguous():
            raise ValueError("out[%r] must be a contiguous %s tensor of shape %s on %s (reuse only the dict of a call with the same "
                             "n, m, dtype and H_img_world)" % (key, dt, shape, device))


_PLANS_MAX = 256
_plans = {}  # validated launches of tracker_geometry_step by (addresses, shapes, strides, dtypes, devices, matrix bytes, threshold)


def _raw_stream(dev_index):
    try:
        return torch._C._cuda_getCurrentRawStream(dev_index)
    except AttributeError:  # pragma: no cover
        return torch.cuda.current_stream(dev_index).cuda_stream


def tracker_geometry_step(dets_bev, trks_world, H_world_bev, iou_threshold=0.3, H_img_world=None, device="cuda", out=None):
    """dets_bev (n, >=5) detections in BEV pixels, trks_world (m, >=5) predicted tracker boxes in the world (numpy or
    tensors; float64 unless both are float32 tensors).  Returns a dict of device tensors: dets_world (n, 5), iou (n, m),
    candidates (n, m) bool = iou > iou_threshold (the gate of rbox_tracker.py:395-405), and, when H_img_world is given,
    dets_img (n, 2) image pixels of the box centres (rbox_world_img, rbox.py:221-226).  One launch, no host synchronisation.
    `out`: a dict returned by an earlier call with the same shapes, to reuse its tensors (e.g. inside a captured graph)."""
    key = None
    if out is not None and isinstance(dets_bev, torch.Tensor) and isinstance(trks_world, torch.Tensor):
        # Steady state of a camera loop: the same device buffers as a call that has already been validated (the launch is ~10 us: the
        # general path below costs more host time than that).  Everything the slow path checks is in the key; the matrices by value.
        try:
            key = (dets_bev.data_ptr(), trks_world.data_ptr(), dets_bev.shape, trks_world.shape, dets_bev.stride(), trks_world.stride(), dets_bev.dtype,
                   trks_world.dtype, dets_bev.device, trks_world.device, np.asarray(H_world_bev).tobytes(),
                   None if H_img_world is None else np.asarray(H_img_world).tobytes(), float(iou_threshold),
                   tuple((k, v.data_ptr(), v.shape, v.dtype, v.device) for k, v in sorted(out.items())))
            plan = _plans.get(key)
        except (AttributeError, TypeError):
            key, plan = None, None
        if plan is not None:
            fn, args, dev_index, _keep = plan
            if torch.cuda.current_device() == dev_index:
                st = fn(*args, _raw_stream(dev_index))
            else:
                with torch.cuda.device(dev_index):
                    st = fn(*args, _raw_stream(dev_index))
            if st:
                _lib.check(st)
            return out
    d, t = _dev(dets_bev, device), _dev(trks_world, device)
    if d.dtype != t.dtype:
        d, t = d.to(torch.float64), t.to(torch.float64)
    d, t = _boxes(d, "dets_bev"), _boxes(t, "trks_world")
    n, m = d.shape[0], t.shape[0]
    if out is not None:
        _check_out_dict(out, n, m, d.dtype, d.device, H_img_world is not None)
    else:
        out = {"dets_world": torch.empty((n, 5), dtype=d.dtype, device=d.device), "iou": torch.empty((n, m), dtype=d.dtype, device=d.device),
               "candidates": torch.empty((n, m), dtype=torch.bool, device=d.device)}
        if H_img_world is not None:
            out["dets_img"] = torch.empty((n, 2), dtype=d.dtype, device=d.device)
    Hwb = _host9(H_world_bev)
    Him = None if H_img_world is None else _host9(H_img_world)
    stream = torch.cuda.current_stream(d.device).cuda_stream
    fn = _lib.load().bevwarp_tracker_step
    args = (d.data_ptr(), n, d.shape[1], t.data_ptr(), m, t.shape[1], Hwb.ctypes.data_as(ctypes.c_void_p),
            None if Him is None else Him.ctypes.data_as(ctypes.c_void_p), float(iou_threshold), out["dets_world"].data_ptr(),
            out["iou"].data_ptr(), out["candidates"].data_ptr(), out["dets_img"].data_ptr() if Him is not None else None, _DTYPES[d.dtype])
    with torch.cuda.device(d.device):
        st = fn(*args, ctypes.c_void_p(stream))
    _lib.check(st)
    if key is not None and d.data_ptr() == dets_bev.data_ptr() and t.data_ptr() == trks_world.data_ptr():  # (no copy was made on the way)
        if len(_plans) >= _PLANS_MAX:
            _plans.clear()
        _plans[key] = (fn, args, d.device.index if d.device.index is not None else torch.cuda.current_device(), (Hwb, Him))  # (the host matrices stay alive)
    return out
