"""Frame ingest / egress around the warp: decoded HWC uint8 frames on the host -> BEV frames, with the
PCIe transfers overlapped with the kernel.

The reference's loop is `ok, img = video.read(); bev = cv2.warpPerspective(img, H, size); writer.write(bev)`
(vis_homo.py:85-111, bev/io/utils.py:124-149): decode, warp and encode strictly one after the other on the host.  Here a
frame travels through three stages that run concurrently on three HIP streams:

    pinned host slot --H2D (copy stream)--> device frame --warp (compute stream)--> device BEV --D2H (copy-back stream)--> pinned host slot

with `depth` slots per stage (default 3), so frame i + 1 uploads and frame i - 1 downloads while frame i is warped; the
link is full duplex.  The decoder can write straight into the next pinned slot (`next_input()`), which removes the host
memcpy a pageable `submit(frame)` needs.  Output is the interleaved BEV frame of the source dtype, or -- `planar=True` --
normalised float32 channel planes (the layout a detector takes), in the same pass.  Results are the resident path's, bit
for bit: the same kernel runs on the same bytes.
"""
import collections
import ctypes

import numpy as np
import torch

from . import _lib
from . import warp as _warp

_hip = None


def _hip_rt():
    """The HIP runtime torch already loaded, bound through ctypes: a frame's eight asynchronous calls (two copies, the
    launch, the events that chain the three streams) cost ~2 us each this way; through torch's stream context managers they
    cost more host time than the transfers take, and the pipeline would be bound by the Python loop."""
    global _hip
    if _hip is None:
        h = ctypes.CDLL("libamdhip64.so")
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        for name, args in (("hipMemcpyAsync", [vp, vp, sz, ctypes.c_int, vp]), ("hipEventCreateWithFlags", [ctypes.POINTER(vp), ctypes.c_uint]),
                           ("hipEventRecord", [vp, vp]), ("hipStreamWaitEvent", [vp, vp, ctypes.c_uint]), ("hipEventSynchronize", [vp]),
                           ("hipEventDestroy", [vp])):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = ctypes.c_int, args
        _hip = h
    return _hip


def _ok(err):
    if err != 0:
        raise _lib.BevWarpError("HIP runtime call failed with error %d" % err)


_H2D, _D2H = 1, 2  # hipMemcpyHostToDevice / hipMemcpyDeviceToHost


class FramePipeline:
    def __init__(self, src_hw, channels, M, dsize, flags=_warp.INTER_LINEAR, depth=3, dtype=torch.uint8, planar=False, scale=1.0 / 255.0, bias=0.0,
                 download=True, device="cuda", zero_copy_out=True):
        """src_hw (H, W) of the decoded frames; M the forward homography (as for warpPerspective); dsize (u_size, v_size).
        download=False leaves the BEV frames on the device (results are device tensors valid until `depth` further frames
        have been submitted).  zero_copy_out (with download): the kernel stores the BEV frame straight into the pinned host
        slot over PCIe -- no device copy of it, no D2H copy behind the kernel (on boxes where the two copy directions share an
        engine that copy serialises with the next frame's upload)."""
        if depth < 2:
            raise ValueError("depth must be >= 2 (one slot in flight per stage boundary)")
        self.device = torch.device(device)
        self.H, self.W, self.C = int(src_hw[0]), int(src_hw[1]), int(channels)
        self.dw, self.dh = int(dsize[0]), int(dsize[1])
        self.flags, self.depth, self.planar, self.scale, self.bias, self.download = flags, depth, planar, scale, bias, download
        self.zero_copy_out = bool(zero_copy_out and download)
        out_shape = (self.C, self.dh, self.dw) if planar else (self.dh, self.dw, self.C)
        out_dtype = torch.float32 if planar else dtype
        self.h_in = [torch.empty((self.H, self.W, self.C), dtype=dtype, pin_memory=True) for _ in range(depth)]
        self.d_in = [torch.empty((self.H, self.W, self.C), dtype=dtype, device=self.device) for _ in range(depth)]
        self.d_out = [torch.empty(out_shape, dtype=out_dtype, device=self.device) for _ in range(depth)]
        self.h_out = [torch.empty(out_shape, dtype=out_dtype, pin_memory=True) for _ in range(depth)] if download else None
        self.minv = _warp.device_inverse(M, self.device, inverse_given=bool(int(flags) & _warp.WARP_INVERSE_MAP))
        self._streams = [torch.cuda.Stream(self.device) for _ in range(3)]  # (kept alive: their raw handles are used below)
        self.s_up, self.s_run, self.s_down = (ctypes.c_void_p(st.cuda_stream) for st in self._streams)
        hip = _hip_rt()

        def events():
            evs = []
            for _ in range(depth):
                e = ctypes.c_void_p()
                _ok(hip.hipEventCreateWithFlags(ctypes.byref(e), 2))  # hipEventDisableTiming
                evs.append(e)
            return evs

        self.ev_up, self.ev_run, self.ev_down = events(), events(), events()
        # one prepared launch per slot: the C ABI call with its arguments bound (validated once, here, through the Python layer)
        for slot in range(depth):
            self._launch_py(slot, torch.cuda.current_stream(self.device))
        torch.cuda.synchronize(self.device)
        interp = int(flags) & 7
        esz = self.d_in[0].element_size()
        lib = _lib.load()
        self._launch = []
        for slot in range(depth):
            s, d = self.d_in[slot], (self.h_out[slot] if self.zero_copy_out else self.d_out[slot])  # (pinned host memory is device-addressable)
            if planar:
                sc = np.ascontiguousarray(np.broadcast_to(np.asarray(scale, dtype=np.float64), (self.C,)))
                bi = np.ascontiguousarray(np.broadcast_to(np.asarray(bias, dtype=np.float64), (self.C,)))
                self._keep = getattr(self, "_keep", []) + [sc, bi]
                args = (s.data_ptr(), d.data_ptr(), 1, self.H, self.W, self.dh, self.dw, self.C, s.numel() * esz, s.stride(0) * esz, d.numel() * 4,
                        d.stride(0) * 4, d.stride(1) * 4, self.minv.data_ptr(), 1, _warp._DTYPES[s.dtype], interp, None,
                        sc.ctypes.data_as(ctypes.c_void_p), bi.ctypes.data_as(ctypes.c_void_p), self.s_run)
                self._launch.append((lib.bevwarp_warp_planar, args))
            else:
                args = (s.data_ptr(), d.data_ptr(), 1, self.H, self.W, self.dh, self.dw, self.C, s.numel() * esz, s.stride(0) * esz, d.numel() * esz,
                        d.stride(0) * esz, self.minv.data_ptr(), 1, _warp._DTYPES[s.dtype], interp, None, self.s_run)
                self._launch.append((lib.bevwarp_warp, args))
        self._in_bytes = self.h_in[0].numel() * esz
        self._out_bytes = self.d_out[0].numel() * self.d_out[0].element_size()
        self.n_in = 0
        self.pending = collections.deque()  # slots whose results have not been handed out yet
        self._closed = False

    def close(self):
        """Drain the three private streams, then release the events.  The device / pinned buffers were only ever handed to those
        streams as raw addresses (torch does not know they are in use there), so they must not be freed before this."""
        if getattr(self, "_closed", True):
            return
        self._closed = True
        try:
            for st in self._streams:
                st.synchronize()
            hip = _hip_rt()
            for e in self.ev_up + self.ev_run + self.ev_down:
                hip.hipEventDestroy(e)
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    def _launch_py(self, slot, stream):
        """The same launch through bev_amd.warp (argument validation; used once per slot at construction)."""
        with torch.cuda.stream(stream):
            if self.planar:
                _warp.warp_to_planar(self.d_in[slot], None, (self.dw, self.dh), scale=self.scale, bias=self.bias, flags=self.flags, out=self.d_out[slot],
                                     M_inv_device=self.minv)
            else:
                _warp.warp_perspective(self.d_in[slot], None, (self.dw, self.dh), flags=self.flags, out=self.d_out[slot], M_inv_device=self.minv)

    # -- input side
    def next_input(self):
        """The pinned host buffer (numpy view, HWC) the NEXT frame should be decoded into; call commit() when it is filled.
        Blocks only if that slot's previous frame has not left the device yet."""
        slot = self.n_in % self.depth
        if self.n_in >= self.depth:
            _ok(_hip_rt().hipEventSynchronize((self.ev_down if self.download and not self.zero_copy_out else self.ev_run)[slot]))  # its previous occupant is through
        return self.h_in[slot].numpy()

    def commit(self):
        if len(self.pending) >= self.depth:  # the slot about to be reused still holds a result nobody has taken
            raise RuntimeError("FramePipeline: %d frames are in flight and none has been taken with result(); a ring of depth %d "
                               "holds at most %d undelivered frames" % (len(self.pending), self.depth, self.depth))
        hip = _hip_rt()
        slot = self.n_in % self.depth
        self.n_in += 1
        _ok(hip.hipMemcpyAsync(self.d_in[slot].data_ptr(), self.h_in[slot].data_ptr(), self._in_bytes, _H2D, self.s_up))
        _ok(hip.hipEventRecord(self.ev_up[slot], self.s_up))
        _ok(hip.hipStreamWaitEvent(self.s_run, self.ev_up[slot], 0))
        fn, args = self._launch[slot]
        _lib.check(fn(*args))
        _ok(hip.hipEventRecord(self.ev_run[slot], self.s_run))
        if self.download and not self.zero_copy_out:
            _ok(hip.hipStreamWaitEvent(self.s_down, self.ev_run[slot], 0))
            _ok(hip.hipMemcpyAsync(self.h_out[slot].data_ptr(), self.d_out[slot].data_ptr(), self._out_bytes, _D2H, self.s_down))
            _ok(hip.hipEventRecord(self.ev_down[slot], self.s_down))
        self.pending.append(slot)

    def submit(self, frame):
        """A decoded frame from pageable memory (numpy HWC): copied into the next pinned slot, then committed."""
        buf = self.next_input()
        np.copyto(buf, np.asarray(frame).reshape(buf.shape))
        self.commit()

    # -- output side
    def ready(self):
        return len(self.pending)

    def result(self):
        """The oldest frame's BEV: a numpy view of a pinned slot (download=True; valid until `depth` further frames have been
        committed) or the device tensor.  Blocks until that frame is through."""
        slot = self.pending.popleft()
        if self.download:
            _ok(_hip_rt().hipEventSynchronize((self.ev_run if self.zero_copy_out else self.ev_down)[slot]))
            return self.h_out[slot].numpy()
        _ok(_hip_rt().hipEventSynchronize(self.ev_run[slot]))
        return self.d_out[slot]

    def run(self, frames):
        """Generator over an iterable of decoded frames: yields their BEV frames in order, keeping depth - 1 frames in flight."""
        for f in frames:
            if len(self.pending) >= self.depth - 1:
                yield self.result()
            self.submit(f)
        while self.pending:
            yield self.result()
