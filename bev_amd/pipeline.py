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

import numpy as np
import torch

from . import warp as _warp


class FramePipeline:
    def __init__(self, src_hw, channels, M, dsize, flags=_warp.INTER_LINEAR, depth=3, dtype=torch.uint8, planar=False, scale=1.0 / 255.0, bias=0.0,
                 download=True, device="cuda"):
        """src_hw (H, W) of the decoded frames; M the forward homography (as for warpPerspective); dsize (u_size, v_size).
        download=False leaves the BEV frames on the device (results are device tensors valid until `depth` further frames
        have been submitted)."""
        if depth < 2:
            raise ValueError("depth must be >= 2 (one slot in flight per stage boundary)")
        self.device = torch.device(device)
        self.H, self.W, self.C = int(src_hw[0]), int(src_hw[1]), int(channels)
        self.dw, self.dh = int(dsize[0]), int(dsize[1])
        self.flags, self.depth, self.planar, self.scale, self.bias, self.download = flags, depth, planar, scale, bias, download
        out_shape = (self.C, self.dh, self.dw) if planar else (self.dh, self.dw, self.C)
        out_dtype = torch.float32 if planar else dtype
        self.h_in = [torch.empty((self.H, self.W, self.C), dtype=dtype, pin_memory=True) for _ in range(depth)]
        self.d_in = [torch.empty((self.H, self.W, self.C), dtype=dtype, device=self.device) for _ in range(depth)]
        self.d_out = [torch.empty(out_shape, dtype=out_dtype, device=self.device) for _ in range(depth)]
        self.h_out = [torch.empty(out_shape, dtype=out_dtype, pin_memory=True) for _ in range(depth)] if download else None
        self.minv = _warp.device_inverse(M, self.device, inverse_given=bool(int(flags) & _warp.WARP_INVERSE_MAP))
        self.s_up, self.s_run, self.s_down = (torch.cuda.Stream(self.device) for _ in range(3))
        self.ev_up = [torch.cuda.Event() for _ in range(depth)]
        self.ev_run = [torch.cuda.Event() for _ in range(depth)]
        self.ev_down = [torch.cuda.Event() for _ in range(depth)]
        self.n_in = 0
        self.pending = collections.deque()  # slots whose results have not been handed out yet

    # -- input side
    def next_input(self):
        """The pinned host buffer (numpy view, HWC) the NEXT frame should be decoded into; call commit() when it is filled.
        Blocks only if that slot's previous frame has not left the device yet."""
        slot = self.n_in % self.depth
        if self.n_in >= self.depth:
            (self.ev_down if self.download else self.ev_run)[slot].synchronize()  # its previous occupant is through
        return self.h_in[slot].numpy()

    def commit(self):
        slot = self.n_in % self.depth
        self.n_in += 1
        with torch.cuda.stream(self.s_up):
            self.d_in[slot].copy_(self.h_in[slot], non_blocking=True)
            self.ev_up[slot].record(self.s_up)
        with torch.cuda.stream(self.s_run):
            self.s_run.wait_event(self.ev_up[slot])
            if self.planar:
                _warp.warp_to_planar(self.d_in[slot], None, (self.dw, self.dh), scale=self.scale, bias=self.bias, flags=self.flags, out=self.d_out[slot],
                                     M_inv_device=self.minv)
            else:
                _warp.warp_perspective(self.d_in[slot], None, (self.dw, self.dh), flags=self.flags, out=self.d_out[slot], M_inv_device=self.minv)
            self.ev_run[slot].record(self.s_run)
        if self.download:
            with torch.cuda.stream(self.s_down):
                self.s_down.wait_event(self.ev_run[slot])
                self.h_out[slot].copy_(self.d_out[slot], non_blocking=True)
                self.ev_down[slot].record(self.s_down)
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
            self.ev_down[slot].synchronize()
            return self.h_out[slot].numpy()
        self.ev_run[slot].synchronize()
        return self.d_out[slot]

    def run(self, frames):
        """Generator over an iterable of decoded frames: yields their BEV frames in order, keeping depth - 1 frames in flight."""
        for f in frames:
            if len(self.pending) >= self.depth - 1:
                yield self.result()
            self.submit(f)
        while self.pending:
            yield self.result()
