#!/usr/bin/env python3
"""bevwarp_warp against bevwarp_warp_classes (verdict table filled once) on one library, launches interleaved (GPU box):
    python tools/ab_classes.py [--dtype u8|f32] [--interp linear|nearest] [--homography keystone|brno|inset] [--src W H] [--dst W H]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--dtype", default="u8")
    p.add_argument("--interp", default="linear")
    p.add_argument("--homography", default="keystone")
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--src", type=int, nargs=2, default=[1920, 1080])
    p.add_argument("--dst", type=int, nargs=2, default=[1024, 1024])
    p.add_argument("--rounds", type=int, default=40)
    args = p.parse_args()
    from bev_amd import _lib, warp
    from tests import workloads as wl
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    B, (sw, sh), (dw, dh), C = args.batch, args.src, args.dst, 3
    tdt, ndt, esz, dt = (torch.uint8, np.uint8, 1, 0) if args.dtype == "u8" else (torch.float32, np.float32, 4, 1)
    interp = 1 if args.interp == "linear" else 0
    base = {"keystone": wl.keystone_H, "inset": wl.keystone_inset_H, "brno": wl.synth_brno_H}[args.homography](sw, sh, dw, dh)
    minv = warp.device_inverse(np.stack([wl.jitter_H(base, g) for g in range(B)]), dev)
    nsets = max(2, int(np.ceil(1.1e9 / (B * (sh * sw + dh * dw) * C * esz))))
    f0 = torch.stack([torch.from_numpy(wl.frame(g, sh, sw, ndt, C)) for g in range(min(B, 4))]).to(dev)
    srcs, dsts = [], []
    for s in range(nsets):
        t = torch.empty((B, sh, sw, C), dtype=tdt, device=dev)
        for i in range(B):
            t[i] = f0[(i + s) % f0.shape[0]] if (i + s) % 3 == 0 else f0[(i + s) % f0.shape[0]].flip(i % 2)
        srcs.append(t)
        dsts.append(torch.empty((B, dh, dw, C), dtype=tdt, device=dev))
    nbytes = lib.bevwarp_tile_classes_bytes(B, sh, sw, dh, dw, C, dt, interp)
    table = torch.zeros(nbytes // 4, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def geom(k):
        s, d = srcs[k], dsts[k]
        return (s.data_ptr(), d.data_ptr(), B, sh, sw, dh, dw, C, s.stride(0) * esz, s.stride(1) * esz, d.stride(0) * esz, d.stride(1) * esz, minv.data_ptr(), B, dt, interp, None)

    def launch(mode, k):
        if mode == "plain":
            st = lib.bevwarp_warp(*geom(k), ctypes.c_void_p(stream))
        else:
            st = lib.bevwarp_warp_classes(*geom(k), table.data_ptr(), 1 if mode == "fill" else 0, ctypes.c_void_p(stream))
        assert st == 0, (mode, st)

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch("fill", 0)
    e1.record()
    torch.cuda.synchronize()
    t = table.cpu().numpy().view(np.uint32)
    print("table: %d entries, %d filled, fill launch %.1f us; verdicts: %s" % (t.size, int((t >> 31).sum()), e0.elapsed_time(e1) * 1e3,
          {hex(int(v)): int(c) for v, c in zip(*np.unique(t & 0x7fffffff, return_counts=True))}))
    dsts[0].zero_()
    launch("plain", 0)
    torch.cuda.synchronize()
    ref = dsts[0].clone()
    dsts[0].zero_()
    launch("use", 0)
    torch.cuda.synchronize()
    print("output with the table %s the plain call's" % ("EQUALS" if torch.equal(ref, dsts[0]) else "DIFFERS FROM"))
    times = {"plain": [], "use": []}
    rng = np.random.default_rng(7)
    for m in times:
        for k in range(3):
            launch(m, k % nsets)
    torch.cuda.synchronize()
    last = -1
    for r in range(args.rounds):
        for m in [("plain", "use")[i] for i in rng.permutation(2)]:
            k = int(rng.integers(nsets - 1))
            k = k if k < last else k + 1 if last >= 0 else k
            last = k
            e0.record()
            launch(m, k)
            e1.record()
            e1.synchronize()
            times[m].append(e0.elapsed_time(e1) * 1e3)
    for m, v in times.items():
        print("%-6s median %7.1f us  min %7.1f us" % (m, np.median(v), np.min(v)))


if __name__ == "__main__":
    main()
