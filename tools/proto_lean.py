#!/usr/bin/env python3
"""Timing of tools/proto_lean.hip (the row-affine interior pipeline as a kernel of its own) against the product, launches interleaved in one
process (GPU box):  python tools/proto_lean.py [--variants 4624 2628 ...] [--rounds 40] [--homography inset|keystone]

variant = PPL * 1000 + ROWS * 100 + AHEAD * 10 + WPE (pixels per lane, passes per wave, tap sets in flight, waves per SIMD asked for).
The prototype treats every tile as interior: the source batch is padded with a frame on either side, and on the `inset` footprint
(every tap inside the frame) its output must equal the product's except in passes that hold a tie pixel (counted)."""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libproto_lean.so")


def build():
    src = os.path.join(ROOT, "tools", "proto_lean.hip")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
                               "-I" + os.path.join(ROOT, "bev_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-Wno-unused-function",
                               "-Wno-undefined-internal", src, "-o", SO])


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--variants", type=int, nargs="+", default=[4624, 4625, 4615, 4634, 2624, 2626, 2628, 2636, 2618, 2826, 2828, 2428, 1828, 1848, 1428, 1448])
    p.add_argument("--rounds", type=int, default=40)
    p.add_argument("--homography", default="inset")
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--detail", action="store_true", help="print every sample by buffer set")
    args = p.parse_args()
    build()
    from bev_amd import _lib, warp
    from tests import workloads as wl
    dev = torch.device("cuda", 0)
    B, sw, sh, dw, dh, C = args.batch, 1920, 1080, 1024, 1024, 3
    base = {"keystone": wl.keystone_H, "inset": wl.keystone_inset_H}[args.homography](sw, sh, dw, dh)
    Ms = np.stack([wl.jitter_H(base, g) for g in range(B)])
    minv = warp.device_inverse(Ms, dev)
    nsets = 5
    f0 = torch.stack([torch.from_numpy(wl.frame(g, sh, sw, np.uint8, C)) for g in range(4)]).to(dev)
    pads, srcs, dsts = [], [], []
    for s in range(nsets):
        t = torch.zeros((B + 2, sh, sw, C), dtype=torch.uint8, device=dev)  # a frame of padding on either side
        for i in range(B):
            t[i + 1] = f0[(i + s) % 4] if (i + s) % 3 == 0 else f0[(i + s) % 4].flip(i % 2)
        pads.append(t)
        srcs.append(t[1:B + 1])
        dsts.append(torch.empty((B, dh, dw, C), dtype=torch.uint8, device=dev))
    lib = _lib.load()
    proto = ctypes.CDLL(SO).proto_lean
    proto.restype = ctypes.c_int
    proto.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_long] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
    ties = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run(v, k):
        s, d = srcs[k], dsts[k]
        if v == 0:
            st = lib.bevwarp_warp(s.data_ptr(), d.data_ptr(), B, sh, sw, dh, dw, C, s.stride(0), s.stride(1), d.stride(0), d.stride(1), minv.data_ptr(), B, 0, 1,
                                  None, ctypes.c_void_p(stream))
        else:
            st = proto(v, s.data_ptr(), d.data_ptr(), minv.data_ptr(), B, sh, sw, dh, dw, s.stride(0), s.stride(1), d.stride(0), d.stride(1), ties.data_ptr(),
                       ctypes.c_void_p(stream))
        assert st == 0, (v, st)

    variants = [0] + args.variants
    dsts[0].zero_()
    run(0, 0)
    torch.cuda.synchronize()
    ref = dsts[0].clone()
    for v in args.variants:
        dsts[0].zero_()
        ties.zero_()
        run(v, 0)
        torch.cuda.synchronize()
        diff = (ref != dsts[0]).any(dim=-1)
        if v % 1000000 >= 10000:
            continue
        print("variant %d: %d of %d pixels differ from the product's (%d wave-tiles hold a tie pixel)" % (v, int(diff.sum()), diff.numel(), int(ties.item())))
    times = {v: [] for v in variants}
    for v in variants:
        for k in range(3):
            run(v, k % nsets)
    torch.cuda.synchronize()
    rng = np.random.default_rng(7)
    last = -1
    sets, prev_v = {}, -1
    for r in range(args.rounds):
        for v in [variants[i] for i in rng.permutation(len(variants))]:
            k = int(rng.integers(nsets - 1))
            k = k if k < last else k + 1 if last >= 0 else k
            last = k
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(v, k)
            e1.record()
            e1.synchronize()
            times[v].append(e0.elapsed_time(e1) * 1e3)
            sets.setdefault(v, []).append((k, prev_v))
            prev_v = v
    for v in variants:
        t = np.array(times[v])
        print("%-8s median %7.1f us  min %7.1f us" % ("product" if v == 0 else v, np.median(t), t.min()))
        if args.detail:
            ks = np.array([k for k, _ in sets[v]])
            print("         by buffer set: " + "  ".join("%d: %s" % (k, " ".join("%.0f" % x for x in np.sort(t[ks == k]))) for k in range(nsets)))


if __name__ == "__main__":
    main()
