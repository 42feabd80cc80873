#!/usr/bin/env python3
"""A/B of several builds of libbevwarp.so in ONE process, launches interleaved (guide rule: perf deltas come from
interleaved rounds in one process on one device).  GPU box:

    python tools/abx.py --libs base=bev_amd/csrc/variants/base.so new=bev_amd/csrc/libbevwarp.so \
        [--dtype u8|f32] [--interp linear|nearest] [--homography keystone|brno] [--rounds 40] [--check]

Every library is loaded with its own ctypes handle and driven through the C ABI directly (bevwarp_warp); inputs are the
bench's configs[1] frames, rotated over > 1 GB of buffers.  Prints median / min kernel time per library and, with
--check, whether every library's output equals the first one's bit for bit."""
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
    p.add_argument("--libs", nargs="+", required=True, help="label=path ...")
    p.add_argument("--dtype", default="u8")
    p.add_argument("--interp", default="linear")
    p.add_argument("--homography", default="keystone")
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--src", type=int, nargs=2, default=[1920, 1080])
    p.add_argument("--dst", type=int, nargs=2, default=[1024, 1024])
    p.add_argument("--channels", type=int, default=3)
    p.add_argument("--rounds", type=int, default=40)
    p.add_argument("--check", action="store_true")
    args = p.parse_args()

    from bev_amd import _lib, warp
    from tests import workloads as wl
    dev = torch.device("cuda", 0)
    B, (sw, sh), (dw, dh), C = args.batch, args.src, args.dst, args.channels
    tdt, ndt, esz = (torch.uint8, np.uint8, 1) if args.dtype == "u8" else (torch.float32, np.float32, 4)
    interp = 1 if args.interp == "linear" else 0
    if args.homography.startswith("rot"):  # rot<degrees>[z<zoom>], e.g. rot15 or rot30z0.6
        deg, _, zoom = args.homography[3:].partition("z")
        base = wl.rotated_H(sw, sh, dw, dh, float(deg), float(zoom) if zoom else 0.6)
    else:
        base = {"keystone": wl.keystone_H, "inset": wl.keystone_inset_H, "brno": wl.synth_brno_H}[args.homography](sw, sh, dw, dh)
    Ms = np.stack([wl.jitter_H(base, g) for g in range(B)])
    minv = warp.device_inverse(Ms, dev)
    set_bytes = B * (sh * sw + dh * dw) * C * esz
    nsets = max(2, int(np.ceil(1.1e9 / set_bytes)))
    srcs, dsts = [], []
    f0 = torch.stack([torch.from_numpy(wl.frame(g, sh, sw, ndt, C)) for g in range(min(B, 4))]).to(dev)
    for s in range(nsets):
        t = torch.empty((B, sh, sw, C), dtype=tdt, device=dev)
        for i in range(B):
            t[i] = f0[(i + s) % f0.shape[0]] if (i + s) % 3 == 0 else f0[(i + s) % f0.shape[0]].flip(i % 2)
        srcs.append(t)
        dsts.append(torch.empty((B, dh, dw, C), dtype=tdt, device=dev))
    libs = []
    for spec in args.libs:
        label, path = spec.split("=", 1)
        lib = ctypes.CDLL(os.path.abspath(path))
        name = "bevwarp_warp"
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _lib.SYMBOLS[name]
        libs.append((label, fn))
    stream = torch.cuda.current_stream(dev).cuda_stream

    def launch(fn, k):
        s, d = srcs[k], dsts[k]
        st = fn(s.data_ptr(), d.data_ptr(), B, sh, sw, dh, dw, C, s.stride(0) * esz, s.stride(1) * esz, d.stride(0) * esz, d.stride(1) * esz,
                minv.data_ptr(), B, 0 if args.dtype == "u8" else 1, interp, None, ctypes.c_void_p(stream))
        assert st == 0, st

    if args.check:
        ref = None
        for label, fn in libs:
            dsts[0].zero_()
            launch(fn, 0)
            torch.cuda.synchronize()
            if ref is None:
                ref = dsts[0].clone()
            else:
                same = torch.equal(ref, dsts[0])
                print("%-12s output %s the first library's" % (label, "EQUALS" if same else "DIFFERS FROM"))
    times = {label: [] for label, _ in libs}
    it = 0
    for label, fn in libs:  # warm-up
        for _ in range(5):
            launch(fn, it % nsets)
            it += 1
    torch.cuda.synchronize()
    rng = np.random.default_rng(7)
    last = -1
    for r in range(args.rounds):
        order = [libs[i] for i in rng.permutation(len(libs))]  # (a fixed order ties every library to its own buffer sets)
        for label, fn in order:
            k = int(rng.integers(nsets - 1))
            k = k if k < last else k + 1 if last >= 0 else k  # any set but the one just used (still warm in the caches)
            last = k
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(fn, k)
            e1.record()
            it += 1
            e1.synchronize()
            times[label].append(e0.elapsed_time(e1) * 1e3)
    mpix = B * dw * dh / 1e6
    for label, _ in libs:
        t = np.array(times[label])
        print("%-12s median %7.1f us  min %7.1f us  mean %7.1f us   %8.0f Mpix/s (median)" % (label, np.median(t), t.min(), t.mean(), mpix / np.median(t) * 1e6))


if __name__ == "__main__":
    main()
