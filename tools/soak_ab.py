#!/usr/bin/env python3
"""Two builds of libbevwarp.so on LARGE launches of seeded random homographies, outputs compared bit for bit (GPU box):

    python tools/soak_ab.py ref=bev_amd/csrc/variants/head.so new=bev_amd/csrc/libbevwarp.so [n_seeds]

The oracle finishes small cases in seconds; paths that only full-size launches take (24-row tiles, the straight-line form of
full-height tiles, the tail split) are soaked here against a build that does not have them.  One launch per case: 24 frames with
per-frame matrices (rotation, 0.4..3 x zoom, keystone, windows partly outside, tie-heavy integer maps), u8 / f32, 1..4 channels,
both interpolations, source widths that do and do not give a 4-byte row stride."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bev_amd import _lib, warp  # noqa: E402
from tests import workloads as wl  # noqa: E402
from tests.test_gpu_rowpath import _random_homography  # noqa: E402


def main():
    specs = [a for a in sys.argv[1:] if "=" in a]
    n_seeds = int([a for a in sys.argv[1:] if "=" not in a][0]) if any("=" not in a for a in sys.argv[1:]) else 12
    dev = torch.device("cuda", 0)
    libs = []
    for spec in specs:
        label, path = spec.split("=", 1)
        fn = ctypes.CDLL(os.path.abspath(path)).bevwarp_warp
        fn.restype, fn.argtypes = _lib.SYMBOLS["bevwarp_warp"]
        libs.append((label, fn))
    stream = torch.cuda.current_stream(dev).cuda_stream
    bad = n = 0
    for seed in range(n_seeds):
        rng = np.random.default_rng(7000 + seed)
        for case in range(6):
            sw, sh = int(rng.integers(600, 1930)), int(rng.integers(400, 1090))
            dw, dh = int(rng.integers(700, 1100)), int(rng.integers(500, 1100))
            C = int(rng.choice([3, 3, 3, 1, 2, 4]))
            u8 = rng.random() < 0.65
            interp = int(rng.integers(0, 2))
            B = 24
            Ms = np.stack([_random_homography(rng, sw, sh, dw, dh) for _ in range(B)])
            minv = warp.device_inverse(Ms, dev)
            ndt, tdt, esz = (np.uint8, torch.uint8, 1) if u8 else (np.float32, torch.float32, 4)
            f0 = torch.from_numpy(np.stack([wl.frame(50 * seed + case + i, sh, sw, ndt, C) for i in range(3)])).to(dev)
            src = f0[torch.arange(B, device=dev) % 3].contiguous()
            outs = []
            for label, fn in libs:
                dst = torch.full((B, dh, dw, C), 77, dtype=tdt, device=dev)
                st = fn(src.data_ptr(), dst.data_ptr(), B, sh, sw, dh, dw, C, src.stride(0) * esz, src.stride(1) * esz, dst.stride(0) * esz, dst.stride(1) * esz,
                        minv.data_ptr(), B, 0 if u8 else 1, interp, None, ctypes.c_void_p(stream))
                assert st == 0, (label, st)
                torch.cuda.synchronize()
                outs.append(dst)
            n += 1
            for (label, _), o in zip(libs[1:], outs[1:]):
                same = torch.equal(o.view(torch.uint8), outs[0].view(torch.uint8))  # (bytes: NaN-safe)
                if not same:
                    bad += 1
                    d = (o.view(torch.uint8) != outs[0].view(torch.uint8)).sum().item()
                    print("MISMATCH seed %d case %d: %s vs %s  %dx%d -> %dx%d C=%d %s interp=%d  bytes differing %d" % (seed, case, label, libs[0][0], sw, sh, dw, dh, C, "u8" if u8 else "f32", interp, d))
    print("soak_ab: %d cases, %d mismatching" % (n, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
