#!/usr/bin/env python3
"""Soak of the HIP warp against the oracle on seeded random homographies beyond what the test suite runs (GPU box):
   python tools/soak_random.py [first_seed] [n_seeds] [big]
`big`: destination heights up to 700 rows (taller-tile launches, more edge-cut / outside tiles per frame)."""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from bev_amd import warp as W  # noqa: E402
from oracle import cpu_oracle as co  # noqa: E402
from tests import workloads as wl  # noqa: E402
from tests.test_gpu_rowpath import _random_homography  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 6
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
big = len(sys.argv) > 3
bad = n = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(1000 + seed)
    for case in range(10):
        sw, sh = int(rng.integers(40, 1300 if big else 700)), int(rng.integers(30, 800 if big else 400))
        dw, dh = int(rng.integers(1, 1100 if big else 600)), int(rng.integers(1, 700 if big else 90))
        c = int(rng.integers(1, 5))
        dtype = np.uint8 if rng.random() < 0.6 else np.float32
        interp = int(rng.integers(0, 2))
        B = int(rng.choice([1, 1, 3, 9] if big else [1, 1, 3]))
        Ms = np.stack([_random_homography(rng, sw, sh, dw, dh) for _ in range(B)])
        frames = np.stack([wl.frame(100 * seed + case + i, sh, sw, dtype, c) for i in range(B)])
        border = None if rng.random() < 0.5 else [float(rng.integers(0, 200))] * c
        got = W.warp_perspective(torch.from_numpy(frames).cuda(), Ms, (dw, dh), flags=interp, border_value=border).cpu().numpy()
        for i in range(B):
            exp = co.warp_perspective(frames[i], Ms[i], (dw, dh), interp, border_value=0 if border is None else border, nthreads=16)
            n += 1
            if not np.array_equal(got[i].reshape(exp.shape), exp):
                bad += 1
                print("MISMATCH seed %d case %d frame %d: %dx%d -> %dx%d c=%d %s interp=%d (%d pixels differ)" % (
                    seed, case, i, sw, sh, dw, dh, c, np.dtype(dtype).name, interp, int((got[i].reshape(exp.shape) != exp).any(-1).sum() if exp.ndim == 3 else (got[i].reshape(exp.shape) != exp).sum())))
        if interp == 1 or rng.random() < 0.5:  # the planar store path of the same launch shape
            sc, bi = np.linspace(0.5, 2.0, c), np.linspace(-1.0, 1.0, c)
            pl = W.warp_to_planar(torch.from_numpy(frames).cuda(), Ms, (dw, dh), scale=sc, bias=bi, flags=interp, border_value=border).cpu().numpy()
            for i in range(B):
                exp = co.warp_perspective(frames[i], Ms[i], (dw, dh), interp, border_value=0 if border is None else border, nthreads=16).reshape(dh, dw, c)
                ref = (exp.astype(np.float32).transpose(2, 0, 1) * sc.astype(np.float32)[:, None, None] + bi.astype(np.float32)[:, None, None]).astype(np.float32)
                n += 1
                if not np.array_equal(pl[i].reshape(ref.shape), ref):
                    bad += 1
                    print("PLANAR MISMATCH seed %d case %d frame %d: %dx%d -> %dx%d c=%d %s interp=%d" % (seed, case, i, sw, sh, dw, dh, c, np.dtype(dtype).name, interp))
    if bad > 5:
        break
# the one-launch composite (f3) against three device warps + the blend launch, on random window pairs
from bev_amd.compo import composite_bev_img, composite_reg_img  # noqa: E402
nc = badc = 0
RT = np.eye(4)
RT[2, 3] = 1.0  # homo_from_KRt(K, Rt_homo=RT) == K: the camera map is then inv(K)
for seed in range(first, first + count):
    rng = np.random.default_rng(5000 + seed)
    for case in range(4):
        sw, sh = int(rng.integers(40, 700)), int(rng.integers(30, 400))
        fw, fh = (sw, sh) if rng.random() < 0.5 else (int(rng.integers(40, 500)), int(rng.integers(30, 300)))
        dw, dh = int(rng.integers(1, 500)), int(rng.integers(1, 120))
        c = int(rng.choice([1, 3, 3, 4]))
        M_bg, M_cam = _random_homography(rng, sw, sh, dw, dh), _random_homography(rng, fw, fh, dw, dh)
        bg, fg, mk = (torch.from_numpy(wl.frame(7 * seed + case + k, h_, w_, np.uint8, c)).cuda() for k, (w_, h_) in enumerate(((sw, sh), (fw, fh), (fw, fh))))
        one, _ = composite_bev_img(bg, fg, mk, np.eye(3), M_bg, np.linalg.inv(M_cam), RT, dw, dh)
        three = composite_reg_img(W.warp_perspective(bg, M_bg, (dw, dh)), W.warp_perspective(fg, M_cam, (dw, dh)), W.warp_perspective(mk, M_cam, (dw, dh)))
        nc += 1
        if not torch.equal(one.reshape(three.shape), three):
            badc += 1
            print("COMPOSITE MISMATCH seed %d case %d: bg %dx%d fg %dx%d -> %dx%d c=%d (%d bytes differ)" % (seed, case, sw, sh, fw, fh, dw, dh, c, int((one.reshape(three.shape) != three).sum())))
print("soak: %d warps compared, %d mismatches; %d composites compared, %d mismatches" % (n, bad, nc, badc))
