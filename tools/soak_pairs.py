#!/usr/bin/env python3
"""Soak of the pair-load tiles (row-affine interior tiles of 8-bit RGB, rows_sample.inc issue_p) against the oracle on seeded random
rectification-form maps (GPU box):  python tools/soak_pairs.py [first_seed] [n_seeds]
Random source / destination sizes, batches from 1 frame to launches with full-height tiles and a split tail, horizontal scales 0.2 .. 2.4
(both sides of the 2 - 1/16 limit), mirrored maps, keystone strengths of both signs, sub-pixel shifts on and off rounding ties, row strides
with and without padding, views that start off a 4-byte boundary, with and without verdict tables (the Python entry's own)."""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from bev_amd import warp as W  # noqa: E402
from oracle import cpu_oracle as co  # noqa: E402
from tests import workloads as wl  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = n = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(5000 + seed)
    big = seed % 4 == 0
    sw, sh = int(rng.integers(300, 1400)), int(rng.integers(100, 500))
    dw = int(rng.choice([256, 300, 512, 640, 1024])) if not big else 512
    dh = int(rng.integers(5, 200)) if not big else int(rng.choice([768, 792, 1024]))
    B = int(rng.choice([1, 2, 5])) if not big else 32
    Ms = []
    for _ in range(B):
        s = rng.uniform(0.2, 2.4) * (-1 if rng.random() < 0.1 else 1)
        sy = rng.uniform(0.1, 0.9) * sh / dh
        k = rng.uniform(-0.3, 0.3) / dh                      # W = 1 + k y: the step changes down the frame
        tx = rng.uniform(0, 40) if s > 0 else sw - rng.uniform(0, 40)
        if rng.random() < 0.3:
            tx = round(tx * 32) / 32                          # rounding ties
        Ms.append(np.array([[s, rng.uniform(-0.2, 0.2), tx], [0.0, sy, rng.uniform(0, 20)], [0.0, k, 1.0]]))
    Ms = np.stack(Ms)
    frames = np.stack([wl.frame(200 + (seed * 7 + i) % 5, sh, sw, np.uint8) for i in range(B)])
    pad = int(rng.choice([0, 1, 4]))                          # padded rows (4: the stride stays a multiple of 4 when sw is)
    off = int(rng.integers(0, 4))
    big_t = torch.zeros((B, sh, sw + pad + off, 3), dtype=torch.uint8, device="cuda")
    big_t[:, :, off:off + sw] = torch.from_numpy(frames).cuda()
    view = big_t[:, :, off:off + sw]
    for rep in range(2):                                      # second call: the verdict table of the first (matrices owned by device_inverse)
        got = W.warp_perspective(view, Ms, (dw, dh), flags=1 | 16).cpu().numpy()
        for i in range(B):
            exp = co.warp_perspective(frames[i], Ms[i], (dw, dh), 1, m_is_inverse=True)
            n += 1
            if not np.array_equal(got[i], exp):
                bad += 1
                print("MISMATCH seed %d frame %d rep %d: %d pixels" % (seed, i, rep, int((got[i] != exp).any(axis=-1).sum())), flush=True)
    if seed % 8 == 0:
        print("seed %d done (%d warps, %d mismatches)" % (seed, n, bad), flush=True)
print("soak_pairs: %d warps compared, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
