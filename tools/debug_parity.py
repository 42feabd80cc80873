"""GPU box: print where the HIP warp differs from the oracle for one case (debug aid)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import cpu_oracle as co
from tests import workloads as wl
from bev_amd import warp as W

kind, sw, sh, dw, dh, dtype, interp = sys.argv[1], *map(int, sys.argv[2:6]), sys.argv[6], int(sys.argv[7])
dt = np.uint8 if dtype == "u8" else np.float32
M = wl.synth_brno_H(sw, sh, dw, dh) if kind == "brno" else wl.keystone_H(sw, sh, dw, dh)
src = wl.frame(0, sh, sw, dt)
got = W.warp_perspective(torch.from_numpy(src).cuda(), M, (dw, dh), flags=interp).cpu().numpy()
exp = co.warp_perspective(src, M, (dw, dh), interp)
bad = (got != exp).any(axis=2)
print("mismatching pixels", bad.sum(), "of", bad.size)
rows = np.nonzero(bad.any(axis=1))[0]
print("rows with mismatches:", len(rows), rows[:40])
for y in rows[:6]:
    xs = np.nonzero(bad[y])[0]
    print("row", y, "n", len(xs), "x range", xs.min(), xs.max(), "first:", xs[:8])
    for x in xs[:4]:
        print("   x", x, "got", got[y, x], "exp", exp[y, x])
# per 256-wide segment statistics
for tx in range(0, dw, 256):
    seg = bad[:, tx:tx + 256]
    full = (seg.sum(axis=1) == seg.shape[1]).sum()
    print("segment x0=%d: rows all-bad %d, partly bad %d, clean %d" % (tx, full, ((seg.sum(axis=1) > 0) & (seg.sum(axis=1) < seg.shape[1])).sum(), (seg.sum(axis=1) == 0).sum()))
