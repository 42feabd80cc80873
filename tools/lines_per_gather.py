#!/usr/bin/env python3
"""Cache lines one gather instruction touches, by lane layout, on the Brno-style BEV of the bench (CPU only, numpy).

    python tools/lines_per_gather.py

For every tile of a frame that lies wholly inside the camera frame, the 128-byte lines of ONE tap row that the 64 lanes of a
gather instruction cover (aligned 12-byte windows, 8-bit RGB, 1920-pixel rows), for the lane layouts the kernel has (row
segments, 16 x 4 patches), two it does not (32 x 2, 8 x 8) and a bound on what any layout could do: a row segment whose
lanes follow the source row through the tile (`shear`: lane l takes column x0 + l of row y0 + ((p + floor(l s)) mod 24),
s = the local slope of constant source y).  DESIGN.md 6.4 quotes the output."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import workloads as wl  # noqa: E402

SW, SH, DW, DH, TH, TW = 1920, 1080, 1024, 1024, 24, 256


def main():
    H = wl.jitter_H(wl.synth_brno_H(SW, SH, DW, DH), 0)
    Mi = np.linalg.inv(H)
    ys, xs = np.mgrid[0:DH, 0:DW].astype(np.float64)
    W = Mi[2, 0] * xs + Mi[2, 1] * ys + Mi[2, 2]
    SX = (Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]) / W
    SY = (Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]) / W
    inside = (SX >= 2) & (SX < SW - 4) & (SY >= 2) & (SY < SH - 3) & (W > 0)
    lane = np.arange(64)

    def lines(sel_x, sel_y):
        sx, sy = np.floor(SX[sel_y, sel_x]).astype(np.int64), np.floor(SY[sel_y, sel_x]).astype(np.int64)
        addr = sy * (SW * 3) + ((sx * 3) & ~3)
        both = np.sort(np.concatenate([addr >> 7, (addr + 11) >> 7], axis=1), axis=1)
        return 1 + (np.diff(both, axis=1) != 0).sum(axis=1)

    def layout(x0, y0, name):
        out = []
        if name == "row segments":
            for y in range(y0, y0 + TH):
                for j in range(4):
                    out.append((x0 + 64 * j + lane, np.full(64, y)))
        elif name.startswith("patches"):
            pw = int(name.split()[1])
            ph = 64 // pw
            for xb in range(x0, x0 + TW, pw):
                for yb in range(y0, y0 + TH - ph + 1, ph):
                    out.append((xb + lane % pw, yb + lane // pw))
        else:
            xc, yc = x0 + TW // 2, y0 + TH // 2
            s = -(SY[yc, xc + 1] - SY[yc, xc]) / (SY[yc + 1, xc] - SY[yc, xc])
            for p in range(TH):
                for j in range(4):
                    x = x0 + 64 * j + lane
                    out.append((x, y0 + (p + np.floor((x - x0) * s).astype(int)) % TH))
        return np.stack([o[0] for o in out]), np.stack([o[1] for o in out])

    names = ["row segments", "patches 16 x 4", "patches 32 x 2", "patches 8 x 8", "sheared row segments"]
    tot, rows, n = {k: 0.0 for k in names}, [], 0
    for ty in range(0, DH - TH + 1, TH):
        for tx in range(0, DW, TW):
            if not inside[ty:ty + TH, tx:tx + TW].all():
                continue
            n += 1
            r = {}
            for k in names:
                sx_, sy_ = layout(tx, ty, k)
                r[k] = float(lines(sx_, sy_).mean())
                tot[k] += r[k]
            rows.append((ty, tx, r))
    print("Brno-style BEV, %d interior tiles of %d per frame; 128-byte lines per gather instruction (one tap row), mean over a tile's gathers" % (n, (DH // TH + 1) * (DW // TW)))
    for k in names:
        print("  %-22s %5.1f" % (k, tot[k] / n))
    print("by tile (y0, x0):")
    for ty, tx, r in rows[::8]:
        print("  %4d %4d  " % (ty, tx) + "  ".join("%s %.1f" % (k.split()[0] + (k.split()[1] if k.startswith("patches") else ""), v) for k, v in r.items()))


if __name__ == "__main__":
    main()
