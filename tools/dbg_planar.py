import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from bev_amd import warp as W
from oracle import cpu_oracle as co
from tests import workloads as wl
B, sw, sh, dw, dh = 8, 640, 360, 512, 256
frames = np.stack([wl.frame(40 + i, sh, sw, np.uint8) for i in range(B)])
Ms = np.stack([wl.jitter_H(wl.keystone_H(sw, sh, dw, dh), i) for i in range(B)])
scale, bias = [1 / 255.0, 0.5, 2.0], [0.0, -1.0, 3.5]
for rep in range(3):
    got = W.warp_to_planar(torch.from_numpy(frames).cuda(), Ms, (dw, dh), scale=scale, bias=bias).cpu().numpy()
    u8g = W.warp_perspective(torch.from_numpy(frames).cuda(), Ms, (dw, dh)).cpu().numpy()
    for i in range(B):
        u8 = co.warp_perspective(frames[i], Ms[i], (dw, dh), 1)
        exp = np.stack([u8[:, :, k].astype(np.float32) * np.float32(scale[k]) + np.float32(bias[k]) for k in range(3)])
        bad = np.argwhere(got[i] != exp)
        if len(bad):
            print("rep", rep, "frame", i, "n", len(bad), "first", bad[:20].tolist())
            for c,y,x in bad[:6]:
                print("   got", got[i][c,y,x], "exp", exp[c,y,x], "u8 exp", u8[y,x], "u8 gpu", u8g[i][y,x])
print("done")
