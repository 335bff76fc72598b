import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
import gpu_util as gu
torch.manual_seed(0)
cin, cout, B, H = 64, 128, 2, 16
w = gu.seeded((cout, cin, 3, 3), 100) / (cin * 9) ** 0.5
b = gu.seeded((cout,), 200) * 0.1
x = gu.seeded((B, cin, H, H), 316)
ref = torch.nn.functional.conv2d(x, w, b, padding=1)
for dt in ("f32", "f32", "bf16"):
    y = gu.conv(w, b, 0, x, dt)
    err = (y - ref).abs()
    bad = err > 1e-2 * ref.abs().max()
    print(dt, "max err", float(err.max()), "bad count", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero()
        print(" bad channels:", sorted(set(idx[:, 1].tolist()))[:40])
        print(" bad rows:", sorted(set(idx[:, 2].tolist())))
        print(" bad cols:", sorted(set(idx[:, 3].tolist())))
        print(" bad imgs:", sorted(set(idx[:, 0].tolist())))
