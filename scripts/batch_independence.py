#!/usr/bin/env python3
"""An image's UNet output must not depend on the batch it sits in (tile choices are functions of the layer shape, never of the batch): image 0 of batches of
1, 2, 3, 7, 33, 64, 100 and 128 crops, every compute mode, bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavedm_amd                                    # noqa: E402
from wavedm_amd import procedural as P              # noqa: E402

torch.set_grad_enabled(False)
R = int(os.environ.get("R", "64"))                  # wavelet-domain patch size: 64 (BASELINE configs[1]) or 128 (configs[2])
cfg = P.raindrop_wavelet_config(image_size=R) if R != 64 else P.raindrop_wavelet_config()
sd = P.procedural_state_dict(cfg, seed=61)
g = torch.Generator().manual_seed(5)
SIZES = (1, 2, 3, 7, 33, 64, 100, 128) if R == 64 else (1, 2, 5, 16, 33)
x = torch.randn(max(SIZES), 96, R, R, generator=g)
t = torch.tensor([470.0])
bad = 0
for dtype in sys.argv[1:] or ["bf16", "f32x3", "f32"]:
    net = wavedm_amd.DiffusionUNet(cfg, dtype=dtype)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    ref = None
    for B in SIZES:
        if dtype == "f32" and B > 33:
            continue
        y = net(x[:B].cuda(), t)[:1].cpu()
        if ref is None:
            ref = y
        same = torch.equal(y, ref)
        bad += 0 if same else 1
        print(f"{dtype} B={B:3d}: image 0 {'identical' if same else 'DIFFERS by %.3e' % float((y - ref).abs().max())}  finite {bool(torch.isfinite(y).all())}")
    del net
    torch.cuda.empty_cache()
sys.exit(1 if bad else 0)
