#!/usr/bin/env python3
"""Random combinations of the experiment switches: every combination must run the full-width UNet (bf16 and f32x3, batch 3) to a finite result within the mode's
bound of the default build's output -- catches alternative paths that only work next to the defaults they were written against."""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavedm_amd                                    # noqa: E402
from wavedm_amd import procedural as P, _lib        # noqa: E402

torch.set_grad_enabled(False)
SW = {"WDM_GN_TILE": "012", "WDM_GN_INLINE": "012", "WDM_BN256": "012", "WDM_CONV_DMA": "01", "WDM_GEMM": "01",
      "WDM_UP4": "01", "WDM_ATTN_FUSED": "0123", "WDM_ATTN_FOLD": "01", "WDM_ATTN_SM": "01"}        # every switch the library has (csrc/common.h: EnvCfg) but the trainer's WDM_WGRAD_BG: 6912 combinations
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = random.Random(7)
R = int(os.environ.get("R", "64"))
cfg = P.raindrop_wavelet_config(image_size=R) if R != 64 else P.raindrop_wavelet_config()
sd = P.procedural_state_dict(cfg, seed=61)
g = torch.Generator().manual_seed(5)
x = torch.randn(3, 96, R, R, generator=g).cuda()
t = torch.tensor([470.0])
bad = 0
for dtype, tol in (("bf16", 3e-2), ("f16", 4e-3), ("f32x3", 1e-4)):
    net = wavedm_amd.DiffusionUNet(cfg, dtype=dtype)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    ref = net(x, t).float().cpu()
    for i in range(N):
        env = {k: rng.choice(list(v)) for k, v in SW.items() if rng.random() < 0.45}
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        _lib.env_refresh()
        try:
            net2 = wavedm_amd.DiffusionUNet(cfg, dtype=dtype)      # (a model sizes its workspace for the switches in force when it is built)
            net2.load_state_dict(sd, strict=True)
            y = net2.cuda()(x, t).float().cpu()
            del net2
            e = float((y - ref).abs().max() / ref.abs().max())
            ok = bool(torch.isfinite(y).all()) and e <= tol
        except Exception as ex:                       # noqa: BLE001
            e, ok = float("nan"), False
            print("   ", str(ex)[:200])
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            _lib.env_refresh()
        bad += 0 if ok else 1
        print(f"{dtype} #{i:2d} {'ok ' if ok else 'BAD'} rel {e:.2e}  " + " ".join(f"{k[4:]}={v}" for k, v in sorted(env.items())))
    del net
    torch.cuda.empty_cache()
sys.exit(1 if bad else 0)
