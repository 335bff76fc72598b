#!/usr/bin/env python3
"""Race screen for the asynchronous (LDS-DMA, counted-wait) kernels: the same full-width batch is sampled N times and every run must
reproduce the first bit for bit; so must a run with a different batch composition (per-image results are batch-independent)."""
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavedm_amd                                    # noqa: E402
from wavedm_amd import procedural as P              # noqa: E402

torch.set_grad_enabled(False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
cfg = P.raindrop_wavelet_config()
cfg.device = dev
args = SimpleNamespace(resume="", sampling_timesteps=5, local_rank=0, image_folder="/tmp/wdm", test_set="raindrop", grid_r=16, max_batch=64)
d = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype=os.environ.get("DTYPE", "bf16"))      # DTYPE=f16: the same screen for the fp16 instantiations
d.model.load_state_dict(P.procedural_state_dict(cfg, seed=61), strict=True)
rainy, x_T = P.synthetic_batch(64, patch_px=256, seed=61)
rainy, x_T = rainy.to(dev), x_T.to(dev)
ref = d.restore_batch(rainy, x_T)[0].clone()
bad = 0
for i in range(N):
    out = d.restore_batch(rainy, x_T)[0]
    if not torch.equal(out, ref):
        bad += 1
        print(f"run {i}: MISMATCH, max diff {float((out - ref).abs().max()):.3e}")
sub = d.restore_batch(rainy[5:22].contiguous(), x_T[5:22].contiguous())[0]
ok_sub = torch.equal(sub, ref[5:22])
print(f"{N} repeats: {N - bad} identical; sub-batch of 17 identical to its rows of the 64-batch: {ok_sub}; finite: {bool(torch.isfinite(ref).all())}")
sys.exit(0 if bad == 0 and ok_sub else 1)
