#!/usr/bin/env python3
"""Per-pass time of the chunked sampler over many passes (torch's stream pool has 32 streams: fresh streams per call wrap around after 8 calls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
import wavedm_amd
from wavedm_amd import procedural as P
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
cfg = P.raindrop_wavelet_config(); cfg.device = dev
args = SimpleNamespace(resume="", sampling_timesteps=int(os.environ.get("S", "25")), local_rank=0, image_folder="/tmp/wdm", test_set="raindrop", grid_r=16, max_batch=64)
d = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="bf16")
d.model.load_state_dict(P.procedural_state_dict(cfg, seed=61), strict=True)
rainy, x_T = P.synthetic_batch(64, patch_px=256, seed=61)
rainy, x_T = rainy.to(dev), x_T.to(dev)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.restore_batch(rainy, x_T)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(os.environ.get("WAVEDM_STREAM_MODE", "fresh"), os.environ.get("GPU_MAX_HW_QUEUES", "-"), " ".join(f"{t:.0f}" for t in ts))
