#!/usr/bin/env python3
"""Round 6: DiffusiveRestoration.restore() on whole 480x720 images (BASELINE configs[4] per GPU) -- images per sampler call x UNet call cap x early stop,
tensors in host memory, identity HFRM stand-in, no PNGs.  Prints img/s per setting (median of 3 passes after one warm-up).
    python scripts/restore_sweep.py [N_IMAGES] [S]"""
import contextlib
import io
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import wavedm_amd
from wavedm_amd import procedural as P

torch.set_grad_enabled(False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda", 0)
cfg = P.raindrop_wavelet_config()
cfg.device = dev
base = SimpleNamespace(resume="", sampling_timesteps=S, local_rank=0, image_folder="/tmp/wdm", test_set="raindrop", grid_r=16)
d = wavedm_amd.DenoisingDiffusion_Wavelet(base, cfg, generator=lambda x: x, dtype=os.environ.get("DTYPE", "bf16"))
d.model.load_state_dict(P.procedural_state_dict(cfg, seed=61), strict=True)
g = torch.Generator().manual_seed(4)
loader = [(torch.rand(1, 6, 480, 720, generator=g), f"img{k}", torch.zeros(1)) for k in range(N)]


def run(per_call, max_batch, early, even="1"):
    os.environ["WAVEDM_EVEN_CALLS"] = even
    a = SimpleNamespace(**vars(base))
    a.images_per_call, a.max_batch, a.early_stop = per_call, max_batch, early
    d.args = a
    rest = wavedm_amd.DiffusiveRestoration(d, a, cfg, save_images=False)
    ts = []
    for k in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            rest.restore(loader, validation="raindrop", r=16)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts[1:])[1]
    print(f"per_call {str(per_call):>5s}  max_batch {max_batch:4d}  early_stop {int(early)}  even_calls {even}  ->  {N / t:6.3f} img/s   ({', '.join(f'{v:.2f}' for v in ts)} s)", flush=True)


for per_call, mb, early, even in [(1, 64, False, "1"), (1, 64, True, "1"), (8, 64, True, "1"), (8, 128, True, "0"), (8, 128, True, "1"), (8, 120, True, "1"), (8, 180, True, "1"),
                                  (8, 360, True, "1"), (16, 128, True, "1"), (16, 144, True, "1"), (16, 240, True, "1"), (None, 128, True, "1"), (None, 128, False, "1")]:
    run(per_call, mb, early, even)
