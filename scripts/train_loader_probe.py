#!/usr/bin/env python3
"""Round 6: does a fork-started DataLoader slow the TRAINING loop the way it slows restore()?  One MI355X, the raindrop_wavelet UNet, the reference's per-GPU batch
(8 crops of 256x256 per iteration), 30 iterations of DenoisingDiffusion_Wavelet.train_step fed by a DataLoader over synthetic crops: num_workers = 0, 4 fork-started workers,
4 workers from a fork server.      python scripts/train_loader_probe.py"""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import wavedm_amd
from wavedm_amd import procedural as P


class Crops(torch.utils.data.Dataset):
    def __len__(self):
        return 4096

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(i)
        return torch.rand(8, 6, 256, 256, generator=g), str(i), torch.zeros(1)


def main():
    dev = torch.device("cuda", 0)
    cfg = P.raindrop_wavelet_config()
    cfg.device = dev
    args = SimpleNamespace(resume="", sampling_timesteps=25, local_rank=0, image_folder="/tmp/wdm", test_set="raindrop", grid_r=16)
    d = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="bf16")
    d.model.load_state_dict(P.procedural_state_dict(cfg, seed=61), strict=True)
    d.make_trainer()
    for name, kw in (("num_workers=0", dict(num_workers=0)), ("4 workers, fork", dict(num_workers=4, multiprocessing_context="fork")),
                     ("4 workers, forkserver", dict(num_workers=4, multiprocessing_context="forkserver")), ("num_workers=0 (again)", dict(num_workers=0))):
        loader = torch.utils.data.DataLoader(Crops(), batch_size=1, shuffle=False, pin_memory=True, **kw)
        it = iter(loader)
        for _ in range(3):
            d.train_step(next(it)[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host = 0.0
        for _ in range(30):
            x = next(it)[0]
            h0 = time.perf_counter()
            d.train_step(x)
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        print(f"{name:<26s} {dt * 1e3:7.2f} ms per iteration (host time inside train_step {host / 30 * 1e3:6.2f} ms)", flush=True)
        del it, loader


if __name__ == "__main__":
    main()
