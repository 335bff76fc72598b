#!/usr/bin/env python3
"""Round 6: host-side timeline of DiffusiveRestoration.restore() through the WHOLE pipeline (PNG pairs on disk -> RainDrop loader -> HFRM -> sampler -> PNGs).
    python scripts/restore_trace.py [N_IMAGES] [S] [workers]"""
import contextlib
import copy
import io
import os
import shutil
import sys
import tempfile
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

import wavedm_amd
from wavedm_amd import procedural as P
from wavedm_amd.datasets import RainDrop


def main():
    torch.set_grad_enabled(False)
    os.environ["WAVEDM_RESTORE_TRACE"] = "1"
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    SAVE = os.environ.get("SAVE", "1") == "1"
    dev = torch.device("cuda", 0)
    cfg = P.raindrop_wavelet_config()
    cfg.device = dev
    root = tempfile.mkdtemp(prefix="wdm_trace_")
    rng = np.random.default_rng(44)
    for sub_ in ("raindrop_test", "train"):
        for leaf in ("input", "gt"):
            os.makedirs(os.path.join(root, "raindrop", sub_, leaf))
    for k in range(N):
        clean = rng.integers(0, 256, (480, 720, 3), dtype=np.uint8)
        drop = np.clip(clean.astype(np.int16) + rng.integers(-40, 41, (480, 720, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(drop).save(os.path.join(root, "raindrop", "raindrop_test", "input", f"{k}_rain.png"), compress_level=1)
        Image.fromarray(clean).save(os.path.join(root, "raindrop", "raindrop_test", "gt", f"{k}_clean.png"), compress_level=1)
    cfg4 = copy.deepcopy(cfg)
    cfg4.data.data_dir, cfg4.data.num_workers = root, W
    a = SimpleNamespace(resume="", sampling_timesteps=S, local_rank=0, image_folder=os.path.join(root, "out"), test_set="raindrop", grid_r=16, world_size=1, rank=0)
    if os.environ.get("PER_CALL"):
        a.images_per_call = int(os.environ["PER_CALL"])
    if os.environ.get("PREFETCH_BYTES"):
        a.prefetch_bytes = int(os.environ["PREFETCH_BYTES"])
    d = wavedm_amd.DenoisingDiffusion_Wavelet(a, cfg, generator="procedural" if os.environ.get("HFRM", "1") == "1" else (lambda x: x), dtype="bf16")
    d.model.load_state_dict(P.procedural_state_dict(cfg, seed=61), strict=True)
    rest = wavedm_amd.DiffusiveRestoration(d, a, cfg4, save_images=SAVE)
    _, val_loader = RainDrop(a, cfg4).get_loaders(parse_patches=False, validation="raindrop")
    if os.environ.get("PIN", "1") == "0" or os.environ.get("MPCTX"):
        val_loader = torch.utils.data.DataLoader(val_loader.dataset, batch_size=1, shuffle=False, num_workers=W, pin_memory=os.environ.get("PIN", "1") == "1",
                                                 multiprocessing_context=os.environ.get("MPCTX") or None)
    if os.environ.get("SYNTH") == "1":                      # no PIL, no files: items made by torch.rand in the workers
        class Synth(torch.utils.data.Dataset):
            def __len__(self):
                return N

            def __getitem__(self, i):
                x = torch.rand(6, 480, 720, generator=torch.Generator().manual_seed(i))
                return x, str(i), x[:3]
        val_loader = torch.utils.data.DataLoader(Synth(), batch_size=1, shuffle=False, num_workers=W, pin_memory=os.environ.get("PIN", "1") == "1",
                                                 multiprocessing_context=os.environ.get("MPCTX") or None)
    t_iter = time.perf_counter()
    n = sum(1 for _ in val_loader)
    print(f"loader alone: {n} items in {time.perf_counter() - t_iter:.3f} s ({W} workers)")
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        src = iter(val_loader) if os.environ.get("ITER_MAIN") == "1" else val_loader        # ITER_MAIN=1: the workers are fork()ed from the MAIN thread, not from restore()'s feeder thread
        with contextlib.redirect_stdout(io.StringIO()):
            rest.restore(src, validation="raindrop", r=16)
        torch.cuda.synchronize()
        print(f"pass {rep}: {time.perf_counter() - t0:.3f} s = {N / (time.perf_counter() - t0):.2f} img/s")
    if os.environ.get("STEPS") == "1":
        from wavedm_amd import sampling
        sampling._TRACE = rest._mark
        with contextlib.redirect_stdout(io.StringIO()):
            rest.restore(iter(val_loader) if os.environ.get("ITER_MAIN") == "1" else val_loader, validation="raindrop", r=16)
    torch.cuda.synchronize()
    ev0 = next((e[3] for e in rest.trace if e[3] is not None), None)
    for t, th, what, ev in rest.trace:
        gpu = f"   GPU reaches it {ev0.elapsed_time(ev):9.1f} ms after the first mark" if ev is not None else ""
        print(f"  {t * 1e3:9.1f} ms  {th:<24s} {what}{gpu}")
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
