#!/usr/bin/env python3
"""Training-step throughput (informational, the headline metric is sampling): the reference's per-GPU batch is
training.batch_size 1 x training.patch_n 8 = 8 crops of 256x256 px (64x64 in the wavelet domain) per iteration.
Prints one JSON line; --cpu also times the oracle's autograd step on the host (2 samples)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavedm_amd import procedural as P          # noqa: E402
from wavedm_amd.training import Trainer          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--prof", action="store_true", help="one more step under the library's per-launch event profiler: per (kernel | shape) rows on stderr")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = P.raindrop_wavelet_config()
    cfg.device = dev
    sd = P.procedural_state_dict(cfg, seed=61)
    tr = Trainer(cfg, dtype=a.dtype)
    tr.load_state_dict(sd)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(a.batch, 96, 64, 64, generator=g).to(dev)
    gd = torch.Generator(device=dev).manual_seed(2)
    losses = []
    for _ in range(2):
        losses.append(float(tr.train_step(x0, generator=gd)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        loss = tr.train_step(x0, generator=gd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    out = {"what": "training step (loss + backward + Adam + EMA), raindrop_wavelet UNet 156.5 M params", "batch": a.batch, "dtype": a.dtype,
           "ms_per_step": dt * 1e3, "samples_per_s": a.batch / dt, "loss_first": losses[0], "loss_last": float(loss),
           # forward 80 GFLOP per sample, backward ~2x
           "approx_tflops": a.batch * 79.94e9 * 3 / dt / 1e12}
    if a.prof:
        from wavedm_amd import _lib
        _lib.prof_enable(True)
        tr.train_step(x0, generator=gd)
        torch.cuda.synchronize()
        rows = sorted(_lib.prof_report(), key=lambda e: -e["ms"])
        _lib.prof_enable(False)
        tot = sum(e["ms"] for e in rows)
        for e in rows[:80]:
            print(f"[shape] {e['kernel']:<72s} n {e['launches']:4d}  avg {e['ms'] / e['launches'] * 1e3:8.1f} us  "
                  f"{(e['flops'] / e['ms'] / 1e9 if e['ms'] else 0):7.1f} TFLOP/s  {100 * e['ms'] / tot:5.1f}%", file=sys.stderr)
        out["profiled_conv_ms"] = tot
    if a.cpu:
        from oracle import wavedm_oracle as O
        xs, e, t = x0[:2].cpu(), torch.randn(2, 3, 64, 64), torch.tensor([700, 120])
        t0 = time.perf_counter()
        O.train_grads(sd, cfg, xs, t, e, O.beta_schedule(cfg))
        out["cpu_oracle_s_per_sample"] = (time.perf_counter() - t0) / 2
        out["cpu_threads"] = torch.get_num_threads()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
