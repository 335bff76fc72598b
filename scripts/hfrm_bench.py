#!/usr/bin/env python3
"""HFRM forward throughput at the reference's evaluation size (480x720, SURVEY.md §6: 1.76 s on the reference's CPU probe).
Informational (the headline metric excludes the HFRM); prints one JSON line.  --cpu also times the oracle on the host."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavedm_amd import procedural as P          # noqa: E402
from wavedm_amd.arch import HFRM                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--h", type=int, default=480)
    ap.add_argument("--w", type=int, default=720)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    m = HFRM(in_channel=3, dim=32, mid_blk_num=6, enc_blk_nums=[2, 2, 2, 4], dec_blk_nums=[2, 2, 2, 2], dtype=a.dtype)
    sd = P.procedural_hfrm_state_dict(seed=61)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(a.batch, 3, a.h, a.w, generator=g).to(dev)
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        y = m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    out = {"what": "HFRM forward", "size": [a.batch, 3, a.h, a.w], "dtype": a.dtype, "ms_per_batch": dt * 1e3,
           "images_per_s": a.batch / dt}
    if a.cpu:
        from oracle import wavedm_oracle as O
        xc = x[:1].cpu()
        O.hfrm_forward(sd, xc[:, :, :64, :64])
        t0 = time.perf_counter()
        yc = O.hfrm_forward(sd, xc)
        out["cpu_oracle_s_per_image"] = time.perf_counter() - t0
        out["cpu_threads"] = torch.get_num_threads()
        out["rel_linf_vs_oracle"] = float((y[:1].cpu() - yc).abs().max() / yc.abs().max())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
