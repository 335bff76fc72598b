#!/usr/bin/env python3
"""Round 6: the training step's ~1 260 launches replayed from ONE hipGraph against direct launches -- loss_and_grads captured (fixed launch sequence for a
batch shape: timesteps, noise and samples are device tensors), Adam + EMA launched behind it (its step count is a kernel argument).  Same buffers, same kernels: same bits.
    python scripts/train_graph_probe.py [BATCH] [ITERS]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavedm_amd import procedural as P          # noqa: E402
from wavedm_amd.training import Trainer          # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
cfg = P.raindrop_wavelet_config()
cfg.device = dev
tr = Trainer(cfg, dtype="bf16")
tr.load_state_dict(P.procedural_state_dict(cfg, seed=61))
g = torch.Generator().manual_seed(1)
x0 = torch.randn(B, 96, 64, 64, generator=g).to(dev)
e = torch.randn(B, 3, 64, 64, generator=g).to(dev)
t = torch.randint(0, 1000, (B,), generator=g).to(dev)


def timed(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def eager():
    tr.loss_and_grads(x0, t, e)
    tr.optimizer_step()


ms_e = timed(eager, IT)
grads_e = None
tr.loss_and_grads(x0, t, e)
torch.cuda.synchronize()
grads_e = tr.grads.clone()
gr = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    tr.loss_and_grads(x0, t, e)
torch.cuda.current_stream().wait_stream(side)
with torch.cuda.graph(gr):
    tr.loss_and_grads(x0, t, e)


def graphed():
    gr.replay()
    tr.optimizer_step()


gr.replay()
torch.cuda.synchronize()
same = torch.equal(tr.grads, grads_e)
ms_g = timed(graphed, IT)
ms_e2 = timed(eager, IT)
print(f"batch {B}: direct {ms_e:.2f} ms / step, hipGraph replay {ms_g:.2f} ms / step, direct again {ms_e2:.2f};  gradients bit-identical: {same}")
