#!/usr/bin/env python3
"""Probe: does running the UNet on two half-batches on two HIP streams (kernels of the two streams fill each other's launch gaps,
prologues and epilogues) beat one stream at the full batch?  Same kernels, same per-image bits (batch composition never changes them)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavedm_amd
from wavedm_amd import _lib
from wavedm_amd import procedural as P

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
cfg = P.raindrop_wavelet_config(image_size=64)
cfg.device = dev
m = wavedm_amd.DiffusionUNet(cfg, dtype="bf16")
m.load_state_dict(P.procedural_state_dict(cfg, seed=61), strict=True)
m.to(dev)
m.pack_weights()
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NIT = 20
x = (torch.randn(B, 64, 64, 96, device=dev) * 0.7).to(torch.bfloat16)
t = torch.tensor([500.0], device=dev)


def ws_for(b):
    n = int(L.wdm_unet_workspace_bytes(m._u, b))
    return torch.empty(n + 256, dtype=torch.uint8, device=dev)


def run(nstream):
    bs = B // nstream
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    wss = [ws_for(bs) for _ in range(nstream)]
    eps = torch.empty(B, 3, 64, 64, device=dev)
    torch.cuda.synchronize()
    def go():
        for it in range(NIT):
            for k, s in enumerate(streams):
                _lib.check(L.wdm_unet_forward(m._u, _lib.ptr(x[k * bs:(k + 1) * bs]), _lib.ptr(t), 1, bs, _lib.ptr(eps[k * bs:(k + 1) * bs]),
                                              _lib.ptr(wss[k]), wss[k].numel(), s.cuda_stream))
    go(); torch.cuda.synchronize()
    t0 = time.time(); go(); torch.cuda.synchronize(); dt = (time.time() - t0) / NIT
    return dt, eps.clone()

d1, e1 = run(1)
for ns in (2, 4):
    d, e = run(ns)
    print(f"B={B}: 1 stream {d1 * 1e3:.3f} ms per UNet call | {ns} streams {d * 1e3:.3f} ms ({d1 / d:.3f}x)  max|diff| {float((e - e1).abs().max()):.3g}")


def run_graph(nstream):
    bs = B // nstream
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    wss = [ws_for(bs) for _ in range(nstream)]
    eps = torch.empty(B, 3, 64, 64, device=dev)
    graphs = []
    torch.cuda.synchronize()
    for k, s in enumerate(streams):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            _lib.check(L.wdm_unet_forward(m._u, _lib.ptr(x[k * bs:(k + 1) * bs]), _lib.ptr(t), 1, bs, _lib.ptr(eps[k * bs:(k + 1) * bs]),
                                          _lib.ptr(wss[k]), wss[k].numel(), torch.cuda.current_stream().cuda_stream))
        graphs.append(g)
    def go():
        for it in range(NIT):
            for k, s in enumerate(streams):
                with torch.cuda.stream(s):
                    graphs[k].replay()
    go(); torch.cuda.synchronize()
    t0 = time.time(); go(); torch.cuda.synchronize(); dt = (time.time() - t0) / NIT
    return dt, eps.clone()

for ns in (8,):
    d, e = run(ns)
    print(f"B={B}: eager {ns} streams {d * 1e3:.3f} ms ({d1 / d:.3f}x)  max|diff| {float((e - e1).abs().max()):.3g}")
for ns in (1, 2, 4, 8):
    try:
        d, e = run_graph(ns)
        print(f"B={B}: graph {ns} streams {d * 1e3:.3f} ms ({d1 / d:.3f}x)  max|diff| {float((e - e1).abs().max()):.3g}")
    except Exception as ex:
        print("graph", ns, "failed:", repr(ex)[:300])
