"""How does the 8x8 LDS-DMA conv scale with the number of workgroups?  768 -> Cout at 8x8 for a range of batch sizes / Cout:
time per launch (wdm_prof events) against workgroups = (B/2) * (Cout/64).  512 workgroup slots (256 CUs x 2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import gpu_util as gu
from wavedm_amd import _lib

torch.manual_seed(0)
for cin, cout, B in [(768, 768, 64), (768, 768, 84), (768, 768, 86), (768, 768, 128), (768, 1024, 64), (768, 512, 64), (768, 512, 128), (768, 384, 64),
                     (768, 768, 42), (768, 768, 32)]:
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    b = torch.randn(cout) * 0.1
    x = torch.randn(B, cin, 8, 8)
    gu.conv(w, b, 0, x, "bf16")
    _lib.prof_enable(True)
    for _ in range(20):
        gu.conv(w, b, 0, x, "bf16")
    rep = _lib.prof_report()
    _lib.prof_enable(False)
    for e in rep:
        if "dma8" in e["kernel"]:
            wgs = ((B + 1) // 2) * ((cout + 63) // 64)
            us = e["ms"] * 1e3 / e["launches"]
            print(f"{e['kernel']:60s} B={B:4d} wgs={wgs:5d} {us:8.2f} us  {e['flops'] / e['launches'] / us / 1e6:7.1f} TF  us/wave-of-512={us / max(1, (wgs + 511) // 512):.2f}", flush=True)
