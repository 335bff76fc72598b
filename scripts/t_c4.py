import sys, time, torch
sys.path.insert(0, '.')
import wavedm_amd
from wavedm_amd import procedural as P, _lib
torch.set_grad_enabled(False)
cfg = P.raindrop_wavelet_config()
net = wavedm_amd.DiffusionUNet(cfg, dtype="bf16").cuda()
net.pack_weights()
for B in (64, 45, 48, 8):
    x = torch.randn(B, 64, 64, 96, device="cuda").to(torch.bfloat16)
    t = torch.tensor([500.0], device="cuda"); eps = torch.empty(B, 3, 64, 64, device="cuda")
    for _ in range(2): net.forward_nhwc(x, t, eps)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): net.forward_nhwc(x, t, eps)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"B={B}: {dt*1e3:.2f} ms per forward, {B*79.94e9/dt/1e12:.0f} TFLOP/s")
_lib.prof_enable(True)
x = torch.randn(45, 64, 64, 96, device="cuda").to(torch.bfloat16); eps = torch.empty(45, 3, 64, 64, device="cuda")
net.forward_nhwc(x, t, eps); torch.cuda.synchronize()
rep = sorted(_lib.prof_report(), key=lambda e: -e["ms"])
for e in rep[:8]: print(e["kernel"], e["launches"], round(e["ms"], 2), "ms", round(e["flops"] / e["ms"] / 1e9), "TF")
