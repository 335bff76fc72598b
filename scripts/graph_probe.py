"""hipGraph replay of the sampling loop against direct launches, one process: time per 20-step pass of 64 crops and bit equality."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
import wavedm_amd
from wavedm_amd import procedural as P, sampling

torch.set_grad_enabled(False)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = P.raindrop_wavelet_config(); cfg.device = torch.device("cuda", 0)
a = SimpleNamespace(resume="", sampling_timesteps=S, local_rank=0, image_folder="/tmp/wdm", test_set="raindrop", grid_r=16, max_batch=64)
d = wavedm_amd.DenoisingDiffusion_Wavelet(a, cfg, generator=lambda x: x, dtype=os.environ.get("DTYPE", "bf16"))
d.model.load_state_dict(P.procedural_state_dict(cfg, seed=61), strict=True)
rainy, x_T = P.synthetic_batch(64, patch_px=256, seed=61)
rainy, x_T = rainy.cuda(), x_T.cuda()
res = {}
for mode in ("0", "1", "0", "1"):
    os.environ["WAVEDM_GRAPH"] = mode
    out = d.restore_batch(rainy, x_T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = d.restore_batch(rainy, x_T)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"WAVEDM_GRAPH={mode}: {dt * 1e3:.1f} ms per pass of {S} steps = {64 / dt * S / 100:.1f} img/s at 100 steps; graphs cached: {0 if sampling._GRAPHS is None else len(sampling._GRAPHS)}")
    res.setdefault(mode, out)
print("same bits:", all(torch.equal(u, v) for u, v in zip(res["0"], res["1"])))
