"""Soak: 120 bf16 training steps of the full model on a fixed random batch; the loss must settle near E|e|^2 = 3*64*64 and stay finite."""
import sys, torch
sys.path.insert(0, '/root/repo')
from wavedm_amd import procedural as P
from wavedm_amd.training import Trainer
cfg = P.raindrop_wavelet_config(); cfg.device = torch.device('cuda', 0)
tr = Trainer(cfg, dtype='bf16', lr=2e-4)
tr.load_state_dict(P.procedural_state_dict(cfg, seed=61))
g = torch.Generator().manual_seed(1)
x0 = torch.randn(8, 96, 64, 64, generator=g).cuda()
gd = torch.Generator(device='cuda').manual_seed(2)
ls = []
for i in range(120):
    ls.append(float(tr.train_step(x0, generator=gd)))
print('loss first 5', [round(v) for v in ls[:5]], 'last 5', [round(v) for v in ls[-5:]], 'finite', all(v == v for v in ls))
print('param finite', bool(torch.isfinite(tr.params).all()), 'ema finite', bool(torch.isfinite(tr.ema).all()))
