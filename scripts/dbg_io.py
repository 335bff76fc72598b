import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from wavedm_amd import procedural as P
import wavedm_amd
from test_gpu_unet import make_diffusion
torch.set_grad_enabled(False)
d, args = make_diffusion(P.reduced_config(), "f32", 6)
g = torch.Generator().manual_seed(21)
items = [(torch.rand(1, 6, 96, 112, generator=g), (f"im{k}",), torch.zeros(1)) for k in range(3)]
res = {}
for per_call in (1, 2):
    for save in (False, True):
        args.images_per_call = per_call
        args.image_folder = f"/tmp/dbg{per_call}{save}"
        rest = wavedm_amd.DiffusiveRestoration(d, args, d.config, save_images=save)
        torch.manual_seed(5)
        o, psnr = rest.restore(items, validation="raindrop", r=4)
        torch.cuda.synchronize()
        res[(per_call, save)] = [t.cpu() for t in o]
        print(per_call, save, [bool(torch.isnan(t).any()) for t in o], psnr)
for k in range(3):
    print(k, [torch.equal(res[(1, False)][k], res[key][k]) for key in res])
