"""DiffusiveRestoration -- the reference's evaluation wrapper (`models/restoration.py:16-196`).

`restore(val_loader, validation, r)` consumes the loader contract `(x[B,6,H,W] in [0,1], img_id,
total)`: DWT of the degraded image, HFRM stand-in -> DWT -> `x_other`, stitched DDIM sampling,
`x0_preds[-5]` (restoration.py:108), concat with the HFRM high-frequency bands, IDWT, clamp, PSNR.
Everything between the H2D copy of the batch and the final PSNR stays on the GPU.  PNG dumps use
PIL (torchvision is not required) and can be switched off with `save_images=False`."""
from __future__ import annotations

import os

import numpy as np
import torch

from .ddm_wavelet import data_transform, inverse_data_transform
from . import sampling


def torchPSNR(tar_img, prd_img):
    """utils/metrics.py:7-11."""
    imdff = torch.clamp(prd_img, 0, 1) - torch.clamp(tar_img, 0, 1)
    rmse = (imdff ** 2).mean().sqrt()
    return 20 * torch.log10(1 / rmse)


def save_image(img, path):
    """utils/logging.py:9-12 without torchvision: (1,3,H,W) or (3,H,W) in [0,1] -> PNG."""
    from PIL import Image
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    a = img.detach().float().cpu()
    if a.dim() == 4:
        a = a[0]
    a = (a.clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).numpy()
    Image.fromarray(a).save(path)


class DiffusiveRestoration:
    def __init__(self, diffusion, args, config, save_images=True):
        self.args = args
        self.config = config
        self.diffusion = diffusion
        self.save_images = save_images
        if os.path.isfile(getattr(args, "resume", "") or ""):
            self.diffusion.model.eval()                                        # restoration.py:23-25
        else:
            print("Pre-trained diffusion model path is missing!")

    def restore(self, val_loader, validation="snow", r=None):
        cfg, d = self.config, self.diffusion
        if not (cfg.data.wavelet and not cfg.data.wavelet_in_unet and cfg.model.use_other_channels):
            raise NotImplementedError("DiffusiveRestoration.restore: only the raindrop_wavelet.yml branch is accelerated")
        image_folder = os.path.join(self.args.image_folder, cfg.data.dataset, validation)
        psnrs, outputs = [], []
        pc, ob = cfg.model.pred_channels, cfg.model.other_channels_begin
        with torch.no_grad():
            for i, (x, y, total) in enumerate(val_loader):
                x = x.flatten(start_dim=0, end_dim=1) if x.ndim == 5 else x
                x = x.to(d.device).float().contiguous()
                x_all = data_transform(x)
                x_cond = d.wavelet_dec(x_all[:, :3].contiguous())              # restoration.py:88
                hf = d.generator(x[:, :3].contiguous())                        # :94 (HFRM, out of path)
                hf_wav = d.wavelet_dec(data_transform(hf).contiguous())        # :95-96
                x_other = hf_wav[:, ob:].contiguous()                          # :102
                xs, x0_preds = self.diffusive_restoration(x_cond, x_other=x_other, r=r, last=False, total=total,
                                                          use_global=False, use_other=True)
                x_output = x0_preds[-5]                                        # :108
                x_output = torch.cat([x_output[:, :pc], hf_wav[:, pc:]], dim=1)  # :114-115
                x_output = inverse_data_transform(d.wavelet_rec(x_output.contiguous()))   # :124,:134
                gt = x[:, 3:]
                psnr = float(torchPSNR(gt, x_output))
                psnrs.append(psnr)
                outputs.append(x_output)
                print(f"image {y}: psnr {psnr:.3f}")
                if self.save_images:
                    name = y[0] if isinstance(y, (list, tuple)) else y
                    save_image(x_output, os.path.join(image_folder, f"{name}_output.png"))
        if psnrs:
            print("psnr all torch", float(np.mean(psnrs)))
        self.last_outputs, self.last_psnrs = outputs, psnrs
        return outputs, psnrs

    def diffusive_restoration(self, x_cond, x_other=None, r=None, last=True, total=None, use_global=False, use_other=False):
        """restoration.py:170-185."""
        p_size = self.config.data.patch_size if self.config.data.wavelet_in_unet else self.config.data.image_size
        h_list, w_list = self.overlapping_grid_indices(x_cond, output_size=p_size, r=r)
        corners = [(i, j) for i in h_list for j in w_list]
        x = torch.randn((x_cond.shape[0], self.config.model.pred_channels, x_cond.shape[2], x_cond.shape[3]),
                        device=self.diffusion.device)
        return self.diffusion.sample_image(x_cond, x, x_other=x_other, last=last, patch_locs=corners, patch_size=p_size,
                                           total=total, use_global=use_global, use_other=use_other)

    def overlapping_grid_indices(self, x_cond, output_size, r=None):
        """restoration.py:187-196."""
        _, c, h, w = x_cond.shape
        return sampling.overlapping_grid_indices(h, w, output_size, r)
