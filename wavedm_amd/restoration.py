"""DiffusiveRestoration -- the reference's evaluation wrapper (`models/restoration.py:16-196`).

`restore(val_loader, validation, r)` consumes the loader contract `(x[B,6,H,W] in [0,1], img_id, total)`: DWT of the
degraded image, HFRM -> DWT -> `x_other`, stitched DDIM sampling, `x0_preds[-5]` (restoration.py:108), concat with the
HFRM high-frequency bands, IDWT, clamp, PSNR, PNG dumps.  Everything between the H2D copy of the batch and the PSNR
numbers stays on the GPU:

* metrics: one device reduction per image pair (imageio.sqdiff) gives the three PSNRs the reference prints
  (torchPSNR on a CPU copy, calculate_psnr_in_GPU, and the numpy calculate_psnr after two float D2H copies);
* PNGs: quantised on the device, copied on a side stream, encoded by a worker thread (imageio.AsyncImageWriter) --
  `save_images=False` switches them off;
* `args.images_per_call` (default 1 = the reference's one image per sampler call): consecutive loader items of the
  same size are restored in ONE sampler call, so a 480x720 image's 45 patches per step become 45 x N -- the start noise
  is still drawn image by image in loader order, and every kernel is batch-composition independent, so each image's
  result is bit-identical to the one-at-a-time run (tested)."""
from __future__ import annotations

import os

import numpy as np
import torch

from .ddm_wavelet import data_transform, inverse_data_transform
from . import imageio, sampling


def torchPSNR(tar_img, prd_img):
    """utils/metrics.py:7-11."""
    imdff = torch.clamp(prd_img, 0, 1) - torch.clamp(tar_img, 0, 1)
    rmse = (imdff ** 2).mean().sqrt()
    return 20 * torch.log10(1 / rmse)


def save_image(img, path):
    """utils/logging.py:9-12 without torchvision, synchronous: (1,3,H,W) or (3,H,W) in [0,1] -> PNG."""
    from PIL import Image
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    a = img.detach().float().cpu()
    if a.dim() == 4:
        a = a[0]
    a = (a * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    Image.fromarray(a).save(path)


class DiffusiveRestoration:
    def __init__(self, diffusion, args, config, save_images=True):
        self.args = args
        self.config = config
        self.diffusion = diffusion
        self.save_images = save_images
        self.writer = None
        if os.path.isfile(getattr(args, "resume", "") or ""):
            self.diffusion.model.eval()                                        # restoration.py:23-25
        else:
            print("Pre-trained diffusion model path is missing!")

    # ---- one sampler call over a group of same-sized loader items ---------------------------------------------
    def _restore_group(self, items, r, image_folder, acc):
        cfg, d = self.config, self.diffusion
        pc, ob = cfg.model.pred_channels, cfg.model.other_channels_begin
        x = torch.cat([it[0] for it in items], dim=0).to(d.device, non_blocking=True).float().contiguous()
        names = [it[1] for it in items]
        inp, gt = x[:, :3].contiguous(), x[:, 3:].contiguous()
        x_cond = d.wavelet_dec.forward_affine(inp)                             # restoration.py:79, :88: DWT(2x - 1) in one kernel
        x_gt = d.wavelet_dec.forward_affine(gt)                                # :89
        hf = d.generator(inp)                                                  # :94 (HFRM)
        hf_wav = d.wavelet_dec.forward_affine(hf.contiguous())                 # :95-96
        x_other = hf_wav[:, ob:].contiguous()                                  # :102
        xs, x0_preds = self.diffusive_restoration(x_cond, x_other=x_other, r=r, last=False, total=None,
                                                  use_global=False, use_other=True)
        pred = x0_preds[-5]                                                    # :108
        rec = lambda lo, hi: d.wavelet_rec.compose(lo, hi, pc)                    # IDWT(cat([lo[:, :pc], hi[:, pc:]])) -> clamp((x + 1) / 2), one kernel
        x_output = rec(pred, hf_wav)                                           # :114-115, :124, :134
        H, W = x_output.shape[-2:]
        m_out = imageio.psnr_from_sums(imageio.sqdiff(gt, x_output), H, W)
        m_cond = imageio.psnr_from_sums(imageio.sqdiff(gt, inp), H, W)         # IDWT(DWT(x)) == x: the "cond" image is the input
        m_hf = imageio.psnr_from_sums(imageio.sqdiff(gt, hf.clamp(0.0, 1.0)), H, W)      # restoration.py:146 clamps x_output_wdnet first
        for k, name in enumerate(names):
            name = name[0] if isinstance(name, (list, tuple)) else name
            acc["torch"].append(m_out[k][0]); acc["y"].append(m_out[k][1]); acc["wdnet"].append(m_hf[k][1])
            print("psnr this", m_out[k][0])
            print("psnr cond", m_cond[k][0])
            if self.save_images:
                sl = slice(k, k + 1)
                w = self.writer
                w.save(rec(x_gt[sl], hf_wav[sl]), os.path.join(image_folder, f"{name}_lrgt_hrwdnet.png"))      # :118-120, :158
                w.save(hf[sl], os.path.join(image_folder, f"{name}_all_wdnet.png"))
                w.save(rec(x_gt[sl], x_cond[sl]), os.path.join(image_folder, f"{name}_lrgt_hrcond.png"))       # :121-123
                w.save(rec(pred[sl], x_gt[sl]), os.path.join(image_folder, f"{name}_lrdiff_hrgt.png"))         # :112-113
                w.save(x_output[sl], os.path.join(image_folder, f"{name}_output.png"))
                w.save(inp[sl], os.path.join(image_folder, f"{name}_cond.png"))
                w.save(gt[sl], os.path.join(image_folder, f"{name}_gt.png"))
        return [x_output[k:k + 1] for k in range(len(names))]

    def restore(self, val_loader, validation="snow", r=None):
        cfg, d = self.config, self.diffusion
        if not (cfg.data.wavelet and not cfg.data.wavelet_in_unet and cfg.model.use_other_channels):
            raise NotImplementedError("DiffusiveRestoration.restore: only the raindrop_wavelet.yml branch is accelerated")
        image_folder = os.path.join(self.args.image_folder, cfg.data.dataset, validation)
        per_call = max(1, int(getattr(self.args, "images_per_call", 1) or 1))
        if self.save_images and self.writer is None:
            self.writer = imageio.AsyncImageWriter()
        acc = {"torch": [], "y": [], "wdnet": []}
        outputs, group = [], []
        with torch.no_grad():
            for i, (x, y, total) in enumerate(val_loader):
                x = x.flatten(start_dim=0, end_dim=1) if x.ndim == 5 else x    # restoration.py:72
                for k in range(x.shape[0]):                                    # loader batches are split into images
                    name = y[k] if isinstance(y, (list, tuple)) and len(y) == x.shape[0] else y
                    item = (x[k:k + 1], name)
                    if group and (len(group) == per_call or group[0][0].shape != item[0].shape):
                        outputs += self._restore_group(group, r, image_folder, acc)
                        group = []
                    group.append(item)
            if group:
                outputs += self._restore_group(group, r, image_folder, acc)
        if self.writer is not None:
            self.writer.flush()
        if acc["torch"]:
            print("psnr all torch", float(np.mean(acc["torch"])))
            print("psnr all np", float(np.mean(acc["y"])))
            print("psnr all GPU", float(np.mean(acc["y"])))       # deliberately the same accumulator: the reference's numpy and torch Y-PSNR agree
            print("psnr all wdnet", float(np.mean(acc["wdnet"])))
        self.last_outputs, self.last_psnrs, self.last_psnrs_y = outputs, acc["torch"], acc["y"]
        return outputs, acc["torch"]

    def diffusive_restoration(self, x_cond, x_other=None, r=None, last=True, total=None, use_global=False, use_other=False):
        """restoration.py:170-185.  The start noise is drawn image by image (the reference sees one image per call), so a
        batched call consumes the generator exactly like the same images restored one after the other."""
        p_size = self.config.data.patch_size if self.config.data.wavelet_in_unet else self.config.data.image_size
        h_list, w_list = self.overlapping_grid_indices(x_cond, output_size=p_size, r=r)
        corners = [(i, j) for i in h_list for j in w_list]
        shp = (1, self.config.model.pred_channels, x_cond.shape[2], x_cond.shape[3])
        x = torch.cat([torch.randn(shp, device=self.diffusion.device) for _ in range(x_cond.shape[0])], dim=0)
        return self.diffusion.sample_image(x_cond, x, x_other=x_other, last=last, patch_locs=corners, patch_size=p_size,
                                           total=total, use_global=use_global, use_other=use_other)

    def overlapping_grid_indices(self, x_cond, output_size, r=None):
        """restoration.py:187-196."""
        _, c, h, w = x_cond.shape
        return sampling.overlapping_grid_indices(h, w, output_size, r)
