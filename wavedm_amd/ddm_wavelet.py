"""DenoisingDiffusion_Wavelet -- the reference's call surface (`models/ddm_wavelet.py:127-506`)
in front of the HIP sampling path.

Kept: constructor `(args, config)`, attributes `.model .wavelet_dec .wavelet_rec .generator
.betas .num_timesteps .device`, `sample_image(...)`, `diffusive_restoration(...)`,
`overlapping_grid_indices(...)`, `generalized_steps_overlapping(...)`, `load_ddm_ckpt(path, ema)`
and the checkpoint dict format (`state_dict` with the reference's keys, optional `ema_helper`).
`.generator` is the HFRM (`wavedm_amd.arch.HFRM`, same constructor arguments as ddm_wavelet.py:137-142) loaded from
`saved_models/raindrop/lastest.pth` (or `args.hfrm_ckpt`) when that file exists; the reference ships no such file
(SURVEY.md §4), so without one the identity stand-in BASELINE.md §3 names is used and a warning says so.
`generator=` overrides both: any callable, or "procedural" for an HFRM with seeded weights (tests, bench).
Training (`train`, `train_step`, `make_trainer`) runs on `wavedm_amd.training.Trainer`: the reference's loop body
with one flat-buffer gradient all-reduce in place of DistributedDataParallel; `--resume` restores weights, EMA shadow, Adam moments and
the step count (ddm_wavelet.py:180-190).
Deliberate differences: the model is NOT wrapped in DistributedDataParallel (inference needs no
gradient all-reduce; `.model.module` is provided for callers that unwrap), outputs stay on the
GPU (no per-step `.to('cpu')`), and the per-step statistics print is opt-in (`verbose=True`)."""
from __future__ import annotations

import os

import torch

from . import _lib, sampling
from .unet import DiffusionUNet
from .wavelet import WaveletTransform


def data_transform(X):             # ddm_wavelet.py:27-28
    return 2 * X - 1.0


def inverse_data_transform(X):     # ddm_wavelet.py:31-32
    return torch.clamp((X + 1.0) / 2.0, 0.0, 1.0)


class DenoisingDiffusion_Wavelet(object):
    def __init__(self, args, config, generator=None, dtype=None, verbose=False):
        super().__init__()
        self.args = args
        self.config = config
        self.device = torch.device(config.device) if hasattr(config, "device") else torch.device("cuda", getattr(args, "local_rank", 0))
        if self.device.type != "cuda":
            raise RuntimeError("wavedm_amd runs on MI355X only: config.device must be a cuda (ROCm) device")
        self.verbose = verbose
        self.patch_group = None     # set to a process group (or True) to shard the PATCHES of each image over the ranks

        self.wavelet_dec = WaveletTransform(scale=2, dec=True)
        self.wavelet_rec = WaveletTransform(scale=2, dec=False)
        # No mode named anywhere (argument, config.model.hip_dtype, WAVEDM_DTYPE): the conformant defaults -- the sampler in f16 (below), the HFRM in its exact fp32 mode
        # (its output IS the high-frequency part of the restored image, restoration.py:114: the bf16 HFRM's ~1e-2 would land there directly; 3.2 ms per 480x720 image
        # instead of 1.8 at seven images per call, against >= 67 ms of sampling).  A named mode is taken as named for both.
        auto = dtype is None and not getattr(config.model, "hip_dtype", None) and not os.environ.get("WAVEDM_DTYPE")
        self.generator = self._make_generator(generator, "f32" if auto else dtype)

        if getattr(config.data, "global_attn", False):                          # ddm_wavelet.py:149-152
            from .unet_global import DiffusionUNet_Global
            self.model = DiffusionUNet_Global(config, dtype=dtype).to(self.device)
        else:
            # No mode named anywhere (argument, config.model.hip_dtype, WAVEDM_DTYPE): the sampler runs in f16 -- the bf16 kernels on fp16 operands, the throughput
            # mode's speed at <= 1e-3 of the fp32 result ON SAMPLER OUTPUTS (xs, x0_preds, restored images; one UNet forward alone sits at 1.15e-3 ... 1.34e-3 of
            # max|eps|: DESIGN 3.9) -- unless the checkpoint holds a weight fp16 cannot: then bf16, with a RuntimeWarning (DiffusionUNet.pack_weights)
            self.model = DiffusionUNet(config, dtype="f16" if auto else dtype).to(self.device)
            if auto:
                self.model._dtype_fallback = "bf16"
        self.start_epoch, self.step = 0, 0
        self.ema_shadow = None
        self.optimizer_state = None

        if os.path.isfile(getattr(args, "resume", "") or ""):
            self.load_ddm_ckpt(args.resume)

        betas = sampling.get_beta_schedule(
            beta_schedule=config.diffusion.beta_schedule, beta_start=config.diffusion.beta_start,
            beta_end=config.diffusion.beta_end, num_diffusion_timesteps=config.diffusion.num_diffusion_timesteps)
        self.betas = torch.from_numpy(betas).float().to(self.device)
        self.num_timesteps = self.betas.shape[0]

    # ---- HFRM (ddm_wavelet.py:137-147) -------------------------------------------------------------
    HFRM_ARGS = dict(in_channel=3, dim=32, mid_blk_num=6, enc_blk_nums=[2, 2, 2, 4], dec_blk_nums=[2, 2, 2, 2])

    def _make_generator(self, generator, dtype):
        if callable(generator):
            return generator
        from .arch import HFRM
        if generator == "procedural":
            from .procedural import procedural_hfrm_state_dict
            g = HFRM(**self.HFRM_ARGS, dtype=dtype)
            g.load_state_dict(procedural_hfrm_state_dict(seed=getattr(self.args, "seed", 61)), strict=True)
            return g.to(self.device).eval().requires_grad_(False)
        if generator is not None:
            raise ValueError("generator must be a callable, 'procedural' or None")
        path = getattr(self.args, "hfrm_ckpt", None) or "saved_models/raindrop/lastest.pth"
        if os.path.isfile(path):
            g = HFRM(**self.HFRM_ARGS, dtype=dtype)
            g.load_state_dict(torch.load(path, map_location="cpu", weights_only=False), strict=True)
            return g.to(self.device).eval().requires_grad_(False)
        import warnings
        warnings.warn(f"HFRM checkpoint {path!r} not found: using the identity stand-in for .generator", stacklevel=3)
        return lambda x: x

    # ---- checkpoint (utils/logging.py:21-29 + ddm_wavelet.py:180-190) -------------------------
    def load_ddm_ckpt(self, load_path, ema=False):
        try:        # zip-format checkpoints are memory-mapped: tensors are paged in one by one while they are copied to the GPU,
            ckpt = torch.load(load_path, map_location="cpu", weights_only=False, mmap=True)     # no 626 MB fp32 staging copy on the host
        except (RuntimeError, ValueError):                                                      # legacy (non-zip) files cannot be mapped
            ckpt = torch.load(load_path, map_location="cpu", weights_only=False)
        self.start_epoch = ckpt.get("epoch", 0)
        self.step = ckpt.get("step", 0)
        sd = ckpt["state_dict"]
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
        self.model.load_state_dict(sd, strict=True)
        self.ema_shadow = ckpt.get("ema_helper")
        self.optimizer_state = ckpt.get("optimizer")                            # restored into the trainer (ddm_wavelet.py:186)
        if ema and self.ema_shadow is not None:                                # EMAHelper.ema, ddm_wavelet.py:55-60
            with torch.no_grad():
                for name, p in self.model.named_parameters():
                    if name in self.ema_shadow:
                        p.copy_(self.ema_shadow[name].to(p.device))
        if hasattr(self.model, "pack_weights"):
            self.model.pack_weights(force=True)
        print("=> loaded checkpoint '{}' (epoch {}, step {})".format(load_path, self.start_epoch, self.step))

    # ---- training (ddm_wavelet.py:200-292) -----------------------------------------------------------------------
    def make_trainer(self, **kw):
        """The training state of this model on the HIP library (wavedm_amd.training.Trainer), initialised from the current weights."""
        from .training import Trainer
        if getattr(self.config.data, "global_attn", False):
            raise NotImplementedError("training the data.global_attn model is not built (inference only, wavedm_amd/unet_global.py)")
        tr = Trainer(self.config, device=self.device, dtype=kw.pop("dtype", None), **kw)
        tr.load_state_dict(self.model.state_dict())
        if self.ema_shadow is not None:
            for k in tr.layout:
                if k in self.ema_shadow:
                    tr._view(tr.ema, k).copy_(self.ema_shadow[k].to(self.device))
        tr.step = self.step
        osd = getattr(self, "optimizer_state", None)
        if osd:                                                                 # --resume: Adam moments and step count (ddm_wavelet.py:186)
            tr.load_optimizer_state_dict(osd)
            tr.step = self.step = max(tr.step, self.step)
        tr.broadcast_state(src=0)                                               # DDP's construction-time broadcast (ddm_wavelet.py:168)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and os.environ.get("WAVEDM_GRAD_BUCKETS", "8") != "0":
            # ... and its bucketed gradient all-reduce that overlaps the backward (Trainer.allreduce_grads_overlapped); WAVEDM_GRAD_BUCKETS=0: one flat all-reduce
            try:
                nb = int(os.environ.get("WAVEDM_GRAD_BUCKETS", "8"))
            except ValueError:
                nb = 8                                                          # a malformed value must not take every rank down at start-up
            tr.enable_grad_buckets(nb)                                          # (clamped to [0, 64] there; bucket bounds depend on the model alone: equal on every rank)
        self.trainer = tr
        return tr

    def assemble_training_sample(self, x):
        """x (n, 6, H, W) in [0,1] = [degraded | ground truth] crops -> the 96-channel wavelet-domain sample of ddm_wavelet.py:218-243
        (`use_other_channels` and `use_gt_in_train` as in raindrop_wavelet.yml): [DWT(input) 48 | DWT(gt) LL 3 | DWT(gt) bands 3..47]."""
        m = self.config.model
        if not (m.use_other_channels and getattr(m, "use_gt_in_train", True)):
            raise NotImplementedError("only the use_other_channels / use_gt_in_train branch of raindrop_wavelet.yml is built")
        x = data_transform(x.to(self.device).float())
        cond = self.wavelet_dec(x[:, :3].contiguous())
        gt = self.wavelet_dec(x[:, 3:].contiguous())
        return torch.cat([cond, gt[:, :m.pred_channels], gt[:, m.other_channels_begin:]], dim=1).contiguous()

    def train_step(self, x, group=None):
        """One iteration of the reference's loop body (:208-272) on a batch of crops x (n, 6, p, p): DWT, q-sample with antithetic
        timesteps, loss, backward, gradient all-reduce, Adam, EMA.  Returns the loss as a device tensor."""
        tr = getattr(self, "trainer", None) or self.make_trainer()
        _lib.pinned_dontfork(x)                          # (a pin_memory=True loader's batch: not copy-on-write at the loader's next fork -- _lib.pinned_dontfork)
        x = x.flatten(start_dim=0, end_dim=1) if x.ndim == 5 else x
        loss = tr.train_step(self.assemble_training_sample(x), group=group)
        self.step = tr.step
        return loss

    def sync_from_trainer(self, ema=False):
        """Copy the trained (or EMA) weights back into the inference model."""
        sd = self.trainer.ema_state_dict() if ema else self.trainer.state_dict()
        self.model.load_state_dict(sd, strict=True)
        if hasattr(self.model, "pack_weights"):                                 # DiffusionUNet_Global packs per call (unet_global.py)
            self.model.pack_weights(force=True)

    def train(self, DATASET, max_steps=None):
        """ddm_wavelet.py:200-292: epochs over DATASET.get_loaders()[0]; on rank 0 the validation sheet (`restore`, :273-278) every
        `training.validation_freq` steps (every 10 when single-process, as the reference) and a checkpoint in the reference's format
        every `training.snapshot_freq` steps.  `max_steps` bounds the run (tests, benchmarks)."""
        import torch.distributed as dist
        train_loader, _ = DATASET.get_loaders()
        tr = getattr(self, "trainer", None) or self.make_trainer()
        npix = self.config.model.pred_channels * self.config.data.image_size ** 2
        rank0 = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
        for epoch in range(self.start_epoch, self.config.training.n_epochs):
            for i, (x, y, total) in enumerate(train_loader):
                loss = self.train_step(x)
                if self.step % 10 == 0 and rank0:
                    print(f"step: {self.step}, loss: {float(loss)}, loss mean: {float(loss) / npix}")
                world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
                vfreq = int(getattr(self.config.training, "validation_freq", 0) or 0) if world > 1 else 10
                if rank0 and vfreq > 0 and self.step % vfreq == 0:                                    # :273-278
                    self.sync_from_trainer()
                    _, val_loader = DATASET.get_loaders(parse_patches=False, validation=self.args.test_set)
                    self.restore(val_loader, validation=self.args.test_set, r=self.args.grid_r, epoch=epoch)
                if rank0 and (self.step % self.config.training.snapshot_freq == 0 or self.step == 1):
                    path = os.path.join(self.config.data.data_dir, "ckpts", f"{self.config.data.dataset}_epoch{epoch + 1}_ddpm.pth.tar")
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    tr.save_checkpoint(path, epoch=epoch + 1)
                if max_steps is not None and self.step >= max_steps:
                    return

    # ---- sampling ----------------------------------------------------------------------------------
    def _require_plain_unet(self, what):
        """data.global_attn builds DiffusionUNet_Global, which only the use_global=True branch of the stitched sampler can drive (it needs `total`,
        the whole image its patches attend to); the reference is equally unusable on the other paths (total=None reaches total.repeat,
        ddm_wavelet.py:482) -- say so instead of failing deep inside the sampler."""
        if getattr(self.config.data, "global_attn", False):
            raise RuntimeError(f"{what}: the model was built with data.global_attn=True (DiffusionUNet_Global); use "
                               "sample_image(..., total=<whole image>, use_global=True)")

    def sample_image(self, x_cond, x, x_other=None, last=True, patch_locs=None, patch_size=None, total=None,
                     use_global=False, use_other=False, stop_at=None):
        """ddm_wavelet.py:295-309.  `stop_at` (not in the reference; default None = every step, like the reference): a negative index k -- the caller reads
        nothing after x0_preds[k], so the steps behind it are skipped and the lists padded with None (sampling.ddim_sample).  DiffusiveRestoration.restore,
        which consumes x0_preds[-5] only (restoration.py:108), passes -5."""
        if last and stop_at is not None:
            raise ValueError("sample_image: last=True returns xs[-1], which an early stop does not compute")
        skip = self.config.diffusion.num_diffusion_timesteps // self.args.sampling_timesteps
        seq = range(0, self.config.diffusion.num_diffusion_timesteps, skip)
        if patch_locs is None:
            # ddm_wavelet.py:305-306 falls through to utils.sampling.generalized_steps (utils/sampling.py:23-44): every image is ONE patch of the
            # model's resolution and the UNet sees [x_cond | x_t] (its conv_in must be built for that width: model.use_other_channels False)
            # (generalized_steps starts from x as given whatever data.begin_from_noise says: there is no q-sample of x_cond on this path)
            self._require_plain_unet("sample_image(patch_locs=None)")
            xs = sampling.ddim_sample(self.model, x, x_cond, None, list(seq), self.betas, corners=None, max_batch=getattr(self.args, "max_batch", 64),
                                      stop_at=stop_at)
            return xs[0][-1] if last else xs
        xs = self.generalized_steps_overlapping(x, x_cond, seq, self.model, self.betas, eta=0., corners=patch_locs,
                                                p_size=patch_size, total=total, use_global=use_global,
                                                x_other=x_other, use_other=use_other, stop_at=stop_at)
        if last:
            xs = xs[0][-1]
        return xs

    def generalized_steps_overlapping(self, x, x_cond, seq, model, b, eta=0., corners=None, p_size=None,
                                      manual_batching=True, total=None, x_other=None, use_global=False, use_other=False, stop_at=None):
        """ddm_wavelet.py:437-506 on the device (eta != 0 included: :500-502, one randn_like(x) per step from the device generator)."""
        if use_global:
            if stop_at is not None:
                raise NotImplementedError("stop_at is not built for the use_global branch")
            return self._ddim_overlapping_global(x, x_cond, list(seq), model, b, corners, p_size, total, eta=eta)
        self._require_plain_unet("generalized_steps_overlapping(use_global=False)")
        if not use_other:
            x_other = None                                                      # ddm_wavelet.py:471-473: the UNet sees [x_cond | x_t] only
        if not self.config.data.begin_from_noise:                              # ddm_wavelet.py:445-447
            a = (1 - b).cumprod(dim=0)[self.num_timesteps - 1]
            x = x_cond[:, :x.shape[1]] * a.sqrt() + x * (1.0 - a).sqrt()
        grp = getattr(self, "patch_group", None)
        if grp is not None:                                                    # patch-sharded latency mode: identical start noise on every rank
            import torch.distributed as dist
            dist.broadcast(x, src=0, group=None if grp is True else grp)
        xs, x0_preds = sampling.ddim_sample(model, x, x_cond, x_other, list(seq), b, corners=corners, p_size=p_size,
                                            max_batch=getattr(self.args, "max_batch", None) or sampling.DEFAULT_MAX_BATCH, patch_group=grp, eta=eta, stop_at=stop_at)
        if self.verbose:
            for i_t, x0, xn in zip(reversed(list(seq)), x0_preds, xs[1:]):
                if x0 is None:
                    break
                print(f"t:{i_t} x0 pred:{x0.mean().item()} x next:{xn.mean().item()}")
        return xs, x0_preds

    def _ddim_overlapping_global(self, x, x_cond, seq, model, b, corners, p_size, total, eta=0.):
        """The `use_global` branch of generalized_steps_overlapping (ddm_wavelet.py:479-483): every patch is [x_cond crop | x_t crop] and the
        model also sees the whole image `total`, repeated for each patch.  Crops and concatenation are tensor plumbing; the UNet and the
        scatter-mean + DDIM update are library kernels (`wdm_ddim_update`, the one the main sampler uses)."""
        if total is None:
            raise ValueError("use_global=True needs `total`, the whole image the patches attend to")
        if not self.config.data.begin_from_noise:
            a = (1 - b).cumprod(dim=0)[self.num_timesteps - 1]
            x = x_cond[:, :x.shape[1]] * a.sqrt() + x * (1.0 - a).sqrt()
        x = _lib.require_cuda_f32(x, "x")
        x_cond = _lib.require_cuda_f32(x_cond, "x_cond")
        total = _lib.require_cuda_f32(total, "total")
        nimg, pc, H, W = x.shape
        if pc != 3:
            raise NotImplementedError("the DDIM update kernels are built for 3 prediction channels")
        p = int(p_size)
        tri = [(im, int(hi), int(wi)) for im in range(nimg) for (hi, wi) in corners]
        patches = torch.tensor(tri, dtype=torch.int32).to(self.device)
        L, h = _lib.lib(), _lib.handle(self.device.index or 0)
        abar = sampling.alpha_bar_table(b)
        seq_next = [-1] + seq[:-1]
        xs, x0_preds, xt = [x], [], x
        cond_p = torch.cat([x_cond[im:im + 1, :, hi:hi + p, wi:wi + p] for (im, hi, wi) in tri], dim=0)
        tot_p = torch.cat([total[im:im + 1] if total.shape[0] == nimg else total[:1] for (im, _, _) in tri], dim=0).contiguous()
        with torch.cuda.device(self.device):
            for i_t, j_t in zip(reversed(seq), reversed(seq_next)):
                at, at_next = abar[i_t + 1], abar[j_t + 1]
                xt_p = torch.cat([xt[im:im + 1, :, hi:hi + p, wi:wi + p] for (im, hi, wi) in tri], dim=0)
                inp = torch.cat([cond_p, xt_p], dim=1).contiguous()
                t = torch.tensor([float(i_t)], device=self.device)
                eps = torch.cat([model(inp[i:i + 8], t, tot_p[i:i + 8]) for i in range(0, len(tri), 8)], dim=0).contiguous()      # manual_batching_size 8
                x0, xn = torch.empty_like(xt), torch.empty_like(xt)
                if eta != 0.:                                                       # ddm_wavelet.py:500-502
                    c1 = eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
                    noise = torch.randn_like(xt)
                    _lib.check(L.wdm_ddim_update_eta(h, _lib.ptr(eps), _lib.ptr(patches), len(tri), p, _lib.ptr(xt), nimg, H, W, float((1 - at).sqrt()),
                                                     float(at.sqrt()), float(at_next.sqrt()), float(c1), float(((1 - at_next) - c1 ** 2).sqrt()),
                                                     _lib.ptr(noise), _lib.ptr(x0), _lib.ptr(xn), _lib.stream_ptr()))
                else:
                    _lib.check(L.wdm_ddim_update(h, _lib.ptr(eps), _lib.ptr(patches), len(tri), p, _lib.ptr(xt), nimg, H, W, float((1 - at).sqrt()),
                                                 float(at.sqrt()), float(at_next.sqrt()), float((1 - at_next).sqrt()), _lib.ptr(x0), _lib.ptr(xn),
                                                 _lib.stream_ptr()))
                x0_preds.append(x0)
                xs.append(xn)
                xt = xn
        return xs, x0_preds

    def overlapping_grid_indices(self, x_cond, output_size, r=None):
        """ddm_wavelet.py:426-435."""
        _, c, h, w = x_cond.shape
        return sampling.overlapping_grid_indices(h, w, output_size, r)

    def diffusive_restoration(self, x_cond, x_other=None, r=None, last=True, total=None, use_global=False, use_other=False, stop_at=None):
        """ddm_wavelet.py:413-424.  `stop_at`: see sample_image."""
        p_size = self.config.data.patch_size if self.config.data.wavelet_in_unet else self.config.data.image_size
        h_list, w_list = self.overlapping_grid_indices(x_cond, output_size=p_size, r=r)
        corners = [(i, j) for i in h_list for j in w_list]
        x = torch.randn((x_cond.shape[0], self.config.model.pred_channels, x_cond.shape[2], x_cond.shape[3]), device=self.device)
        return self.sample_image(x_cond, x, x_other=x_other, patch_locs=corners, last=last, patch_size=p_size,
                                 total=total, use_global=use_global, use_other=use_other, stop_at=stop_at)

    def restore(self, val_loader, validation="snow", r=None, epoch=0):
        """ddm_wavelet.py:340-411, the training loop's validation sheet: the first two items of `val_loader` are restored and
        [input | x0_preds[-5] LL + ground-truth bands | output | ground truth] of each go into one 4-column PNG
        `<image_folder>/<dataset>/<validation>/<y>_output_epoch<epoch>.png`.  Returns that path."""
        cfg = self.config
        if not (cfg.data.wavelet and not cfg.data.wavelet_in_unet and cfg.model.use_other_channels):
            raise NotImplementedError("DenoisingDiffusion_Wavelet.restore: only the raindrop_wavelet.yml branch is accelerated")
        from . import imageio
        image_folder = os.path.join(self.args.image_folder, cfg.data.dataset, validation)
        pc, ob = cfg.model.pred_channels, cfg.model.other_channels_begin
        tiles, y = [], None
        with torch.no_grad():
            for i, (x, y, total) in enumerate(val_loader):
                print(f"starting processing from image {y}")
                x = x.flatten(start_dim=0, end_dim=1) if x.ndim == 5 else x
                x = x.to(self.device).float().contiguous()
                x_all = data_transform(x)
                inp, gt = x[:, :3].contiguous(), x[:, 3:].contiguous()
                x_cond = self.wavelet_dec(x_all[:, :3].contiguous())
                x_gt = self.wavelet_dec(x_all[:, 3:].contiguous())
                hf_wav = self.wavelet_dec(data_transform(self.generator(inp)).contiguous())
                _, x0_preds = self.diffusive_restoration(x_cond, x_other=hf_wav[:, ob:].contiguous(), r=r, last=False,
                                                         use_global=False, use_other=True,
                                                         stop_at=-5 if getattr(self.args, "early_stop", True) else None)      # only x_output_list[1][-5] is read (ddm_wavelet.py:378)
                pred = x0_preds[-5]
                rec = lambda lo, hi: inverse_data_transform(self.wavelet_rec(torch.cat([lo[:, :pc], hi[:, pc:]], dim=1).contiguous()))
                x_output, hrgt = rec(pred, hf_wav), rec(pred, x_gt)
                H, W = x_output.shape[-2:]
                print("psnr", imageio.psnr_from_sums(imageio.sqdiff(gt, x_output), H, W)[0][0])
                tiles += [inp, hrgt, x_output, gt]            # IDWT(DWT(x)) == x: the "cond" tile is the input
                if i == 1:
                    break
        if not tiles:
            return None
        grid = imageio.make_grid(torch.cat(tiles, dim=0), nrow=4)
        path = os.path.join(image_folder, f"{y}_output_epoch{epoch}.png")
        os.makedirs(image_folder, exist_ok=True)
        from PIL import Image
        Image.fromarray(imageio.to_u8_hwc(grid)[0].cpu().numpy()).save(path)
        return path

    # ---- batched independent crops (BASELINE.json configs 0-3) --------------------------------------------
    def restore_batch(self, rainy01, x_T, hfrm_out01=None, keep=-5, early_stop=False):
        """B independent patch_size x patch_size crops: DWT -> S-step DDIM -> IDWT, all on the GPU.

        rainy01 (B,3,4R,4R) in [0,1]; x_T (B,3,R,R) start noise; returns the restored (B,3,4R,4R) in [0,1] built
        from x0_preds[keep] like restoration.py:108-134, plus (xs[-1], x0_preds[keep])."""
        self._require_plain_unet("restore_batch")
        x_cond = self.wavelet_dec.forward_affine(rainy01)                        # DWT(2x - 1): data_transform folded into the kernel
        hf = self.generator(rainy01) if hfrm_out01 is None else hfrm_out01
        hf_wav = x_cond if hf is rainy01 else self.wavelet_dec.forward_affine(hf)   # identity stand-in: the same tensor, not a second pass
        ob = self.config.model.other_channels_begin
        x_other = hf_wav[:, ob:].contiguous()
        skip = self.config.diffusion.num_diffusion_timesteps // self.args.sampling_timesteps
        seq = list(range(0, self.config.diffusion.num_diffusion_timesteps, skip))
        xs, x0_preds = sampling.ddim_sample(self.model, x_T, x_cond, x_other, seq, self.betas, corners=None,
                                            max_batch=getattr(self.args, "max_batch", 64), keep={keep, -1},     # x0_preds[keep], xs[-1] (xs[keep] after an early stop)
                                            stop_at=keep if early_stop else None)      # early_stop: skip the discarded tail
        pc = self.config.model.pred_channels
        x0 = x0_preds[keep]
        out = self.wavelet_rec.compose(x0, hf_wav, pc)                            # IDWT of [x0 low bands | HFRM bands], clamp((x + 1) / 2) on the way out
        return out, (xs[-1] if xs[-1] is not None else xs[len(seq) + keep + 1]), x0
