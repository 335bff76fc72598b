"""Training step of the wavelet-domain UNet on the HIP library.

`Trainer(config)` owns five flat fp32 device buffers -- parameters, gradients, Adam m / v, EMA shadow -- in the layout the library
reports (`wdm_trainer_param_info`), and runs the body of the reference's training loop (`models/ddm_wavelet.py:259-272`):

    loss = trainer.loss_and_grads(x0, t, e)     # noise_estimation_loss (:108-124) forward + backward
    trainer.allreduce_grads()                   # what DistributedDataParallel does in the reference (:168), one RCCL all-reduce
    trainer.optimizer_step()                    # torch.optim.Adam (utils/optimize.py:5-8) + EMAHelper.update (:48-53)

`state_dict()` / `load_state_dict()` use the reference's keys and shapes, `ema_state_dict()` is `EMAHelper.state_dict()`; a checkpoint
written by `save_checkpoint` has the reference's dict format (ddm_wavelet.py:282-292) and loads in `DenoisingDiffusion_Wavelet`."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import torch

from . import _lib, sampling
from .unet import _make_config, reference_param_order, resolve_dtype


class Trainer:
    def __init__(self, config, device=None, dtype=None, lr=None, betas=(0.9, 0.999), eps=None, weight_decay=None, ema_mu=0.9999, use_mse=None):
        self.config = config
        self.device = torch.device(device if device is not None else getattr(config, "device", "cuda:0"))
        if self.device.type != "cuda":
            raise RuntimeError("wavedm_amd.Trainer runs on MI355X only (no CPU path)")
        self._dtype_code = resolve_dtype(config, dtype)
        if self._dtype_code in (_lib.WDM_F32X3, _lib.WDM_F16):      # training has two modes: f32x3 / f16 (inference modes) train in exact fp32
            self._dtype_code = _lib.WDM_F32
        opt = getattr(config, "optim", None)
        self.lr = float(lr if lr is not None else getattr(opt, "lr", 4e-5))
        self.eps = float(eps if eps is not None else getattr(opt, "eps", 1e-8))
        self.weight_decay = float(weight_decay if weight_decay is not None else getattr(opt, "weight_decay", 0.0))
        self.betas, self.ema_mu = (float(betas[0]), float(betas[1])), float(ema_mu)
        if opt is not None and (getattr(opt, "optimizer", "Adam") != "Adam" or getattr(opt, "amsgrad", False)):
            raise NotImplementedError("wavedm_amd.Trainer implements optim.optimizer: Adam with amsgrad: False (utils/optimize.py:6-8, raindrop_wavelet.yml)")
        if float(getattr(getattr(config, "model", None), "dropout", 0.0) or 0.0) != 0.0:
            raise NotImplementedError("wavedm_amd.Trainer: model.dropout != 0 is not built (raindrop_wavelet.yml trains with dropout 0.0; "
                                      "the backward pass in csrc/train_unet.hip has no dropout mask)")
        # training.use_mse (ddm_wavelet.py:263-266): back-propagate the x0-space loss instead of the noise-space one
        self.use_mse = bool(use_mse if use_mse is not None else getattr(getattr(config, "training", None), "use_mse", False))
        L = _lib.lib()
        self._cfg = _make_config(config, self._dtype_code)
        t = C.c_void_p()
        _lib.check(L.wdm_trainer_create(None, C.byref(self._cfg), C.byref(t)))
        self._t = t
        self.layout = OrderedDict()                     # name -> (offset, shape)
        name, ndim, shape, off = C.c_char_p(), C.c_int(), (C.c_int64 * 4)(), C.c_int64()
        for i in range(L.wdm_trainer_num_params(t)):
            _lib.check(L.wdm_trainer_param_info(t, i, C.byref(name), C.byref(ndim), C.byref(shape), C.byref(off)))
            self.layout[name.value.decode()] = (int(off.value), tuple(int(shape[k]) for k in range(ndim.value)))
        n = int(L.wdm_trainer_num_floats(t))
        self._n_floats = n
        with torch.cuda.device(self.device):
            self.params = torch.zeros(n, device=self.device)
            self.grads = torch.zeros(n, device=self.device)
            self.exp_avg = torch.zeros(n, device=self.device)
            self.exp_avg_sq = torch.zeros(n, device=self.device)
            self.ema = torch.zeros(n, device=self.device)
        _lib.check(L.wdm_trainer_set_objective(t, 1 if self.use_mse else 0))
        _lib.check(L.wdm_trainer_set_buffers(t, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), _lib.ptr(self.ema)))
        betas_t = sampling.get_beta_schedule(beta_schedule=config.diffusion.beta_schedule, beta_start=config.diffusion.beta_start,
                                             beta_end=config.diffusion.beta_end, num_diffusion_timesteps=config.diffusion.num_diffusion_timesteps)
        self.betas_t = torch.from_numpy(betas_t).float().to(self.device)
        self.num_timesteps = int(self.betas_t.shape[0])
        self.step = 0
        self._ws = None
        m = config.model
        self._c_t0 = int(m.in_channels)                 # x0 = [x_cond (in_channels) | x_tar (pred_channels) | x_other]
        self._loss = torch.zeros(1, device=self.device)

    def __del__(self):
        try:
            if getattr(self, "_t", None):
                _lib.lib().wdm_trainer_destroy(self._t)
                self._t = None
        except Exception:
            pass

    # ---- parameters in the reference's naming ----------------------------------------------------------------------
    def _view(self, flat, name):
        off, shape = self.layout[name]
        n = 1
        for v in shape:
            n *= v
        return flat[off:off + n].view(shape)

    def state_dict(self):
        return OrderedDict((k, self._view(self.params, k).clone()) for k in self.layout)

    def ema_state_dict(self):
        return OrderedDict((k, self._view(self.ema, k).clone()) for k in self.layout)

    def grad_dict(self):
        return OrderedDict((k, self._view(self.grads, k).clone()) for k in self.layout)

    def load_state_dict(self, sd, strict=True, init_ema=True):
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
        missing = [k for k in self.layout if k not in sd]
        extra = [k for k in sd if k not in self.layout]
        if strict and (missing or extra):
            raise RuntimeError(f"Trainer.load_state_dict: missing {missing[:4]}, unexpected {extra[:4]}")
        for k in self.layout:
            if k in sd:
                self._view(self.params, k).copy_(sd[k].to(self.device, torch.float32))
        if init_ema:
            self.ema.copy_(self.params)                 # EMAHelper.register (ddm_wavelet.py:40-46)

    # ---- one step ----------------------------------------------------------------------------------------------------
    def loss_and_grads(self, x0, t, e, return_output=False):
        """x0 (B, 96, R, R) wavelet-domain [x_cond | x_tar | x_other]; t (B,) long; e (B, 3, R, R).  Fills self.grads; returns the
        loss as a 0-dim device tensor (and the network output when asked)."""
        x0 = _lib.require_cuda_f32(x0, "x0")
        e = _lib.require_cuda_f32(e, "e")
        B, Cc, R, _ = x0.shape
        t = t.to(self.device)
        a = (1 - self.betas_t).cumprod(dim=0).index_select(0, t.long())          # ddm_wavelet.py:109
        sa, s1m = a.sqrt().contiguous(), (1.0 - a).sqrt().contiguous()
        tf = t.float().contiguous()
        out = torch.empty(B, int(self.config.model.out_ch), R, R, device=self.device) if return_output else None
        with torch.cuda.device(self.device):
            # workspace = every saved activation + its gradient + operand transposes of the largest layer.  It is sized from the batch
            # (288 GB of HBM: generosity is cheap) and doubled on demand: the library reports exhaustion as an error, never overruns.
            # ... plus the forward AND dgrad weight layouts of every conv, packed at the start of the step and kept for its whole length (train_unet.hip: pack_region):
            # about twice the parameter bytes in the model dtype -- without this term small batches took the double-and-retry path on every first step
            dsize = 2 if self._dtype_code == _lib.WDM_BF16 else 4
            need = B * 96 * R * R * 4 * 160 + 2 * self._n_floats * dsize + (1 << 28)
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            for attempt in range(4):
                ws = self._ws
                rc = _lib.lib().wdm_trainer_step(self._t, _lib.ptr(x0), _lib.ptr(tf), _lib.ptr(sa), _lib.ptr(s1m), _lib.ptr(e), B, self._c_t0, _lib.ptr(self._loss),
                                                 _lib.ptr(out) if out is not None else None, _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
                if rc == _lib.WDM_ENOMEM and attempt < 3:
                    torch.cuda.synchronize(self.device)
                    n = ws.numel() * 2
                    self._ws = ws = None
                    self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
                    continue
                _lib.check(rc)
                break
        return (self._loss[0], out) if return_output else self._loss[0]

    def allreduce_grads(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.grads, op=dist.ReduceOp.SUM, group=group)
            self.grads.div_(dist.get_world_size(group))              # DDP averages

    # ---- gradient all-reduce in buckets that overlap the backward (what DistributedDataParallel does for the reference, ddm_wavelet.py:168) ----------------
    def enable_grad_buckets(self, n: int = 8):
        """The next loss_and_grads calls record an event behind the last launch that writes each of (at most) n buckets of the flat gradient buffer --
        it fills from its end while the backward runs -- so that allreduce_grads_overlapped can start a bucket's all-reduce before the backward is over.
        n = 0 switches it off; n is clamped to the C API's limit of 64 buckets."""
        import ctypes as C
        n = max(0, min(64, int(n)))
        with torch.cuda.device(self.device):
            self._gev = [torch.cuda.Event() for _ in range(int(n))]
            for ev in self._gev:
                ev.record()                                           # torch creates the handle on first use
            arr = (C.c_void_p * max(1, len(self._gev)))(*[C.c_void_p(int(ev.cuda_event)) for ev in self._gev])
            _lib.check(_lib.lib().wdm_trainer_set_grad_events(self._t, arr, len(self._gev)))
            self._comm_stream = torch.cuda.Stream(device=self.device) if self._gev else None

    def grad_buckets(self):
        """[(lo, hi)] in elements of self.grads, in the order the backward completes them (the last step's cut)."""
        import ctypes as C
        n = len(getattr(self, "_gev", []) or [])
        if n == 0:
            return []
        bounds, nb = (C.c_int64 * (n + 1))(), C.c_int()
        _lib.check(_lib.lib().wdm_trainer_grad_buckets(self._t, bounds, n + 1, C.byref(nb)))
        return [(int(bounds[k + 1]), int(bounds[k])) for k in range(nb.value)]

    def allreduce_grads_overlapped(self, group=None):
        """Call right behind loss_and_grads (which only ENQUEUES the step): bucket k's all-reduce waits for its event on a side stream and runs while
        the main stream is still in the backward; the embedding-MLP / temb_proj gradients, final only at the end of the step, follow the whole step.
        Same sums as allreduce_grads (a bucket is a slice of the same buffer)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            return
        buckets = self.grad_buckets()
        if not buckets:
            return self.allreduce_grads(group)
        main, side = torch.cuda.current_stream(self.device), self._comm_stream
        works = []
        with torch.cuda.stream(side):
            for k, (lo, hi) in enumerate(buckets):
                side.wait_event(self._gev[k])
                works.append(dist.all_reduce(self.grads[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))
            side.wait_stream(main)                                    # the rest is final when the whole step is
            lo_all, hi_all = buckets[-1][0], buckets[0][1]
            if lo_all > 0:
                works.append(dist.all_reduce(self.grads[:lo_all], op=dist.ReduceOp.SUM, group=group, async_op=True))
            if hi_all < self.grads.numel():
                works.append(dist.all_reduce(self.grads[hi_all:], op=dist.ReduceOp.SUM, group=group, async_op=True))
            for w in works:
                w.wait()
        main.wait_stream(side)
        self.grads.div_(dist.get_world_size(group))

    def optimizer_step(self):
        self.step += 1
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().wdm_trainer_adam_ema(self._t, self.step, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.ema_mu,
                                                       _lib.stream_ptr()))

    def train_step(self, x0, group=None, generator=None):
        """The body of the reference's loop for one batch of wavelet-domain samples x0 (B,96,R,R): noise, antithetic timesteps
        (ddm_wavelet.py:249-256), loss, backward, all-reduce, Adam, EMA.  Returns the loss (device tensor)."""
        n = x0.shape[0]
        e = torch.randn((n, int(self.config.model.out_ch)) + tuple(x0.shape[2:]), device=self.device, generator=generator)
        t = torch.randint(low=0, high=self.num_timesteps, size=(n // 2 + 1,), device=self.device, generator=generator)
        t = torch.cat([t, self.num_timesteps - t - 1], dim=0)[:n]
        loss = self.loss_and_grads(x0, t, e)
        if getattr(self, "_gev", None):
            self.allreduce_grads_overlapped(group)
        else:
            self.allreduce_grads(group)
        self.optimizer_step()
        return loss

    # ---- optimizer state in torch.optim.Adam's state_dict layout (what the reference saves and restores, ddm_wavelet.py:186, :288) ---------
    def param_order(self):
        """Parameter names in the reference's `model.parameters()` order: torch.optim indexes its state by that position."""
        return [k for k, _ in reference_param_order([(k, v[1]) for k, v in self.layout.items()]) if k in self.layout]

    def optimizer_state_dict(self):
        names = self.param_order()
        state = {i: {"step": torch.tensor(float(self.step)), "exp_avg": self._view(self.exp_avg, k).detach().cpu().clone(),
                     "exp_avg_sq": self._view(self.exp_avg_sq, k).detach().cpu().clone()} for i, k in enumerate(names)}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, osd):
        """Accepts torch.optim.Adam.state_dict() (the reference's checkpoints and this trainer's) and this trainer's round-1 flat format."""
        if not osd:
            return False
        if "state" in osd and "param_groups" in osd:
            names = self.param_order()
            st = osd["state"]
            if len(st) == 0:
                return False
            if len(st) != len(names):
                raise RuntimeError(f"optimizer state holds {len(st)} parameters, the model {len(names)}")
            step = None
            for i, k in enumerate(names):
                e = st[i]
                if tuple(e["exp_avg"].shape) != self.layout[k][1]:
                    raise RuntimeError(f"optimizer state {i} has shape {tuple(e['exp_avg'].shape)}, parameter {k} {self.layout[k][1]}")
                self._view(self.exp_avg, k).copy_(e["exp_avg"].to(self.device, torch.float32))
                self._view(self.exp_avg_sq, k).copy_(e["exp_avg_sq"].to(self.device, torch.float32))
                step = int(float(e["step"])) if step is None else step
            g = osd["param_groups"][0]
            self.lr, self.eps, self.weight_decay = float(g["lr"]), float(g["eps"]), float(g["weight_decay"])
            self.betas = (float(g["betas"][0]), float(g["betas"][1]))
            if g.get("amsgrad", False):
                raise NotImplementedError("amsgrad optimizer state")
            if step is not None:
                self.step = step
            return True
        if "exp_avg" in osd and "exp_avg_sq" in osd:                # round-1 format of this repository: the two flat buffers
            self.exp_avg.copy_(osd["exp_avg"].to(self.device))
            self.exp_avg_sq.copy_(osd["exp_avg_sq"].to(self.device))
            self.step = int(osd.get("step", self.step))
            return True
        raise RuntimeError("unrecognised optimizer state in the checkpoint")

    def save_checkpoint(self, path, epoch=0):
        """The reference's checkpoint dict (ddm_wavelet.py:282-292): its own `load_ddm_ckpt` (:180-190) reads this file."""
        torch.save({"epoch": epoch, "step": self.step, "state_dict": {k: v.cpu() for k, v in self.state_dict().items()},
                    "optimizer": self.optimizer_state_dict(),
                    "ema_helper": {k: v.cpu() for k, v in self.ema_state_dict().items()}, "params": None, "config": None}, path)

    def broadcast_state(self, src=0, group=None):
        """What DistributedDataParallel does at construction (ddm_wavelet.py:168): every rank starts from rank `src`'s parameters (and here
        also its EMA shadow and Adam moments, so that a resumed run is identical on every rank)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            for buf in (self.params, self.ema, self.exp_avg, self.exp_avg_sq):
                dist.broadcast(buf, src=src, group=group)
            st = torch.tensor([self.step], device=self.device, dtype=torch.int64)
            dist.broadcast(st, src=src, group=group)
            self.step = int(st.item())
