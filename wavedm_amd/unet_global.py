"""DiffusionUNet_Global -- the optional `data.global_attn` model of the reference (`models/unet.py:397-636`) on the HIP library.

Same trunk as `DiffusionUNet` plus a second, small branch over the whole image `x_global`: `global_conv_in`, one strided (transposed)
4x4 convolution per level, and an `Attn_Global` after every level in which the patch features (queries: 2x2 pixel blocks) attend to
the 8x8-pooled tokens of the whole-image map.  The flag is off in every YAML the reference ships and the model only
runs when the last level keeps its channel count (its last `down_global.attn` normalises the un-convolved whole-image map with the last
level's GroupNorm, unet.py:609-610); it is built here for completeness of the config surface, composed from the block
entry points of `include/wavedm.h` (ResnetBlock, AttnBlock, 3x3 / 1x1 / down / up convolutions: the executors the main model runs, in
the chosen compute dtype) and the four fp32 operators of `csrc/global_attn.hip`.  NCHW fp32 at every block boundary: this variant is
parity-checked against the reference's own output (tests/golden/global.npz), not tuned.  No CPU / PyTorch fallback: every arithmetic
step is a kernel of libwavedm_hip.so; torch only allocates, slices and concatenates.

Reference quirks kept: both attention inputs are normalised with `norm_patch` (`norm_global` is registered, loaded and unused, :433-434);
the middle starts from `hs[-1]`, so the output of the last level's global attention on the way down is discarded (:612)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .procedural import unet_global_param_shapes
from .unet import _Node, resolve_dtype


class DiffusionUNet_Global(nn.Module):
    def __init__(self, config, dtype=None):
        super().__init__()
        if not getattr(config.data, "global_attn", False):
            raise ValueError("DiffusionUNet_Global is the data.global_attn: True model")
        self.config = config
        m = config.model
        self.ch, self.temb_ch = int(m.ch), int(m.ch) * 4
        self.ch_mult = tuple(m.ch_mult)
        if len(self.ch_mult) > 1 and self.ch_mult[-1] != self.ch_mult[-2]:
            raise ValueError("DiffusionUNet_Global: ch_mult[-1] must equal ch_mult[-2] (the reference model, unet.py:609-610, normalises the "
                             "whole-image map of level L-2 with the GroupNorm of level L-1)")
        self.num_res_blocks = int(m.num_res_blocks)
        self.attn_resolutions = list(m.attn_resolutions)
        self.resolution = int(config.data.image_size)
        self.in_channels = int(m.in_channels) * 2 if getattr(config.data, "conditional", True) else int(m.in_channels)
        self.global_in_channels = int(m.in_channels)
        self.out_ch = int(m.out_ch)
        self._dtype_code = resolve_dtype(config, dtype)
        self._scratch = None
        for key, shape in unet_global_param_shapes(config).items():            # the reference's registration order (golden: global.npz "names")
            self._register(key, shape)

    def _register(self, key, shape):
        parts = key.split(".")
        node = self
        for comp in parts[:-1]:
            if comp not in node._modules:
                node.add_module(comp, _Node())
            node = node._modules[comp]
        p = torch.empty(shape, dtype=torch.float32)
        if len(shape) > 1:
            nn.init.kaiming_uniform_(p, a=5 ** 0.5)
        elif parts[-1] == "weight":
            p.fill_(1.0)
        else:
            p.zero_()
        node.register_parameter(parts[-1], nn.Parameter(p))

    @property
    def module(self):
        return self

    # ---- block calls -------------------------------------------------------------------------------------------------
    def _sd(self):
        return {k: v for k, v in self.named_parameters()}

    def _scr(self, dev):
        if self._scratch is None or self._scratch.device != dev:
            self._scratch = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        return self._scratch

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def _resblock(self, sd, name, x0, x1, temb):
        L, h = _lib.lib(), _lib.handle(x0.device.index or 0)
        g = lambda k: sd.get(f"{name}.{k}")
        p = _lib.ResblockParams()
        p.cin, p.cout = g("conv1.weight").shape[1], g("conv1.weight").shape[0]
        for f, k in (("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"), ("conv1_w", "conv1.weight"), ("conv1_b", "conv1.bias"),
                     ("temb_w", "temb_proj.weight"), ("temb_b", "temb_proj.bias"), ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"),
                     ("conv2_w", "conv2.weight"), ("conv2_b", "conv2.bias"), ("nin_w", "nin_shortcut.weight"), ("nin_b", "nin_shortcut.bias")):
            t = g(k)
            setattr(p, f, t.data_ptr() if t is not None else None)
        B, c0, H, W = x0.shape
        c1 = x1.shape[1] if x1 is not None else 0
        y = torch.empty(B, p.cout, H, W, device=x0.device)
        sc = self._scr(x0.device)
        _lib.check(L.wdm_resblock_forward(h, C.byref(p), self._p(x0), c0, self._p(x1), c1, self._p(temb), temb.shape[0], temb.shape[1], B, H, W, self._p(y),
                                          self._dtype_code, self._p(sc), sc.numel(), _lib.stream_ptr()))
        return y

    def _attn(self, sd, name, x):
        L, h = _lib.lib(), _lib.handle(x.device.index or 0)
        p = _lib.AttnParams()
        p.c = x.shape[1]
        for f, k in (("norm_w", "norm.weight"), ("norm_b", "norm.bias"), ("q_w", "q.weight"), ("q_b", "q.bias"), ("k_w", "k.weight"), ("k_b", "k.bias"),
                     ("v_w", "v.weight"), ("v_b", "v.bias"), ("proj_w", "proj_out.weight"), ("proj_b", "proj_out.bias")):
            setattr(p, f, sd[f"{name}.{k}"].data_ptr())
        B, Cc, H, W = x.shape
        y = torch.empty_like(x)
        sc = self._scr(x.device)
        _lib.check(L.wdm_attn_forward(h, C.byref(p), self._p(x), B, H, W, self._p(y), self._dtype_code, self._p(sc), sc.numel(), _lib.stream_ptr()))
        return y

    def _conv(self, sd, name, x, mode):
        """mode: 0 conv3x3 s1 p1, 1 Downsample, 2 Upsample, 3 conv1x1 (wdm_conv_forward)."""
        L, h = _lib.lib(), _lib.handle(x.device.index or 0)
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        B, cin, H, W = x.shape
        Ho, Wo = (H // 2, W // 2) if mode == 1 else (H * 2, W * 2) if mode == 2 else (H, W)
        y = torch.empty(B, w.shape[0], Ho, Wo, device=x.device)
        sc = self._scr(x.device)
        _lib.check(L.wdm_conv_forward(h, self._p(w), self._p(b), cin, w.shape[0], mode, self._p(x), B, H, W, self._p(y), self._dtype_code, self._p(sc),
                                      sc.numel(), _lib.stream_ptr()))
        return y

    def _conv_direct(self, sd, name, x, k, stride, pad, groups=1, transposed=False):
        L, h = _lib.lib(), _lib.handle(x.device.index or 0)
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        B, cin, H, W = x.shape
        cout = w.shape[1] if transposed else w.shape[0]
        Ho = (H - 1) * stride - 2 * pad + k if transposed else (H + 2 * pad - k) // stride + 1
        Wo = (W - 1) * stride - 2 * pad + k if transposed else (W + 2 * pad - k) // stride + 1
        y = torch.empty(B, cout, Ho, Wo, device=x.device)
        _lib.check(L.wdm_conv2d_direct(h, self._p(x), self._p(w), self._p(b), B, cin, H, W, cout, k, stride, pad, groups, 1 if transposed else 0, self._p(y),
                                       _lib.stream_ptr()))
        return y

    def _groupnorm(self, sd, name, x, silu=False):
        L, h = _lib.lib(), _lib.handle(x.device.index or 0)
        B, Cc, H, W = x.shape
        y = torch.empty_like(x)
        _lib.check(L.wdm_groupnorm(h, self._p(x), self._p(sd[name + ".weight"]), self._p(sd[name + ".bias"]), B, Cc, H, W, 1e-6, 1 if silu else 0, self._p(y),
                                   _lib.stream_ptr()))
        return y

    def _attn_global(self, sd, name, x_patch, x_global):
        """Attn_Global.forward (unet.py:432-462)."""
        L, h = _lib.lib(), _lib.handle(x_patch.device.index or 0)
        B, Cc, H, W = x_patch.shape
        lp, gp = sd[name + ".q.weight"].shape[-1], sd[name + ".k.weight"].shape[-1]
        hn = self._groupnorm(sd, name + ".norm_patch", x_patch)
        gn = self._groupnorm(sd, name + ".norm_patch", x_global)              # :434: norm_patch, not norm_global
        q = self._conv_direct(sd, name + ".q", hn, lp, lp, 0)
        k = self._conv_direct(sd, name + ".k", gn, gp, gp, 0, groups=Cc)
        v = self._conv_direct(sd, name + ".v", gn, gp, gp, 0, groups=Cc)
        nq, nk = q.shape[2] * q.shape[3], k.shape[2] * k.shape[3]
        o = torch.empty_like(q)
        _lib.check(L.wdm_cross_attention(h, self._p(q), self._p(k), self._p(v), B, Cc, nq, nk, self._p(o), _lib.stream_ptr()))
        # proj_out works on the query grid (H / lp: 4x4 at the deepest level of the fixture), below the 8x8 tile of the MFMA conv kernels
        o = self._conv(sd, name + ".proj_out", o, 3) if (o.shape[2] % 8 == 0 and o.shape[3] % 8 == 0 and Cc % 32 == 0) else \
            self._conv_direct(sd, name + ".proj_out", o, 1, 1, 0)
        y = torch.empty_like(x_patch)
        _lib.check(L.wdm_upsample_add(h, self._p(x_patch), self._p(o), B, Cc, H, W, lp, self._p(y), _lib.stream_ptr()))
        return y

    # ---- forward (unet.py:585-636) ---------------------------------------------------------------------------------------
    def forward(self, x, t, x_global):
        x = _lib.require_cuda_f32(x, "x")
        x_global = _lib.require_cuda_f32(x_global, "x_global")
        assert x.shape[2] == x.shape[3] == self.resolution and x.shape[1] == self.in_channels and x_global.shape[1] == self.global_in_channels
        dev = x.device
        sd = self._sd()
        if any(p.device != dev for p in sd.values()):
            raise RuntimeError("DiffusionUNet_Global: move the module to the GPU first (.to('cuda')); there is no CPU path")
        L, h = _lib.lib(), _lib.handle(dev.index or 0)
        nres, nrb = len(self.ch_mult), self.num_res_blocks
        with torch.cuda.device(dev), torch.no_grad():
            t = t.to(device=dev, dtype=torch.float32).contiguous()
            if t.numel() not in (1, x.shape[0]):
                raise ValueError("t must hold one timestep or one per image")
            temb = torch.empty(t.numel(), self.temb_ch, device=dev)
            sc = self._scr(dev)
            _lib.check(L.wdm_temb_forward(h, self._p(t), t.numel(), self.ch, self._p(sd["temb.dense.0.weight"]), self._p(sd["temb.dense.0.bias"]),
                                          self._p(sd["temb.dense.1.weight"]), self._p(sd["temb.dense.1.bias"]), self._p(temb), self._p(sc), sc.numel(),
                                          _lib.stream_ptr()))
            res = self.resolution
            # the two input convolutions have 3 / 6 input channels (no 32-channel K slab to feed the MFMA kernels): direct kernels
            hg = self._conv_direct(sd, "global_conv_in", x_global, 3, 1, 1)
            hs = [self._conv_direct(sd, "conv_in", x, 3, 1, 1)]
            hcur = hs[-1]
            for l in range(nres):
                for b in range(nrb):
                    hcur = self._resblock(sd, f"down.{l}.block.{b}", hcur, None, temb)
                    if res in self.attn_resolutions:
                        hcur = self._attn(sd, f"down.{l}.attn.{b}", hcur)
                    hs.append(hcur)
                if l != nres - 1:
                    hcur = self._conv(sd, f"down.{l}.downsample.conv", hcur, 1)
                    hs.append(hcur)
                    res //= 2
                    hg = self._conv_direct(sd, f"down_global.{l}.conv", hg, 4, 2, 1)
                hcur = self._attn_global(sd, f"down_global.{l}.attn", hcur, hg)
            hcur = hs[-1]                                                        # :612 (the last global attention's output is dropped)
            hcur = self._resblock(sd, "mid.block_1", hcur, None, temb)
            hcur = self._attn(sd, "mid.attn_1", hcur)
            hcur = self._resblock(sd, "mid.block_2", hcur, None, temb)
            for l in reversed(range(nres)):
                for b in range(nrb + 1):
                    hcur = self._resblock(sd, f"up.{l}.block.{b}", hcur, hs.pop(), temb)
                    if res in self.attn_resolutions:
                        hcur = self._attn(sd, f"up.{l}.attn.{b}", hcur)
                if l != 0:
                    hcur = self._conv(sd, f"up.{l}.upsample.conv", hcur, 2)
                    res *= 2
                    hg = self._conv_direct(sd, f"up_global.{l}.conv", hg, 4, 2, 1, transposed=True)
                hcur = self._attn_global(sd, f"up_global.{l}.attn", hcur, hg)
            hcur = self._groupnorm(sd, "norm_out", hcur, silu=True)
            return self._conv(sd, "conv_out", hcur, 0)
