"""DiffusionUNet -- drop-in for the reference's `models/unet.py:196-395` whose forward runs on
libwavedm_hip.so.

* same constructor (`DiffusionUNet(config)` with the YAML namespace), same `state_dict()` keys
  and tensor shapes (conv OIHW fp32, Linear [out,in]) so reference checkpoints load with
  `load_state_dict(strict=True)`;
* `forward(x, t)`: x (B, 96, R, R) fp32 NCHW on the GPU, t (n,) float with n in {1, B}
  -> (B, 3, R, R) fp32 NCHW, exactly the reference call;
* compute dtype: `config.model.hip_dtype` / env WAVEDM_DTYPE in {"bf16" (default: throughput), "f16" (the bf16 kernels on fp16 operands:
  the bf16 speed with three more mantissa bits, weights must lie inside +-65504), "f32" (exact-fp32 parity mode), "f32x3" (fast parity mode:
  fp32 tensors, every product as three bf16 MFMAs on hi/lo-split operands)}.

The parameter tree is generated from the library's own parameter table (wdm_unet_param_info), so
Python and C++ cannot disagree about names or shapes."""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
import torch.nn as nn

from . import _lib
from .procedural import unet_in_channels


class _Node(nn.Module):
    """Anonymous container; children are named after the state_dict path components."""


def _make_config(config, dtype_code):
    m = config.model
    cfg = _lib.UNetConfig()
    cfg.ch = int(m.ch)
    mult = list(m.ch_mult)
    cfg.n_levels = len(mult)
    for i, v in enumerate(mult):
        cfg.ch_mult[i] = int(v)
    cfg.num_res_blocks = int(m.num_res_blocks)
    ar = list(m.attn_resolutions)
    cfg.n_attn_res = len(ar)
    for i, v in enumerate(ar):
        cfg.attn_resolutions[i] = int(v)
    cfg.in_channels = unet_in_channels(config)
    cfg.out_ch = int(m.out_ch)
    cfg.resolution = int(config.data.image_size)
    cfg.resamp_with_conv = 1 if m.resamp_with_conv else 0
    cfg.dtype = dtype_code
    return cfg


def resolve_dtype(config=None, dtype=None):
    name = dtype or getattr(getattr(config, "model", None), "hip_dtype", None) or os.environ.get("WAVEDM_DTYPE", "bf16")
    if name not in _lib.DTYPES:
        raise ValueError(f"unknown compute dtype {name!r} (use 'bf16', 'f16', 'f32' or 'f32x3')")
    return _lib.DTYPES[name]


def reference_param_order(table):
    """[(key, shape), ...] in the library's order -> the reference's `named_parameters()` order (unet.py:81-105: norm1, conv1, temb_proj,
    norm2, conv2, nin_shortcut; levels ascending).  torch.optim state dicts index parameters by position (ddm_wavelet.py:186, :288), so
    the module tree and the trainer's optimizer state use this order.  The library lists the temb_proj layers last (they are one
    concatenated matrix there) and the up path in execution order."""
    temb = {k: s for k, s in table if ".temb_proj." in k}
    ordered = []
    for k, s in table:
        if ".temb_proj." in k:
            continue
        ordered.append((k, s))
        if k.endswith(".conv1.bias"):
            blk = k[:-len(".conv1.bias")]
            for leaf in ("weight", "bias"):
                kk = f"{blk}.temb_proj.{leaf}"
                if kk in temb:
                    ordered.append((kk, temb.pop(kk)))
    assert not temb, f"unplaced temb_proj parameters: {list(temb)[:3]}"
    ups = [i for i, (k, _) in enumerate(ordered) if k.startswith("up.")]
    if ups:
        seg = ordered[ups[0]:ups[-1] + 1]
        assert all(k.startswith("up.") for k, _ in seg)
        seg.sort(key=lambda e: int(e[0].split(".")[1]))              # stable: up.0, up.1, ... each in its own order
        ordered[ups[0]:ups[-1] + 1] = seg
    return ordered


class DiffusionUNet(nn.Module):
    def __init__(self, config, dtype=None):
        super().__init__()
        d = config.data
        if getattr(d, "global_attn", False):
            raise ValueError("data.global_attn: True selects wavedm_amd.DiffusionUNet_Global (unet.py:397-636), not DiffusionUNet")
        self.config = config
        # optional input / output re-arrangements around the same network (unet.py:309-350, :387-391), off in raindrop_wavelet.yml
        self.use_window = bool(getattr(d, "use_window", False))
        self.window_size = int(getattr(d, "window_size", 2))
        self.use_wavelet_in_unet = bool(getattr(d, "wavelet_in_unet", False))
        self.resolution = int(d.image_size)
        self.in_channels = unet_in_channels(config)
        self.out_ch = int(config.model.out_ch)
        self.ch = int(config.model.ch)
        self.temb_ch = self.ch * 4
        self._dtype_fallback = None                          # set by DenoisingDiffusion_Wavelet's automatic f16 choice: the mode to fall back to, loudly
        self._u = None
        self._set_dtype(resolve_dtype(config, dtype))
        L, u = _lib.lib(), self._u
        if self.use_wavelet_in_unet:                         # registered first, like unet.py:204-206 (frozen, state_dict keys only)
            from .wavelet import WaveletTransform
            self.wavelet_dec = WaveletTransform(scale=2, dec=True)
            self.wavelet_rec = WaveletTransform(scale=2, dec=False)
        self._names = []
        name, ndim, shape = C.c_char_p(), C.c_int(), (C.c_int64 * 4)()
        table = []
        for i in range(L.wdm_unet_num_params(u)):
            _lib.check(L.wdm_unet_param_info(u, i, C.byref(name), C.byref(ndim), C.byref(shape)))
            table.append((name.value.decode(), tuple(int(shape[k]) for k in range(ndim.value))))
        ordered = reference_param_order(table)
        for key, shp in ordered:
            self._names.append(key)
            self._register(key, shp)
        self._packed = None
        self._packed_sig = None
        self._ws, self._ws_need, self._ws_gen = {}, {}, None

    def _set_dtype(self, code):
        """(Re)create the library object for a compute mode.  The parameter table does not depend on the mode; the packed buffer and the workspaces do."""
        L = _lib.lib()
        if getattr(self, "_u", None):
            L.wdm_unet_destroy(self._u)
            self._u = None
        self._dtype_code = code
        self._torch_dtype = {_lib.WDM_BF16: torch.bfloat16, _lib.WDM_F16: torch.float16}.get(code, torch.float32)
        self._cfg = _make_config(self.config, code)
        u = C.c_void_p()
        _lib.check(L.wdm_unet_create(None, C.byref(self._cfg), C.byref(u)))
        self._u = u
        self._packed = None
        self._packed_sig = None
        self._ws, self._ws_need, self._ws_gen = {}, {}, None

    @property
    def dtype_name(self):
        return {_lib.WDM_F32: "f32", _lib.WDM_BF16: "bf16", _lib.WDM_F32X3: "f32x3", _lib.WDM_F16: "f16"}[self._dtype_code]

    @property
    def module(self):
        """The reference wraps the UNet in DistributedDataParallel and callers unwrap with `.module`
        (ddm_wavelet.py:286); inference needs no DDP, so `.module` is the model itself."""
        return self

    # ---- parameter tree ------------------------------------------------------------------------
    def _register(self, key, shape):
        parts = key.split(".")
        node = self
        for comp in parts[:-1]:
            if comp not in node._modules:
                node.add_module(comp, _Node())
            node = node._modules[comp]
        p = torch.empty(shape, dtype=torch.float32)
        leaf = parts[-1]
        # from-scratch initialisation = torch's reset_parameters of nn.Conv2d / nn.Linear (what the reference's modules get) and GroupNorm
        if len(shape) > 1:                                      # conv / linear weight
            nn.init.kaiming_uniform_(p, a=math.sqrt(5))
            fan_in = 1
            for v in shape[1:]:
                fan_in *= v
            self._last_fan_in = fan_in
        elif "norm" in parts[-2]:                               # GroupNorm: gamma 1, beta 0
            p.fill_(1.0) if leaf == "weight" else p.zero_()
        elif leaf == "bias":                                    # uniform(+-1/sqrt(fan_in)) of the weight registered just before it
            bound = 1.0 / math.sqrt(getattr(self, "_last_fan_in", 1))
            nn.init.uniform_(p, -bound, bound)
        else:
            p.fill_(1.0)
        node.register_parameter(leaf, nn.Parameter(p))

    def __del__(self):
        try:
            if getattr(self, "_u", None):
                _lib.lib().wdm_unet_destroy(self._u)
                self._u = None
        except Exception:
            pass

    # ---- weights -> packed device buffer ---------------------------------------------------------
    def packed_bytes(self):
        return int(_lib.lib().wdm_unet_packed_bytes(self._u))

    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def pack_weights(self, force=False):
        """(Re)pack the fp32 parameters into the single device buffer the kernels read."""
        sig = self._signature()
        if not force and self._packed is not None and sig == self._packed_sig:
            return self._packed
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("DiffusionUNet: move the module to the GPU first (.to('cuda')); there is no CPU path")
        L = _lib.lib()
        with torch.cuda.device(dev):
            if self._packed is None or self._packed.device != dev:
                self._packed = torch.empty(self.packed_bytes() + 256, dtype=torch.uint8, device=dev)
                _lib.check(L.wdm_unet_set_packed(self._u, _lib.ptr(self._packed), self._packed.numel()))
            sd = dict(self.named_parameters())
            try:
                for key in self._names:
                    src = sd[key].detach().contiguous()
                    _lib.check(L.wdm_unet_load_param(self._u, key.encode(), _lib.ptr(src), src.numel(), _lib.stream_ptr()))
            except RuntimeError as e:
                if self._dtype_fallback is None or self._dtype_code != _lib.WDM_F16 or "fp16 range" not in str(e):
                    raise
                # the automatic f16 choice met a checkpoint fp16 cannot hold: say so and carry on in the fallback mode (an EXPLICIT dtype='f16' raises instead)
                import warnings
                fb, self._dtype_fallback = self._dtype_fallback, None
                warnings.warn(f"wavedm_amd: THIS CHECKPOINT DOES NOT FIT THE DEFAULT f16 MODE ({e}); falling back to dtype={fb!r} "
                              f"(16-bit throughput mode, ~3e-3 of the fp32 result instead of <= 1e-3 -- pass dtype='f32x3' for parity at a third of the speed)",
                              RuntimeWarning, stacklevel=2)
                self._set_dtype(_lib.DTYPES[fb])
                return self.pack_weights(force=True)
        self._packed_sig = sig
        return self._packed

    def adopt_packed(self):
        """Declare the packed buffer valid after its bytes were filled externally (RCCL broadcast)."""
        _lib.check(_lib.lib().wdm_unet_mark_loaded(self._u))
        self._packed_sig = self._signature()

    def alloc_packed(self, device):
        L = _lib.lib()
        with torch.cuda.device(device):
            self._packed = torch.empty(self.packed_bytes() + 256, dtype=torch.uint8, device=device)
            _lib.check(L.wdm_unet_set_packed(self._u, _lib.ptr(self._packed), self._packed.numel()))
        return self._packed

    def workspace(self, B, device, slot=0):
        """Workspace of a forward call at batch B.  slot: calls that may be in flight at the same time (one per HIP stream: sampling.ddim_sample) need one each."""
        # ONE buffer per (device, slot), sized for the largest batch the slot has run and reused for smaller ones (the library takes any buffer of at
        # least wdm_unet_workspace_bytes(B)): chunks of different sizes on different streams never evict each other's buffers while kernels read them.
        # Everything is dropped when the device changes or the experiment switches were re-read (they change the arena's size: _lib.env_refresh()).
        gen = (str(device), _lib.env_generation())
        if self._ws_gen != gen:
            self._ws, self._ws_need, self._ws_gen = {}, {}, gen
        need = self._ws_need.get(B)
        if need is None:
            need = int(_lib.lib().wdm_unet_workspace_bytes(self._u, B))
            if need == 0:
                raise RuntimeError("wdm_unet_workspace_bytes failed: " + _lib.lib().wdm_last_error().decode())
            self._ws_need[B] = need
        buf = self._ws.get(slot)
        if buf is None or buf.numel() < need + 256:
            self._ws[slot] = buf = torch.empty(need + 256, dtype=torch.uint8, device=device)
        return buf

    # ---- forward -------------------------------------------------------------------------------
    def temb_table(self, t, B=1):
        """Rows of the timestep-dependent part of the network (embedding MLP + every temb_proj) for all timesteps t (n,) fp32 device at once:
        (n, temb_rows) fp32.  A sampler builds it once per run and feeds forward_nhwc one row per step (include/wavedm.h: wdm_unet_temb_table)."""
        assert t.dtype == torch.float32 and t.is_cuda and t.dim() == 1
        self.pack_weights()
        L = _lib.lib()
        with torch.cuda.device(t.device):
            out = torch.empty(t.numel(), int(L.wdm_unet_temb_rows(self._u)), dtype=torch.float32, device=t.device)
            ws = self.workspace(B, t.device)
            _lib.check(L.wdm_unet_temb_table(self._u, _lib.ptr(t), int(t.numel()), _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        return out

    def forward_nhwc(self, x96, t, eps_out, temb_row=None, ws_slot=0):
        """x96: (B,R,R,Cin) NHWC in the compute dtype; t: (n,) fp32 device; eps_out: (B,out_ch,R,R) fp32.
        temb_row: one row of temb_table() for this call's timestep (then t is not read).  ws_slot: see workspace()."""
        B = x96.shape[0]
        assert x96.is_contiguous() and x96.dtype == self._torch_dtype and tuple(x96.shape[1:]) == (self.resolution, self.resolution, self.in_channels)
        assert eps_out.is_contiguous() and eps_out.dtype == torch.float32 and tuple(eps_out.shape) == (B, self.out_ch, self.resolution, self.resolution)
        self.pack_weights()
        with torch.cuda.device(x96.device):
            ws = self.workspace(B, x96.device, ws_slot)
            if temb_row is not None:
                assert temb_row.dtype == torch.float32 and temb_row.is_cuda and temb_row.is_contiguous() and temb_row.dim() == 1
                _lib.check(_lib.lib().wdm_unet_forward_temb(self._u, _lib.ptr(x96), _lib.ptr(temb_row), B, _lib.ptr(eps_out),
                                                            _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
                return eps_out
            assert t.dtype == torch.float32 and t.is_cuda and t.numel() in (1, B)
            _lib.check(_lib.lib().wdm_unet_forward(self._u, _lib.ptr(x96), _lib.ptr(t), int(t.numel()), B, _lib.ptr(eps_out),
                                                   _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        return eps_out

    # ---- unet.py:309-344 ---------------------------------------------------------------------------
    @staticmethod
    def to_win(x, p):
        B, Cc, H, W = x.shape
        return x.view(B, Cc, p, H // p, p, W // p).permute(0, 1, 2, 4, 3, 5).contiguous().view(B, -1, H // p, W // p)

    @staticmethod
    def win_back(x, p):
        B, Cc, H, W = x.shape
        return x.view(B, Cc // (p * p), p, p, H, W).permute(0, 1, 2, 4, 3, 5).contiguous().view(B, Cc // (p * p), H * p, W * p)

    def forward(self, x, t):
        x = _lib.require_cuda_f32(x, "DiffusionUNet input")
        if self.use_window:                                  # convert_image_to_patches (:323-331)
            p = self.window_size
            x = torch.cat([self.to_win(x[:, :3].contiguous(), p), self.to_win(x[:, 3:].contiguous(), p)], dim=1)
        if self.use_wavelet_in_unet:                         # all_wavlet_dec (:338-344)
            x = torch.cat([self.wavelet_dec(x[:, :3].contiguous()), self.wavelet_dec(x[:, 3:].contiguous())], dim=1)
        h = self._core_forward(x.contiguous(), t)
        if self.use_window:
            h = self.win_back(h, self.window_size)           # convert_patches_to_image (:333-336)
        if self.use_wavelet_in_unet:
            h = self.wavelet_rec(h.contiguous())
        return h

    def _core_forward(self, x, t):
        B, Cc, H, W = x.shape
        assert H == W == self.resolution, "input resolution != config.data.image_size (unet.py:351)"
        assert Cc == self.in_channels and t.dim() == 1
        L, h = _lib.lib(), _lib.handle(x.device.index or 0)
        t = t.to(device=x.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(x.device):
            x96 = torch.empty(B, H, W, Cc, device=x.device, dtype=self._torch_dtype)
            _lib.check(L.wdm_nchw_to_nhwc(h, _lib.ptr(x), _lib.ptr(x96), B, Cc, H, W, self._dtype_code, _lib.stream_ptr()))
            eps = torch.empty(B, self.out_ch, H, W, device=x.device, dtype=torch.float32)
            return self.forward_nhwc(x96, t, eps)
