"""HFRM -- drop-in for the reference's `models/arch.py:206-253` (the high-frequency refinement module that
`utils/restoration.py:94` runs once per image before the diffusion sampler) whose forward runs on
libwavedm_hip.so.

* same constructor signature and defaults as the reference class, same `state_dict()` keys / shapes / order
  (448 tensors, 15.94 M parameters at the `ddm_wavelet.py:139` configuration), so `lastest.pth` loads with
  `load_state_dict(strict=True)`;
* `forward(x)`: x (B, 3, H, W) fp32 NCHW on the GPU, H and W multiples of 16 -> (B, 3, H, W) fp32 NCHW.

The parameter tree comes from the library's own table (wdm_hfrm_param_info)."""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib
from .unet import _Node, resolve_dtype


class HFRM(nn.Module):
    # batch chunk: keeps every activation below the kernels' 4 GB buffer-descriptor range
    MAX_PIXELS = 1 << 22

    def __init__(self, in_channel=3, dim=32, mid_blk_num=1, enc_blk_nums=[], dec_blk_nums=[], win_size=8, dtype=None):
        super().__init__()
        if len(enc_blk_nums) != len(dec_blk_nums) or not 1 <= len(enc_blk_nums) <= 8:
            raise ValueError("HFRM: enc_blk_nums and dec_blk_nums must have the same length (1..8)")
        self.in_channel, self.dim = int(in_channel), int(dim)
        self.padder_size = 2 ** len(enc_blk_nums)
        self._dtype_code = resolve_dtype(None, dtype)
        if self._dtype_code in (_lib.WDM_F32X3, _lib.WDM_F16):      # the HFRM has two modes; the parity modes of the UNet (f32x3, f16) map to its exact one
            self._dtype_code = _lib.WDM_F32
        cfg = _lib.HFRMConfig()
        cfg.in_channel, cfg.dim, cfg.mid_blk_num = self.in_channel, self.dim, int(mid_blk_num)
        cfg.n_enc, cfg.n_dec = len(enc_blk_nums), len(dec_blk_nums)
        for i, v in enumerate(enc_blk_nums):
            cfg.enc_blk_nums[i] = int(v)
        for i, v in enumerate(dec_blk_nums):
            cfg.dec_blk_nums[i] = int(v)
        cfg.dtype = self._dtype_code
        self._cfg = cfg
        L = _lib.lib()
        m = C.c_void_p()
        _lib.check(L.wdm_hfrm_create(None, C.byref(cfg), C.byref(m)))
        self._m = m
        self._names = []
        name, ndim, shape = C.c_char_p(), C.c_int(), (C.c_int64 * 4)()
        for i in range(L.wdm_hfrm_num_params(m)):
            _lib.check(L.wdm_hfrm_param_info(m, i, C.byref(name), C.byref(ndim), C.byref(shape)))
            key = name.value.decode()
            self._names.append(key)
            self._register(key, tuple(int(shape[k]) for k in range(ndim.value)))
        self._packed = None
        self._packed_sig = None
        self._ws = {}

    def _register(self, key, shape):
        parts = key.split(".")
        node = self
        for comp in parts[:-1]:
            if comp not in node._modules:
                node.add_module(comp, _Node())
            node = node._modules[comp]
        p = torch.empty(shape, dtype=torch.float32)
        leaf = parts[-1]
        if leaf in ("beta", "gamma"):          # arch.py:165-166: zeros
            p.zero_()
        elif len(shape) > 1:
            nn.init.kaiming_uniform_(p, a=math.sqrt(5))
        elif leaf == "weight":
            p.fill_(1.0)
        else:
            p.zero_()
        node.register_parameter(leaf, nn.Parameter(p))

    def __del__(self):
        try:
            if getattr(self, "_m", None):
                _lib.lib().wdm_hfrm_destroy(self._m)
                self._m = None
        except Exception:
            pass

    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def pack_weights(self, force=False):
        sig = self._signature()
        if not force and self._packed is not None and sig == self._packed_sig:
            return self._packed
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("HFRM: move the module to the GPU first (.to('cuda')); there is no CPU path")
        L = _lib.lib()
        with torch.cuda.device(dev):
            if self._packed is None or self._packed.device != dev:
                self._packed = torch.empty(int(L.wdm_hfrm_packed_bytes(self._m)) + 256, dtype=torch.uint8, device=dev)
                _lib.check(L.wdm_hfrm_set_packed(self._m, _lib.ptr(self._packed), self._packed.numel()))
            sd = dict(self.named_parameters())
            for key in self._names:
                src = sd[key].detach().contiguous()
                _lib.check(L.wdm_hfrm_load_param(self._m, key.encode(), _lib.ptr(src), src.numel(), _lib.stream_ptr()))
            _lib.check(L.wdm_hfrm_finalize(self._m, _lib.stream_ptr()))
        self._packed_sig = sig
        return self._packed

    def _workspace(self, B, H, W, device):
        key = (B, H, W, str(device))
        if key not in self._ws:
            n = int(_lib.lib().wdm_hfrm_workspace_bytes(self._m, B, H, W))
            if n == 0:
                raise RuntimeError("wdm_hfrm_workspace_bytes failed: " + _lib.lib().wdm_last_error().decode())
            self._ws = {key: torch.empty(n + 256, dtype=torch.uint8, device=device)}
        return self._ws[key]

    def forward(self, x):
        x = _lib.require_cuda_f32(x, "HFRM input")
        B, Cc, H, W = x.shape
        if Cc != self.in_channel:
            raise ValueError(f"HFRM: {Cc} input channels, expected {self.in_channel}")
        self.pack_weights()
        y = torch.empty_like(x)
        chunk = max(1, self.MAX_PIXELS // (H * W))
        L = _lib.lib()
        with torch.cuda.device(x.device):
            for b0 in range(0, B, chunk):
                nb = min(chunk, B - b0)
                ws = self._workspace(nb, H, W, x.device)
                _lib.check(L.wdm_hfrm_forward(self._m, _lib.ptr(x[b0:b0 + nb]), nb, H, W, _lib.ptr(y[b0:b0 + nb]),
                                              _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        return y
