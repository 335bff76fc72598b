"""WaveletTransform -- drop-in for the reference's `models/wavelet.py:6-50` on the HIP path.

Same constructor signature and call semantics: `WaveletTransform(scale=2, dec=True)(x)` maps
(B,3,H,W) -> (B,48,H/4,W/4) with channel = sub_band*3 + rgb; `dec=False` is the inverse.  The
reference builds a frozen grouped (de)conv from a pickle; here the +-0.25 Walsh/Haar basis is
computed in closed form inside the kernel (`csrc/elementwise.hip: dwt_fwd_kernel`), so no pickle
is needed (`params_path` is accepted and ignored).  The module still carries the frozen `conv.weight` (48,1,4,4) the
reference registers (wavelet.py:26-33), so a state_dict that contains it -- a DiffusionUNet saved with
data.wavelet_in_unet -- loads with strict=True; the kernels do not read it."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib


def haar_packet_weight() -> torch.Tensor:
    """(48,1,4,4): row c*16+j = sub-band j's 4x4 filter, entries +-1/4 with sign (-1)^(j0*(q>>1) + j1*(p>>1) + j2*(q&1) + j3*(p&1))
    for j = j3 j2 j1 j0 -- the values the reference un-pickles as dct['rec4'] (wavelet.py:22-27)."""
    j = torch.arange(16).view(16, 1, 1)
    p = torch.arange(4).view(1, 4, 1)
    q = torch.arange(4).view(1, 1, 4)
    e = (j & 1) * (q >> 1) + ((j >> 1) & 1) * (p >> 1) + ((j >> 2) & 1) * (q & 1) + ((j >> 3) & 1) * (p & 1)
    f = 0.25 - 0.5 * (e & 1).float()
    return f.repeat(3, 1, 1).unsqueeze(1).contiguous()


class _FrozenConv(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(haar_packet_weight(), requires_grad=False)


class WaveletTransform(nn.Module):
    def __init__(self, scale=1, dec=True, params_path="./models/wavelet_weights_c2.pkl", transpose=True):
        super().__init__()
        if scale != 2 or not transpose:
            raise NotImplementedError("wavedm_amd.WaveletTransform implements scale=2, transpose=True "
                                      "(the only configuration the reference constructs, ddm_wavelet.py:134-135)")
        self.scale, self.dec, self.transpose = scale, dec, transpose
        self.conv = _FrozenConv()

    def forward_affine(self, x, scale=2.0, shift=-1.0):
        """dec only: DWT(scale * x + shift) in one kernel -- `wavelet_dec(data_transform(x))` of the reference (restoration.py:88-96)."""
        assert self.dec
        x = _lib.require_cuda_f32(x, "WaveletTransform input")
        B, C, H, W = x.shape
        if C != 3 or H % 4 or W % 4:
            raise ValueError(f"WaveletTransform(dec): expected (B,3,4h,4w), got {tuple(x.shape)}")
        y = torch.empty(B, 48, H // 4, W // 4, device=x.device, dtype=torch.float32)
        if B:
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().wdm_dwt_fwd_affine(_lib.handle(x.device.index or 0), _lib.ptr(x), float(scale), float(shift), _lib.ptr(y), B, H, W,
                                                         _lib.stream_ptr()))
        return y

    def compose(self, lo, hi, n_lo, to_unit_range=True):
        """rec only: IDWT(cat([lo[:, :n_lo], hi[:, n_lo:]])) [then clamp((x + 1) / 2, 0, 1)] in one kernel -- restoration.py:114-134."""
        assert not self.dec
        hi = _lib.require_cuda_f32(hi, "WaveletTransform hi")
        lo = _lib.require_cuda_f32(lo, "WaveletTransform lo")
        B, C, hh, ww = hi.shape
        if C != 48 or lo.shape[0] != B or tuple(lo.shape[2:]) != (hh, ww) or not (0 <= n_lo <= min(48, lo.shape[1])):
            raise ValueError(f"WaveletTransform.compose: lo {tuple(lo.shape)} / hi {tuple(hi.shape)} / n_lo {n_lo}")
        y = torch.empty(B, 3, hh * 4, ww * 4, device=hi.device, dtype=torch.float32)
        if B:
            with torch.cuda.device(hi.device):
                _lib.check(_lib.lib().wdm_dwt_inv_compose(_lib.handle(hi.device.index or 0), _lib.ptr(lo), lo.shape[1], int(n_lo), _lib.ptr(hi), _lib.ptr(y),
                                                          B, hh, ww, 1 if to_unit_range else 0, _lib.stream_ptr()))
        return y

    def forward(self, x):
        x = _lib.require_cuda_f32(x, "WaveletTransform input")
        L, h = _lib.lib(), _lib.handle(x.device.index or 0)
        with torch.cuda.device(x.device):
            if self.dec:
                B, C, H, W = x.shape
                if C != 3 or H % 4 or W % 4:
                    raise ValueError(f"WaveletTransform(dec): expected (B,3,4h,4w), got {tuple(x.shape)}")
                y = torch.empty(B, 48, H // 4, W // 4, device=x.device, dtype=torch.float32)
                if B:
                    _lib.check(L.wdm_dwt_fwd(h, _lib.ptr(x), _lib.ptr(y), B, H, W, _lib.stream_ptr()))
                return y
            B, C, hh, ww = x.shape
            if C != 48:
                raise ValueError(f"WaveletTransform(rec): expected (B,48,h,w), got {tuple(x.shape)}")
            y = torch.empty(B, 3, hh * 4, ww * 4, device=x.device, dtype=torch.float32)
            if B:
                _lib.check(L.wdm_dwt_inv(h, _lib.ptr(x), _lib.ptr(y), B, hh, ww, _lib.stream_ptr()))
            return y
