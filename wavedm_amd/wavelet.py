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
