"""Helpers for the `-m gpu` parity tests: thin ctypes callers of the per-block C entry points."""
import ctypes as C

import torch

from wavedm_amd import _lib
from wavedm_amd import procedural as P

DT = {"f32": _lib.WDM_F32, "bf16": _lib.WDM_BF16, "f32x3": _lib.WDM_F32X3, "f16": _lib.WDM_F16}
# tolerances (max-norm relative): f32 is the parity mode of BASELINE.json's north_star (1e-3);
# bf16 is the throughput mode -- its deviation is bounded here (<= 2x the measured worst case) so regressions show, it is not a parity claim.
# f16 (fp16 operands, fp32 accumulation) is held to the parity bound itself.
TOL = {"f32": 1e-3, "f32x3": 1e-3, "f16": 1e-3, "bf16": 1e-2}


def dev():
    return torch.device("cuda", 0)


def seeded(shape, seed, kind="randn"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn if kind == "randn" else torch.rand)(*shape, generator=g, dtype=torch.float32)


def blk_sd(prefix, shapes):
    return {prefix + "." + k: torch.from_numpy(P.procedural_tensor(prefix + "." + k, s)) for k, s in shapes.items()}


def scratch(nbytes=1 << 30):
    return torch.empty(nbytes, dtype=torch.uint8, device=dev())


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def resblock(sd, name, x0, x1, temb, dtype):
    """sd: CPU state dict of one block (keys name.norm1.weight ...); x0/x1/temb CPU tensors -> CPU output."""
    L, h = _lib.lib(), _lib.handle(0)
    d = {k: v.to(dev()).contiguous() for k, v in sd.items()}
    g = lambda k: d.get(name + "." + k)
    cin = g("conv1.weight").shape[1]
    cout = g("conv1.weight").shape[0]
    p = _lib.ResblockParams()
    p.cin, p.cout = cin, cout
    for f, k in (("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"), ("conv1_w", "conv1.weight"), ("conv1_b", "conv1.bias"),
                 ("temb_w", "temb_proj.weight"), ("temb_b", "temb_proj.bias"), ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"),
                 ("conv2_w", "conv2.weight"), ("conv2_b", "conv2.bias"), ("nin_w", "nin_shortcut.weight"), ("nin_b", "nin_shortcut.bias")):
        t = g(k)
        setattr(p, f, t.data_ptr() if t is not None else None)
    x0d = x0.to(dev()).contiguous()
    x1d = x1.to(dev()).contiguous() if x1 is not None else None
    td = temb.to(dev()).contiguous()
    B, c0, H, W = x0d.shape
    c1 = x1d.shape[1] if x1d is not None else 0
    y = torch.empty(B, cout, H, W, device=dev())
    sc = scratch()
    _lib.check(L.wdm_resblock_forward(h, C.byref(p), _p(x0d), c0, _p(x1d), c1, _p(td), td.shape[0], td.shape[1], B, H, W, _p(y),
                                      DT[dtype], _p(sc), sc.numel(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    return y.cpu()


def attn(sd, name, x, dtype):
    L, h = _lib.lib(), _lib.handle(0)
    d = {k: v.to(dev()).contiguous() for k, v in sd.items()}
    g = lambda k: d[name + "." + k]
    p = _lib.AttnParams()
    p.c = x.shape[1]
    for f, k in (("norm_w", "norm.weight"), ("norm_b", "norm.bias"), ("q_w", "q.weight"), ("q_b", "q.bias"), ("k_w", "k.weight"),
                 ("k_b", "k.bias"), ("v_w", "v.weight"), ("v_b", "v.bias"), ("proj_w", "proj_out.weight"), ("proj_b", "proj_out.bias")):
        setattr(p, f, g(k).data_ptr())
    xd = x.to(dev()).contiguous()
    B, Cc, H, W = xd.shape
    y = torch.empty_like(xd)
    sc = scratch()
    _lib.check(L.wdm_attn_forward(h, C.byref(p), _p(xd), B, H, W, _p(y), DT[dtype], _p(sc), sc.numel(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    return y.cpu()


def conv(w, b, mode, x, dtype):
    """mode: 0 conv3x3 s1 p1, 1 Downsample, 2 Upsample, 3 conv1x1."""
    L, h = _lib.lib(), _lib.handle(0)
    wd, bd, xd = w.to(dev()).contiguous(), b.to(dev()).contiguous(), x.to(dev()).contiguous()
    B, cin, H, W = xd.shape
    cout = wd.shape[0]
    Ho, Wo = (H // 2, W // 2) if mode == 1 else (H * 2, W * 2) if mode == 2 else (H, W)
    y = torch.empty(B, cout, Ho, Wo, device=dev())
    sc = scratch()
    _lib.check(L.wdm_conv_forward(h, _p(wd), _p(bd), cin, cout, mode, _p(xd), B, H, W, _p(y), DT[dtype], _p(sc), sc.numel(),
                                  _lib.stream_ptr()))
    torch.cuda.synchronize()
    return y.cpu()


def temb(t, ch, w0, b0, w1, b1):
    L, h = _lib.lib(), _lib.handle(0)
    td = t.to(dev()).float().contiguous()
    ws = [v.to(dev()).contiguous() for v in (w0, b0, w1, b1)]
    out = torch.empty(td.numel(), ch * 4, device=dev())
    sc = scratch(1 << 24)
    _lib.check(L.wdm_temb_forward(h, _p(td), td.numel(), ch, _p(ws[0]), _p(ws[1]), _p(ws[2]), _p(ws[3]), _p(out), _p(sc), sc.numel(),
                                  _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out.cpu()
