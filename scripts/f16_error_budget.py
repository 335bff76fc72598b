"""Where does the f16 mode's error come from?  CPU experiment on the oracle (test infrastructure): the full-width UNet forward with fp16 (or bf16)
roundings emulated at selectable sites --
  W  conv / attention-projection weights rounded once           (the packed matrices)
  S  every tensor a kernel stores (conv outputs after bias/temb/residual, attention q|k, v, P, O)
  A  the activation a conv stages for the MFMA: round(silu(gn(round_S(x))))   (the prologue's second rounding)
-- and the max-norm relative error of eps against the plain fp32 forward.  Usage: python scripts/f16_error_budget.py [f16|bf16] [R] [t]"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from oracle import wavedm_oracle as O
from wavedm_amd import procedural as P

DT = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = float(sys.argv[3]) if len(sys.argv) > 3 else 990.0
SITES = set()
ONLY = None      # per_block(): roundings only inside the block of this name


def rnd(x):
    return x.to(DT).to(torch.float32)


def on(name):
    return ONLY is None or name.startswith(ONLY + ".") or name == ONLY


def rel_linf(a, b):
    return float((a - b).abs().max() / b.abs().max())


def conv(sd, name, x, stride=1, padding=0):
    w = sd[name + ".weight"]
    if "W" in SITES and on(name):
        w = rnd(w)
    if "A" in SITES and on(name):
        x = rnd(x)
    y = F.conv2d(x, w, sd[name + ".bias"], stride=stride, padding=padding)
    return y


def resnet_block(sd, name, x, temb):
    S = (lambda v: rnd(v)) if ("S" in SITES and on(name)) else (lambda v: v)
    h = conv(sd, name + ".conv1", O.silu(O.group_norm(sd, name + ".norm1", x)), padding=1)
    h = S(h + O.linear(sd, name + ".temb_proj", O.silu(temb))[:, :, None, None])
    h = conv(sd, name + ".conv2", O.silu(O.group_norm(sd, name + ".norm2", h)), padding=1)
    if (name + ".nin_shortcut.weight") in sd:
        x = conv(sd, name + ".nin_shortcut", x)        # fused into conv2's accumulator in the 16-bit modes: no rounding of its own
    return S(x + h)


def attn_block(sd, name, x):
    S = (lambda v: rnd(v)) if ("S" in SITES and on(name)) else (lambda v: v)
    h = S(O.group_norm(sd, name + ".norm", x))
    q, k, v = S(conv(sd, name + ".q", h)), S(conv(sd, name + ".k", h)), S(conv(sd, name + ".v", h))
    b, c, hh, ww = q.shape
    n = hh * ww
    q = q.reshape(b, c, n).permute(0, 2, 1)
    k = k.reshape(b, c, n)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = S(F.softmax(w_, dim=2))
    v = v.reshape(b, c, n)
    o = S(torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww))
    return S(x + conv(sd, name + ".proj_out", o))


def run(sites, sd, cfg, x, t):
    global SITES
    SITES = set(sites)
    keep = (O.conv, O.resnet_block, O.attn_block)
    O.conv, O.resnet_block, O.attn_block = conv, resnet_block, attn_block
    try:
        return O.unet_forward(sd, cfg, rnd(x) if ("S" in SITES and ONLY is None) else x, t)
    finally:
        O.conv, O.resnet_block, O.attn_block = keep


def per_block():
    """WSA roundings inside ONE block at a time: which blocks carry the error."""
    global ONLY
    cfg = P.raindrop_wavelet_config(image_size=R)
    sd = P.procedural_state_dict(cfg)
    rainy, x_T = P.synthetic_batch(1, patch_px=4 * R)
    xc = O.dwt_fwd(2 * rainy - 1)
    x96 = torch.cat([xc, x_T, xc[:, 3:]], dim=1)
    t = torch.tensor([T])
    want = O.unet_forward(sd, cfg, x96, t)
    blocks = sorted({k.rsplit(".", 2)[0] for k in sd if k.endswith("conv1.weight") or k.endswith("proj_out.weight")})
    blocks += ["conv_in", "conv_out"] + sorted({k.rsplit(".", 2)[0] for k in sd if "sample.conv.weight" in k})
    res = []
    for b in blocks:
        ONLY = b
        got = run("WSA", sd, cfg, x96, t)
        res.append((rel_linf(got, want), b))
        print(f"{b:28s} {res[-1][0]:.3e}", flush=True)
    ONLY = None
    print("top:", sorted(res, reverse=True)[:8])


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[4] == "blocks":
        per_block()
        sys.exit(0)
    torch.manual_seed(0)
    cfg = P.raindrop_wavelet_config(image_size=R)
    sd = P.procedural_state_dict(cfg)
    rainy, x_T = P.synthetic_batch(1, patch_px=4 * R)
    xc = O.dwt_fwd(2 * rainy - 1)
    x96 = torch.cat([xc, x_T, xc[:, 3:]], dim=1)
    t = torch.tensor([T])
    want = O.unet_forward(sd, cfg, x96, t)
    for sites in ("W", "S", "A", "WS", "WA", "SA", "WSA"):
        got = run(sites, sd, cfg, x96, t)
        print(f"{sys.argv[1] if len(sys.argv) > 1 else 'f16'} R={R} t={T:.0f} sites {sites:4s}: rel_linf {rel_linf(got, want):.3e}", flush=True)
