"""CPU ORACLE for the WaveDM sampling hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.
The product path (`wavedm_amd/`) never does: it fails loudly when the HIP library is missing.

What it is: a from-scratch, functional, fp32 torch-CPU restatement of the reference's algorithm
for SURVEY.md §8(a) rows a1-a17.  It works on a flat `state_dict` (name -> tensor) and plain
tensors, has no nn.Module tree, and shares no code with the reference.  Each function cites the
reference lines it restates.

Pinning: the reference has no tests / golden vectors of its own, so this oracle is
pinned against the reference itself, imported (with stub modules) in the build container by
`tests/golden/make_golden.py`; that script asserts oracle == reference on every fixture it
writes and the fixtures are committed under `tests/golden/`.  `tests/test_oracle_golden.py`
re-checks the oracle against those fixtures on any box (no reference needed).

All layouts here are the reference's: NCHW fp32.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# a1-a3  Haar wavelet-packet DWT / IDWT      (reference: models/wavelet.py:7-49)
# ----------------------------------------------------------------------------------------------
def haar_filters() -> torch.Tensor:
    """(16,4,4) fp32 analysis filters f_j[p][q], j = sub-band index.

    The reference loads them from `wavelet_weights_c2.pkl['rec4']` (wavelet.py:26-33): a 16x16
    orthonormal Walsh/Haar basis with entries +-0.25.  Closed form (checked against the pickle's
    sign matrix by tests/golden/make_golden.py):
        f_j[p][q] = 0.25 * (-1)^( j0*(q>>1) + j1*(p>>1) + j2*(q&1) + j3*(p&1) ),  j = j3 j2 j1 j0.
    """
    f = torch.empty(16, 4, 4, dtype=torch.float32)
    for j in range(16):
        j0, j1, j2, j3 = j & 1, (j >> 1) & 1, (j >> 2) & 1, (j >> 3) & 1
        for p in range(4):
            for q in range(4):
                e = j0 * (q >> 1) + j1 * (p >> 1) + j2 * (q & 1) + j3 * (p & 1)
                f[j, p, q] = -0.25 if (e & 1) else 0.25
    return f


def dwt_fwd(x: torch.Tensor) -> torch.Tensor:
    """(B,3,H,W) -> (B,48,H/4,W/4); out channel = j*3 + c  (wavelet.py:37-43).

    Grouped stride-4 conv gives channel c*16+j; the reference then permutes to sub-band-major
    j*3+c via view(B,3,16,h,w).transpose(1,2)."""
    B, C, H, W = x.shape
    assert C == 3 and H % 4 == 0 and W % 4 == 0
    w = haar_filters().repeat(3, 1, 1).unsqueeze(1)            # (48,1,4,4): row c*16+j = f_j
    y = F.conv2d(x, w, stride=4, groups=3)                     # (B, c*16+j, h, w)
    h, wd = H // 4, W // 4
    return y.view(B, 3, 16, h, wd).transpose(1, 2).reshape(B, 48, h, wd)


def dwt_inv(y: torch.Tensor) -> torch.Tensor:
    """(B,48,h,w) with channel j*3+c -> (B,3,4h,4w)  (wavelet.py:44-49)."""
    B, C, h, wd = y.shape
    assert C == 48
    w = haar_filters().repeat(3, 1, 1).unsqueeze(1)
    yy = y.view(B, 16, 3, h, wd).transpose(1, 2).reshape(B, 48, h, wd)   # back to c*16+j
    return F.conv_transpose2d(yy, w, stride=4, groups=3)


def data_transform(x):             # restoration.py:8-9
    return 2 * x - 1.0


def inverse_data_transform(x):     # restoration.py:12-13
    return torch.clamp((x + 1.0) / 2.0, 0.0, 1.0)


# ----------------------------------------------------------------------------------------------
# a4-a12  UNet                                (reference: models/unet.py:10-395)
# ----------------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """unet.py:10-28: [sin(t*w_i), cos(t*w_i)], w_i = exp(-i*ln(1e4)/(dim/2-1))."""
    assert t.dim() == 1
    half = dim // 2
    w = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    e = t.float()[:, None] * w[None, :]
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1)
    if dim % 2 == 1:
        e = F.pad(e, (0, 1, 0, 0))
    return e


def silu(x):                       # unet.py:31-33
    return x * torch.sigmoid(x)


def group_norm(sd, name, x):       # unet.py:36-37: 32 groups, eps 1e-6, affine
    return F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], eps=1e-6)


def conv(sd, name, x, stride=1, padding=0):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def resnet_block(sd, name, x, temb):
    """unet.py:119-138."""
    h = conv(sd, name + ".conv1", silu(group_norm(sd, name + ".norm1", x)), padding=1)
    h = h + linear(sd, name + ".temb_proj", silu(temb))[:, :, None, None]
    h = conv(sd, name + ".conv2", silu(group_norm(sd, name + ".norm2", h)), padding=1)   # dropout p=0
    if (name + ".nin_shortcut.weight") in sd:
        x = conv(sd, name + ".nin_shortcut", x)
    return x + h


def attn_block(sd, name, x):
    """unet.py:168-193: single head, scale C^-0.5, softmax over keys."""
    h = group_norm(sd, name + ".norm", x)
    q, k, v = conv(sd, name + ".q", h), conv(sd, name + ".k", h), conv(sd, name + ".v", h)
    b, c, hh, ww = q.shape
    n = hh * ww
    q = q.reshape(b, c, n).permute(0, 2, 1)          # b, n, c
    k = k.reshape(b, c, n)                           # b, c, n
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))        # b, n(query), n(key)
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, n)
    o = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + conv(sd, name + ".proj_out", o)


def downsample(sd, name, x):
    """unet.py:71-78: zero-pad right/bottom by one, conv3x3 stride 2 pad 0."""
    return conv(sd, name + ".conv", F.pad(x, (0, 1, 0, 1)), stride=2)


def upsample(sd, name, x):
    """unet.py:51-56: nearest x2, conv3x3 pad 1."""
    return conv(sd, name + ".conv", F.interpolate(x, scale_factor=2.0, mode="nearest"), padding=1)


def to_win(x, p):                   # unet.py:309-314
    B, C, H, W = x.shape
    return x.view(B, C, p, H // p, p, W // p).permute(0, 1, 2, 4, 3, 5).contiguous().view(B, -1, H // p, W // p)


def win_back(x, p):                 # unet.py:316-321
    B, C, H, W = x.shape
    return x.view(B, C // (p ** 2), p, p, H, W).permute(0, 1, 2, 4, 3, 5).contiguous().view(B, C // (p ** 2), H * p, W * p)


def unet_forward(sd, config, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """unet.py:346-395, including the optional use_window / wavelet_in_unet re-arrangements (:347-350, :387-391; both off in
    raindrop_wavelet.yml).

    Default layout: x (B, 96, R, R) = [x_cond 0:48 | x_t 48:51 | x_other 51:96]; t: (n,) float, n in {1, B}.
    """
    d = config.data
    if getattr(d, "use_window", False):
        p = d.window_size
        x = torch.cat([to_win(x[:, :3], p), to_win(x[:, 3:], p)], dim=1)
    if getattr(d, "wavelet_in_unet", False):
        x = torch.cat([dwt_fwd(x[:, :3]), dwt_fwd(x[:, 3:])], dim=1)
    h = _unet_core(sd, config, x, t)
    if getattr(d, "use_window", False):
        h = win_back(h, d.window_size)
    if getattr(d, "wavelet_in_unet", False):
        h = dwt_inv(h)
    return h


def _unet_core(sd, config, x, t):
    m = config.model
    ch, ch_mult = m.ch, tuple(m.ch_mult)
    nres, nrb, attn_res = len(ch_mult), m.num_res_blocks, list(m.attn_resolutions)
    assert x.shape[2] == x.shape[3] == config.data.image_size

    temb = timestep_embedding(t, ch)
    temb = linear(sd, "temb.dense.1", silu(linear(sd, "temb.dense.0", temb)))

    res = config.data.image_size
    hs = [conv(sd, "conv_in", x, padding=1)]
    for l in range(nres):
        for b in range(nrb):
            h = resnet_block(sd, f"down.{l}.block.{b}", hs[-1], temb)
            if res in attn_res:
                h = attn_block(sd, f"down.{l}.attn.{b}", h)
            hs.append(h)
        if l != nres - 1:
            hs.append(downsample(sd, f"down.{l}.downsample", hs[-1]))
            res //= 2

    h = hs[-1]
    h = resnet_block(sd, "mid.block_1", h, temb)
    h = attn_block(sd, "mid.attn_1", h)
    h = resnet_block(sd, "mid.block_2", h, temb)

    for l in reversed(range(nres)):
        for b in range(nrb + 1):
            h = resnet_block(sd, f"up.{l}.block.{b}", torch.cat([h, hs.pop()], dim=1), temb)
            if res in attn_res:
                h = attn_block(sd, f"up.{l}.attn.{b}", h)
        if l != 0:
            h = upsample(sd, f"up.{l}.upsample", h)
            res *= 2

    return conv(sd, "conv_out", silu(group_norm(sd, "norm_out", h)), padding=1)


def attn_global(sd, name, x_patch, x_global, local_patch=2):
    """Attn_Global.forward, unet.py:432-462.  NOTE the reference normalises BOTH inputs with `norm_patch` (:433-434; `norm_global` is
    registered but unused) and its k / v are depthwise 8x8 stride-8 convolutions of the whole-image map."""
    h_ = group_norm(sd, name + ".norm_patch", x_patch)
    hg = group_norm(sd, name + ".norm_patch", x_global)
    c = x_patch.shape[1]
    q = F.conv2d(h_, sd[name + ".q.weight"], sd[name + ".q.bias"], stride=local_patch)
    gp = sd[name + ".k.weight"].shape[-1]
    k = F.conv2d(hg, sd[name + ".k.weight"], sd[name + ".k.bias"], stride=gp, groups=c)
    v = F.conv2d(hg, sd[name + ".v.weight"], sd[name + ".v.bias"], stride=gp, groups=c)
    b, _, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, -1)
    w_ = F.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)
    v = v.reshape(b, c, -1)
    o = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    o = conv(sd, name + ".proj_out", o)
    if local_patch > 1:
        o = F.interpolate(o, scale_factor=float(local_patch), mode="nearest")
    return x_patch + o


def unet_global_forward(sd, config, x, t, x_global):
    """DiffusionUNet_Global.forward, unet.py:585-636 -- including its quirk that the middle starts from `hs[-1]` (:612), i.e. the output
    of the LAST level's global attention is discarded while the earlier levels' feed the next level."""
    m = config.model
    ch, ch_mult = m.ch, tuple(m.ch_mult)
    nres, nrb, attn_res = len(ch_mult), m.num_res_blocks, list(m.attn_resolutions)
    temb = linear(sd, "temb.dense.1", silu(linear(sd, "temb.dense.0", timestep_embedding(t, ch))))
    res = config.data.image_size
    hg = conv(sd, "global_conv_in", x_global, padding=1)
    hs = [conv(sd, "conv_in", x, padding=1)]
    h = hs[-1]
    for l in range(nres):
        for b in range(nrb):
            h = resnet_block(sd, f"down.{l}.block.{b}", h, temb)
            if res in attn_res:
                h = attn_block(sd, f"down.{l}.attn.{b}", h)
            hs.append(h)
        if l != nres - 1:
            h = downsample(sd, f"down.{l}.downsample", h)
            hs.append(h)
            res //= 2
            hg = F.conv2d(hg, sd[f"down_global.{l}.conv.weight"], sd[f"down_global.{l}.conv.bias"], stride=2, padding=1)
        h = attn_global(sd, f"down_global.{l}.attn", h, hg)
    h = hs[-1]
    h = resnet_block(sd, "mid.block_1", h, temb)
    h = attn_block(sd, "mid.attn_1", h)
    h = resnet_block(sd, "mid.block_2", h, temb)
    for l in reversed(range(nres)):
        for b in range(nrb + 1):
            h = resnet_block(sd, f"up.{l}.block.{b}", torch.cat([h, hs.pop()], dim=1), temb)
            if res in attn_res:
                h = attn_block(sd, f"up.{l}.attn.{b}", h)
        if l != 0:
            h = upsample(sd, f"up.{l}.upsample", h)
            res *= 2
            hg = F.conv_transpose2d(hg, sd[f"up_global.{l}.conv.weight"], sd[f"up_global.{l}.conv.bias"], stride=2, padding=1)
        h = attn_global(sd, f"up_global.{l}.attn", h, hg)
    return conv(sd, "conv_out", silu(group_norm(sd, "norm_out", h)), padding=1)


# ----------------------------------------------------------------------------------------------
# a13-a16  schedule, grid, DDIM sampler       (reference: models/ddm_wavelet.py, utils/sampling.py)
# ----------------------------------------------------------------------------------------------
def beta_schedule(config) -> torch.Tensor:
    """ddm_wavelet.py:87-105 (linear only is used) -> fp32 as at :177."""
    d = config.diffusion
    assert d.beta_schedule == "linear"
    b = np.linspace(d.beta_start, d.beta_end, d.num_diffusion_timesteps, dtype=np.float64)
    return torch.from_numpy(b).float()


def compute_alpha(betas: torch.Tensor, t: int) -> torch.Tensor:
    """utils/sampling.py:10-13: abar(t) = cumprod(1 - [0, beta])[t+1] in fp32; abar(-1) = 1."""
    b = torch.cat([torch.zeros(1), betas], dim=0)
    return (1 - b).cumprod(dim=0)[t + 1]


def timestep_seq(num_timesteps: int, sampling_timesteps: int):
    """ddm_wavelet.py:296-297."""
    skip = num_timesteps // sampling_timesteps
    return list(range(0, num_timesteps, skip))


def overlapping_grid_indices(h: int, w: int, p: int, r: int = 16):
    """ddm_wavelet.py:426-435 (dup restoration.py:187-196)."""
    r = 16 if r is None else r
    h_list = list(range(0, h - p + 1, r))
    w_list = list(range(0, w - p + 1, r))
    if h_list[-1] + p < h:
        h_list.append(h - p)
    if w_list[-1] + p < w:
        w_list.append(w - p)
    return h_list, w_list


def grid_corners(h, w, p, r=16):
    hl, wl = overlapping_grid_indices(h, w, p, r)
    return [(i, j) for i in hl for j in wl]          # ddm_wavelet.py:419


def overlap_count_mask(h, w, p, corners) -> torch.Tensor:
    """ddm_wavelet.py:451-453 (integer-valued)."""
    m = torch.zeros(h, w, dtype=torch.int32)
    for (hi, wi) in corners:
        m[hi:hi + p, wi:wi + p] += 1
    return m


def ddim_overlapping(sd, config, x, x_cond, x_other, corners, p, sampling_timesteps,
                     betas=None, model=None, chunk=8, eta=0.0, noises=None):
    """ddm_wavelet.py:437-506, begin_from_noise=True; use_other=False <=> x_other is None (:471-478).
    eta != 0 (:500-502): `noises` is the list of the per-step `torch.randn_like(x)` draws (None: drawn here from torch's default generator, like the reference).

    x: (1,3,H,W) start noise, x_cond: (1,48,H,W), x_other: (1,45,H,W) or None.
    Returns (xs, x0_preds) lists like the reference (len S+1 and S).
    `model(x96, t)` defaults to this file's `unet_forward`.
    The reference draws `randn_like` every step and multiplies it by c1 = 0: no effect on values.
    """
    if betas is None:
        betas = beta_schedule(config)
    if model is None:
        model = lambda x96, t: unet_forward(sd, config, x96, t)
    n = x.size(0)
    seq = timestep_seq(config.diffusion.num_diffusion_timesteps, sampling_timesteps)
    seq_next = [-1] + list(seq[:-1])
    xs, x0_preds = [x], []
    mask = torch.zeros_like(x)
    for (hi, wi) in corners:
        mask[:, :, hi:hi + p, wi:wi + p] += 1
    with torch.no_grad():
        for i_t, j_t in zip(reversed(seq), reversed(seq_next)):
            t = torch.ones(n) * i_t
            at, at_next = compute_alpha(betas, i_t), compute_alpha(betas, j_t)
            xt = xs[-1]
            acc = torch.zeros_like(x)
            xt_p = torch.cat([xt[:, :, hi:hi + p, wi:wi + p] for (hi, wi) in corners], dim=0)
            xc_p = torch.cat([x_cond[:, :, hi:hi + p, wi:wi + p] for (hi, wi) in corners], dim=0)
            if x_other is not None:
                xo_p = torch.cat([x_other[:, :, hi:hi + p, wi:wi + p] for (hi, wi) in corners], dim=0)
            for i in range(0, len(corners), chunk):
                x96 = torch.cat([xc_p[i:i + chunk], xt_p[i:i + chunk]] + ([xo_p[i:i + chunk]] if x_other is not None else []), dim=1)
                out = model(x96, t)
                for idx, (hi, wi) in enumerate(corners[i:i + chunk]):
                    acc[0, :, hi:hi + p, wi:wi + p] += out[idx]
            et = acc / mask
            x0_t = (xt - et * (1 - at).sqrt()) / at.sqrt()
            x0_preds.append(x0_t)
            if eta == 0.0:
                c2 = (1 - at_next).sqrt()                  # c1 = 0 for eta = 0
                xs.append(at_next.sqrt() * x0_t + c2 * et)
            else:                                          # ddm_wavelet.py:500-502
                c1 = eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
                c2 = ((1 - at_next) - c1 ** 2).sqrt()
                z = torch.randn_like(x) if noises is None else noises[len(x0_preds) - 1]
                xs.append(at_next.sqrt() * x0_t + c1 * z + c2 * et)
    return xs, x0_preds


def ddim_batch(sd, config, x_T, x_cond, x_other, sampling_timesteps, betas=None, chunk=8):
    """B independent 64x64 patches, each the single-corner case of `ddim_overlapping`
    (corners=[(0,0)], p=H): what BASELINE.json's configs 0-3 run.  Per-image trajectories are
    independent, so this is the reference path applied image by image, batched through the UNet
    in chunks.  Returns (xs, x0_preds) as lists of (B,3,R,R)."""
    if betas is None:
        betas = beta_schedule(config)
    B = x_T.size(0)
    seq = timestep_seq(config.diffusion.num_diffusion_timesteps, sampling_timesteps)
    seq_next = [-1] + list(seq[:-1])
    xs, x0_preds = [x_T], []
    with torch.no_grad():
        for i_t, j_t in zip(reversed(seq), reversed(seq_next)):
            t = torch.ones(1) * i_t
            at, at_next = compute_alpha(betas, i_t), compute_alpha(betas, j_t)
            xt = xs[-1]
            et = torch.cat([unet_forward(sd, config,
                                         torch.cat([x_cond[i:i + chunk], xt[i:i + chunk]]
                                                   + ([x_other[i:i + chunk]] if x_other is not None else []), dim=1), t)
                            for i in range(0, B, chunk)], dim=0)
            x0_t = (xt - et * (1 - at).sqrt()) / at.sqrt()
            x0_preds.append(x0_t)
            xs.append(at_next.sqrt() * x0_t + (1 - at_next).sqrt() * et)
    return xs, x0_preds


# ----------------------------------------------------------------------------------------------
# a17  restoration glue                       (reference: models/restoration.py:63-168)
# ----------------------------------------------------------------------------------------------
def restore(sd, config, x01, x_T, sampling_timesteps, r=16, hfrm=None, keep=-5):
    """restoration.py:70-134 for one image: x01 (1,3,H,W) in [0,1] -> restored (1,3,H,W) in [0,1].

    `hfrm` is the out-of-path high-frequency module (restoration.py:94); identity stand-in when
    None.  `x_T` replaces the reference's `torch.randn` (restoration.py:177) so
    the result is reproducible.  Output = IDWT(cat[x0_preds[keep][:, :3], hfrm_wav[:, 3:]])."""
    p = config.data.image_size
    x_cond = dwt_fwd(data_transform(x01))
    hf = x01 if hfrm is None else hfrm(x01)
    hf_wav = dwt_fwd(data_transform(hf))
    x_other = hf_wav[:, config.model.other_channels_begin:]
    corners = grid_corners(x_cond.shape[2], x_cond.shape[3], p, r)
    xs, x0_preds = ddim_overlapping(sd, config, x_T, x_cond, x_other, corners, p, sampling_timesteps)
    pc = config.model.pred_channels
    out = torch.cat([x0_preds[keep][:, :pc], hf_wav[:, pc:]], dim=1)
    return inverse_data_transform(dwt_inv(out)), xs, x0_preds


def torch_psnr(tar, prd):          # utils/metrics.py:7-11
    d = torch.clamp(prd, 0, 1) - torch.clamp(tar, 0, 1)
    return 20 * torch.log10(1 / (d ** 2).mean().sqrt())


# ----------------------------------------------------------------------------------------------
# SURVEY.md §8(f)-1  HFRM high-frequency refinement module   (reference: models/arch.py:132-253)
# ----------------------------------------------------------------------------------------------
HFRM_DEFAULT = dict(in_channel=3, dim=32, mid_blk_num=6, enc_blk_nums=(2, 2, 2, 4), dec_blk_nums=(2, 2, 2, 2))   # ddm_wavelet.py:137-142


def layernorm2d(sd, name, x, eps=1e-6):
    """arch.py:7-43: per-pixel normalisation over the channel axis, biased variance, affine."""
    mu = x.mean(1, keepdim=True)
    var = (x - mu).pow(2).mean(1, keepdim=True)
    y = (x - mu) / (var + eps).sqrt()
    return sd[name + ".weight"].view(1, -1, 1, 1) * y + sd[name + ".bias"].view(1, -1, 1, 1)


def hfrm_block(sd, name, x):
    """ResidualBlock, arch.py:158-204."""
    dim = x.shape[1]
    h = layernorm2d(sd, name + ".norm1", x)
    h = conv(sd, name + ".conv1", h)                                                          # 1x1 dim -> 2 dim
    h = F.conv2d(h, sd[name + ".conv2.weight"], sd[name + ".conv2.bias"], padding=1, groups=2 * dim)   # depthwise 3x3
    h = h[:, :dim] * h[:, dim:]                                                               # SpatialAttn (gate)
    s = conv(sd, name + ".channel_attn.chan_conv", F.adaptive_avg_pool2d(h, 1))               # ChannelAttn
    h = h * s
    h = conv(sd, name + ".conv3", h)
    y = x + h * sd[name + ".beta"]
    h = conv(sd, name + ".conv4", layernorm2d(sd, name + ".norm2", y))
    h = h[:, :dim] * h[:, dim:]
    h = conv(sd, name + ".conv5", h)
    return y + h * sd[name + ".gamma"]


def hfrm_forward(sd, x, enc_blk_nums=(2, 2, 2, 4), mid_blk_num=6, dec_blk_nums=(2, 2, 2, 2)):
    """HFRM.forward, arch.py:235-253.  x: (B,3,H,W) raw [0,1] image, H and W multiples of 16."""
    B, C, H, W = x.shape
    inp = x
    x = conv(sd, "conv_in", x, padding=1)
    encs = []
    for i, num in enumerate(enc_blk_nums):
        for j in range(num):
            x = hfrm_block(sd, f"encoders.{i}.{j}", x)
        encs.append(x)
        x = conv(sd, f"downs.{i}", x, stride=2)                                               # 2x2 stride 2
    for j in range(mid_blk_num):
        x = hfrm_block(sd, f"mid_blks.{j}", x)
    for i, (num, skip) in enumerate(zip(dec_blk_nums, encs[::-1])):
        x = F.pixel_shuffle(F.conv2d(x, sd[f"ups.{i}.0.weight"]), 2)                          # 1x1 (no bias) + PixelShuffle(2)
        x = x + skip
        for j in range(num):
            x = hfrm_block(sd, f"decoders.{i}.{j}", x)
    x = conv(sd, "conv_out", x, padding=1)
    return (x + inp)[:, :, :H, :W]


# ------------------------------------------------------------------------------------------------
# SURVEY.md §8(f)-2  output metrics / 8-bit conversion / synthetic data directory  (reference: utils/metrics.py,
# utils/logging.py:9-12, datasets/raindrop.py)
# ------------------------------------------------------------------------------------------------
def psnr_torch(gt, out):
    """utils/metrics.py:7-11 on a (1,3,H,W) pair."""
    d = out.clamp(0, 1).double() - gt.clamp(0, 1).double()
    return float(20 * torch.log10(1 / (d ** 2).mean().sqrt()))


def psnr_y(gt, out):
    """utils/metrics.py:30-51 (calculate_psnr_in_GPU(.., test_y_channel=True)); the numpy calculate_psnr(.., True) on the same
    images scaled to 0..255 is the same number."""
    wts = torch.tensor([24.966, 128.553, 65.481], dtype=torch.float64)[None, :, None, None]
    ya = ((gt.double() * wts).sum(dim=1) + 16.0) / 255
    yb = ((out.double() * wts).sum(dim=1) + 16.0) / 255
    return float(20. * torch.log10(1. / torch.sqrt(((ya - yb) ** 2).mean())))


def to_u8_hwc(img):
    """torchvision.utils.save_image's quantisation (torchvision is absent from this image: restated from its documented
    behaviour `mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(uint8)`; parity for this one function is unpinned)."""
    return img.mul(255).add(0.5).clamp(0, 255).permute(0, 2, 3, 1).to(torch.uint8)


def pil_to_tensor(pic):
    """torchvision.transforms.ToTensor for uint8 PIL images (HWC uint8 -> CHW float /255)."""
    import numpy as np
    a = np.asarray(pic)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)


def synthetic_raindrop_dir(root, seed=303, sizes=((1000, 640), (300, 500), (720, 480))):
    """Write <root>/raindrop/raindrop_test/{input/<k>_rain.png, gt/<k>_clean.png}: smooth seeded random images of the given
    (width, height) sizes.  Returns the sizes.  Used by the golden generator and by the tests (same seed -> same files)."""
    import os
    import numpy as np
    from PIL import Image
    rng = np.random.Generator(np.random.PCG64(seed))
    for sub in ("input", "gt"):
        os.makedirs(os.path.join(root, "raindrop", "raindrop_test", sub), exist_ok=True)
    for k, (w, h) in enumerate(sizes):
        base = rng.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
        clean = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR))
        rain = np.clip(clean.astype(np.int32) + rng.integers(-40, 41, size=clean.shape), 0, 255).astype(np.uint8)
        Image.fromarray(rain).save(os.path.join(root, "raindrop", "raindrop_test", "input", f"{k}_rain.png"))
        Image.fromarray(clean).save(os.path.join(root, "raindrop", "raindrop_test", "gt", f"{k}_clean.png"))
    return [list(s) for s in sizes]


# ------------------------------------------------------------------------------------------------
# SURVEY.md §8(f)-3  training step   (reference: models/ddm_wavelet.py:108-124 loss, :200-272 step, :34-60 EMA,
# utils/optimize.py:5-8 Adam(lr, betas=(0.9,0.999), eps, weight_decay, amsgrad=False))
# ------------------------------------------------------------------------------------------------
def make_grid(tensor, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid (torchvision==0.9.0 per the reference's requirements.txt:9; the package is absent from this image,
    so this restates its published algorithm) as called at ddm_wavelet.py:409: normalize=False, scale_each=False.
    xmaps = min(nrow, N), ymaps = ceil(N / xmaps); cell = (H + padding, W + padding); the canvas is filled with pad_value and image k
    is narrowed into row k // xmaps, column k % xmaps at offset `padding`."""
    import math
    if tensor.dim() == 2:
        tensor = tensor.unsqueeze(0)
    if tensor.dim() == 3:
        if tensor.size(0) == 1:
            tensor = torch.cat((tensor, tensor, tensor), 0)
        tensor = tensor.unsqueeze(0)
    if tensor.dim() == 4 and tensor.size(1) == 1:
        tensor = torch.cat((tensor, tensor, tensor), 1)
    if tensor.size(0) == 1:
        return tensor.squeeze(0)
    nmaps = tensor.size(0)
    xmaps = min(nrow, nmaps)
    ymaps = int(math.ceil(float(nmaps) / xmaps))
    height, width = int(tensor.size(2) + padding), int(tensor.size(3) + padding)
    grid = tensor.new_full((tensor.size(1), height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for yy in range(ymaps):
        for xx in range(xmaps):
            if k >= nmaps:
                break
            grid.narrow(1, yy * height + padding, height - padding).narrow(2, xx * width + padding, width - padding).copy_(tensor[k])
            k += 1
    return grid


def noise_estimation_loss(sd, config, x0, t, e, betas):
    """ddm_wavelet.py:108-124 for the raindrop_wavelet.yml branch (use_other_channels, inp_channels = 48, pred_channels = 3).
    x0: (B, 96, R, R) = [x_cond 48 | gt LL 3 | other 45]; t: (B,) long; e: (B,3,R,R).  -> (simple_loss, output, x0_pred, mse_loss)"""
    m = config.model
    inp, pc = m.in_channels, m.pred_channels
    a = (1 - betas).cumprod(dim=0).index_select(0, t).view(-1, 1, 1, 1)
    x_inp, x_tar, x_other = x0[:, :inp], x0[:, inp:inp + pc], x0[:, inp + pc:]
    xt = x_tar * a.sqrt() + e * (1.0 - a).sqrt()
    output = unet_forward(sd, config, torch.cat([x_inp, xt, x_other], dim=1), t.float())
    x0_pred = (xt - output * (1 - a).sqrt()) / a.sqrt()
    simple_loss = (e - output).square().sum(dim=(1, 2, 3))
    mse_loss = (x_tar - x0_pred).square().sum(dim=(1, 2, 3))
    return simple_loss.mean(dim=0), output, x0_pred, mse_loss.mean(dim=0)


def train_grads(sd, config, x0, t, e, betas, use_mse=False):
    """loss.backward() of the step above -- mse_loss.backward() with training.use_mse (ddm_wavelet.py:263-266): -> (loss, output,
    {name: grad}) by torch autograd over the functional forward.  The returned loss is the noise-space one either way."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    with torch.enable_grad():
        loss, output, _, mse = noise_estimation_loss(leaf, config, x0, t, e, betas)
        (mse if use_mse else loss).backward()
    return loss.detach(), output.detach(), {k: v.grad for k, v in leaf.items()}


def adam_step(p, g, m, v, step, lr=4e-5, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam (amsgrad=False), one parameter tensor, `step` counted from 1.  Returns (p, m, v)."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    return p - (lr / bc1) * (m / denom), m, v


def ema_update(shadow, p, mu=0.9999):
    """EMAHelper.update, ddm_wavelet.py:48-53 (the reference constructs EMAHelper() with its default mu = 0.9999)."""
    return (1.0 - mu) * p + mu * shadow
