"""Procedural (seeded, reproducible) weights and synthetic inputs.

No pretrained checkpoint ships with the reference (the HFRM file
`saved_models/raindrop/lastest.pth` and the DDPM ckpt are both absent) and 156 M random weights
cannot be committed, so tests, golden fixtures and bench.py all re-create the *same* weights from a
named, seeded generator on whatever box they run on (SURVEY.md §8c / Appendix B recipe):

    for each state_dict entry `name`:  rng = numpy PCG64(seed ^ crc32(name))
        ndim > 1           -> N(0,1) / sqrt(fan_in)        (conv OIHW, Linear [out,in])
        1-D "*.weight"     -> 1 + 0.1 N(0,1)               (GroupNorm gamma)
        everything else    -> 0.1 N(0,1)                   (biases, GroupNorm beta)

The state_dict key list / shapes follow the reference model's parameter naming
(`models/unet.py:196-307`): they are generated here from the config alone.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np


# ----------------------------------------------------------------------------------------------
# config helpers
# ----------------------------------------------------------------------------------------------
def dict2namespace(d):
    """YAML dict -> nested namespace (same contract as the reference's `eval_diffusion.py:47-55`)."""
    ns = SimpleNamespace()
    for k, v in d.items():
        setattr(ns, k, dict2namespace(v) if isinstance(v, dict) else v)
    return ns


def raindrop_wavelet_config(image_size: int = 64, ch: int = 128, ch_mult=(1, 2, 4, 6),
                            num_res_blocks: int = 2, attn_resolutions=(16,)):
    """Every key of `configs/raindrop_wavelet.yml` with its values (checked against the reference's file by tests/golden/make_golden.py),
    except the two deployment-specific ones: data.data_dir ("" here) and data.num_workers (0 here)."""
    return dict2namespace({
        "data": {"dataset": "RainDrop", "image_size": image_size, "patch_size": image_size * 4,
                 "lap": False, "global_attn": False, "wavelet": True, "wavelet_in_unet": False,
                 "use_window": False, "window_size": 2, "begin_from_noise": True,
                 "num_workers": 0, "data_dir": "", "conditional": True},
        "model": {"pred_channels": 3, "use_other_channels": True, "other_channels_begin": 3,
                  "use_gt_in_train": True, "in_channels": 48, "out_ch": 3, "ch": ch,
                  "ch_mult": list(ch_mult), "num_res_blocks": num_res_blocks,
                  "attn_resolutions": list(attn_resolutions), "dropout": 0.0, "ema_rate": 0.999,
                  "ema": True, "resamp_with_conv": True},
        "diffusion": {"beta_schedule": "linear", "beta_start": 0.0001, "beta_end": 0.02,
                      "num_diffusion_timesteps": 1000},
        "training": {"use_mse": False, "patch_n": 8, "batch_size": 1, "n_epochs": 38000, "n_iters": 2000000,
                     "snapshot_freq": 3000, "validation_freq": 3000},
        "sampling": {"batch_size": 1, "last_only": True},
        "optim": {"weight_decay": 0.0, "optimizer": "Adam", "lr": 0.00004, "amsgrad": False, "eps": 0.00000001},
    })


def reduced_config():
    """Reduced-width fixture model of SURVEY.md §8c (1.03 M params, attention at res 8)."""
    return raindrop_wavelet_config(image_size=16, ch=32, ch_mult=(1, 2), attn_resolutions=(8,))


def variant_config(kind):
    """Reduced configs exercising the optional branches of `models/unet.py` + the NCHW input shape each takes:
    "no_other" (model.use_other_channels False: 48 + 3 input channels), "window" (data.use_window: 6 image channels at 32x32 ->
    2 x 12 window channels at 16x16, 12 -> 3 back), "wavelet_in_unet" (DWT/IDWT inside the model: 6 channels at 64x64 -> 96 at 16x16)."""
    c = reduced_config()
    if kind == "no_other":
        c.model.use_other_channels = False
        return c, (2, 51, 16, 16)
    if kind == "window":
        c.data.use_window, c.data.window_size = True, 2
        c.model.use_other_channels, c.model.in_channels, c.model.pred_channels, c.model.out_ch = False, 12, 12, 12
        return c, (2, 6, 32, 32)
    if kind == "wavelet_in_unet":
        c.data.wavelet_in_unet = True
        c.model.use_other_channels, c.model.in_channels, c.model.pred_channels, c.model.out_ch = False, 48, 48, 48
        return c, (2, 6, 64, 64)
    raise ValueError(kind)


VARIANTS = ("no_other", "window", "wavelet_in_unet")


def unet_in_channels(config) -> int:
    m = config.model
    if m.use_other_channels:
        return m.in_channels * 2 + m.pred_channels - m.other_channels_begin
    return m.in_channels + m.pred_channels


# ----------------------------------------------------------------------------------------------
# state_dict layout (names + shapes) from the config alone
# ----------------------------------------------------------------------------------------------
def unet_param_shapes(config) -> "OrderedDict[str, tuple]":
    """Ordered {state_dict key: shape} of the wavelet-domain DiffusionUNet for `config`.

    Mirrors the module tree of the reference (`models/unet.py:197-307`): temb.dense.{0,1},
    conv_in, down.{l}.block.{b}.*, down.{l}.attn.{b}.*, down.{l}.downsample.conv, mid.*,
    up.{l}.block.{b}.*, up.{l}.attn.{b}.*, up.{l}.upsample.conv, norm_out, conv_out.
    Registration order matters only for reproducing `state_dict()` key order; values are keyed
    by name.
    """
    m = config.model
    ch, out_ch, ch_mult = m.ch, m.out_ch, tuple(m.ch_mult)
    nrb, attn_res = m.num_res_blocks, list(m.attn_resolutions)
    in_ch = unet_in_channels(config)
    temb_ch = ch * 4
    shapes: "OrderedDict[str, tuple]" = OrderedDict()

    def conv(name, cin, cout, k):
        shapes[name + ".weight"] = (cout, cin, k, k)
        shapes[name + ".bias"] = (cout,)

    def lin(name, cin, cout):
        shapes[name + ".weight"] = (cout, cin)
        shapes[name + ".bias"] = (cout,)

    def norm(name, c):
        shapes[name + ".weight"] = (c,)
        shapes[name + ".bias"] = (c,)

    def resblock(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cin, cout, 3)
        lin(name + ".temb_proj", temb_ch, cout)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".nin_shortcut", cin, cout, 1)

    def attn(name, c):
        norm(name + ".norm", c)
        for p in ("q", "k", "v", "proj_out"):
            conv(name + "." + p, c, c, 1)

    lin("temb.dense.0", ch, temb_ch)
    lin("temb.dense.1", temb_ch, temb_ch)
    conv("conv_in", in_ch, ch, 3)

    res = config.data.image_size
    in_ch_mult = (1,) + ch_mult
    nres = len(ch_mult)
    block_in = None
    for l in range(nres):
        block_in = ch * in_ch_mult[l]
        block_out = ch * ch_mult[l]
        for b in range(nrb):
            resblock(f"down.{l}.block.{b}", block_in, block_out)
            block_in = block_out
        # the reference registers `down.block` (all blocks) before `down.attn`
        if res in attn_res:
            for b in range(nrb):
                attn(f"down.{l}.attn.{b}", block_out)
        if l != nres - 1:
            conv(f"down.{l}.downsample.conv", block_in, block_in, 3)
            res //= 2
    resblock("mid.block_1", block_in, block_in)
    attn("mid.attn_1", block_in)
    resblock("mid.block_2", block_in, block_in)

    up = {}
    for l in reversed(range(nres)):
        entries: "OrderedDict[str, tuple]" = OrderedDict()
        saved, shapes = shapes, entries
        block_out = ch * ch_mult[l]
        skip_in = ch * ch_mult[l]
        for b in range(nrb + 1):
            if b == nrb:
                skip_in = ch * in_ch_mult[l]
            resblock(f"up.{l}.block.{b}", block_in + skip_in, block_out)
            block_in = block_out
        if res in attn_res:
            for b in range(nrb + 1):
                attn(f"up.{l}.attn.{b}", block_out)
        if l != 0:
            conv(f"up.{l}.upsample.conv", block_in, block_in, 3)
            res *= 2
        shapes = saved
        up[l] = entries
    for l in range(nres):  # the reference prepends, so state_dict order is up.0, up.1, ...
        shapes.update(up[l])
    norm("norm_out", block_in)
    conv("conv_out", block_in, out_ch, 3)
    return shapes


def unet_global_param_shapes(config) -> "OrderedDict[str, tuple]":
    """Ordered {state_dict key: shape} of `DiffusionUNet_Global` (models/unet.py:463-583) in the reference's registration order: temb,
    conv_in, down, global_conv_in, down_global, mid, up, up_global, norm_out, conv_out.  Its input is [x_cond | x_t] (2 x in_channels when
    data.conditional, :473) plus the whole image `x_global` with model.in_channels channels (:506)."""
    m = config.model
    ch, ch_mult = m.ch, tuple(m.ch_mult)
    nres = len(ch_mult)
    in_ch_mult = (1,) + ch_mult
    base = OrderedDict(unet_param_shapes(_global_base_config(config)))
    out: "OrderedDict[str, tuple]" = OrderedDict()

    def take(prefix):
        for k in [k for k in base if k.startswith(prefix)]:
            out[k] = base.pop(k)

    def conv(name, cin, cout, k, transposed=False, depthwise=False):
        out[name + ".weight"] = (cin, cout, k, k) if transposed else ((cout, 1, k, k) if depthwise else (cout, cin, k, k))
        out[name + ".bias"] = (cout,)

    def attn_global(name, c, lp=2, gp=8):
        for n in ("norm_patch", "norm_global"):
            out[f"{name}.{n}.weight"] = (c,)
            out[f"{name}.{n}.bias"] = (c,)
        conv(name + ".q", c, c, lp)
        conv(name + ".k", c, c, gp, depthwise=True)
        conv(name + ".v", c, c, gp, depthwise=True)
        conv(name + ".proj_out", c, c, 1)

    take("temb.")
    take("conv_in.")
    take("down.")
    conv("global_conv_in", m.in_channels, ch, 3)
    for l in range(nres):
        block_in, block_out = ch * in_ch_mult[l], ch * ch_mult[l]
        if l != nres - 1:
            conv(f"down_global.{l}.conv", block_in, block_out, 4)
        attn_global(f"down_global.{l}.attn", block_out)
    take("mid.")
    take("up.")
    block_in = ch * ch_mult[-1]
    ups = {}
    for l in reversed(range(nres)):
        block_out = ch * ch_mult[l]
        saved, out = out, OrderedDict()
        if l != 0:
            conv(f"up_global.{l}.conv", block_in, block_out, 4, transposed=True)
        attn_global(f"up_global.{l}.attn", block_out)
        ups[l], out = out, saved
        block_in = block_out
    for l in range(nres):                      # `self.up_global.insert(0, ...)`: state_dict order is up_global.0, up_global.1, ...
        out.update(ups[l])
    take("norm_out.")
    take("conv_out.")
    assert not base, list(base)[:3]
    return out


def _global_base_config(config):
    """The plain-UNet view of a global_attn config: same trunk, input = 2 x in_channels (conditional) and no `other` channels."""
    import copy
    c = copy.deepcopy(config)
    c.model.use_other_channels = False
    c.model.pred_channels = c.model.in_channels if getattr(c.data, "conditional", True) else 0
    return c


def global_config():
    """Reduced `data.global_attn: True` fixture: 16x16 patches with a 32x32 whole-image map (2x2 = 4 key / value tokens on the way down,
    4x4 = 16 on the way up).  ch_mult = (1, 1): the reference's DiffusionUNet_Global only runs when the last level keeps the channel count
    (its last `down_global.attn` normalises the un-convolved whole-image map with the last level's GroupNorm, unet.py:609-610)."""
    c = raindrop_wavelet_config(image_size=16, ch=32, ch_mult=(1, 1), attn_resolutions=(8,))
    c.data.global_attn = True
    c.model.use_other_channels = False
    c.model.in_channels, c.model.pred_channels, c.model.out_ch = 3, 3, 3
    return c


def procedural_global_state_dict(config, seed: int = 61):
    import torch
    return OrderedDict((k, torch.from_numpy(procedural_tensor(k, s, seed))) for k, s in unet_global_param_shapes(config).items())


# ----------------------------------------------------------------------------------------------
# procedural values
# ----------------------------------------------------------------------------------------------
def procedural_tensor(name: str, shape, seed: int = 61) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64((seed ^ zlib.crc32(name.encode())) & 0xFFFFFFFF))
    z = rng.standard_normal(size=shape, dtype=np.float64)
    if len(shape) > 1:
        fan_in = int(np.prod(shape[1:]))
        w = z / np.sqrt(fan_in)
    elif name.endswith(".weight"):
        w = 1.0 + 0.1 * z
    else:
        w = 0.1 * z
    return w.astype(np.float32)


def procedural_state_dict(config, seed: int = 61):
    """OrderedDict name -> torch.float32 CPU tensor for the UNet of `config`."""
    import torch
    sd = OrderedDict()
    for name, shape in unet_param_shapes(config).items():
        sd[name] = torch.from_numpy(procedural_tensor(name, shape, seed))
    return sd


def synthetic_batch(batch: int, patch_px: int = 256, seed: int = 61):
    """Synthetic raindrop crops: rainy ~ U[0,1) of shape (B,3,px,px) and the
    start noise x_T ~ N(0,1) of shape (B,3,px/4,px/4); HFRM output stand-in = rainy (identity).
    Returned as CPU float32 torch tensors so every box regenerates identical inputs."""
    import torch
    g = torch.Generator().manual_seed(seed)
    rainy = torch.rand(batch, 3, patch_px, patch_px, generator=g, dtype=torch.float32)
    x_T = torch.randn(batch, 3, patch_px // 4, patch_px // 4, generator=g, dtype=torch.float32)
    return rainy, x_T


# ----------------------------------------------------------------------------------------------
# HFRM (models/arch.py:206-253) parameter layout + procedural values
# ----------------------------------------------------------------------------------------------
def hfrm_param_shapes(in_channel=3, dim=32, mid_blk_num=6, enc_blk_nums=(2, 2, 2, 4), dec_blk_nums=(2, 2, 2, 2)):
    """Ordered {state_dict key: shape} of the reference HFRM (448 tensors, 15.94 M parameters at the defaults)."""
    shapes = OrderedDict()

    def block(name, d):
        shapes[name + ".beta"] = (1, d, 1, 1)
        shapes[name + ".gamma"] = (1, d, 1, 1)
        for cname, co, ci, k in (("conv1", 2 * d, d, 1), ("conv2", 2 * d, 1, 3), ("conv3", d, d, 1),
                                 ("channel_attn.chan_conv", d, d, 1), ("conv4", 2 * d, d, 1), ("conv5", d, d, 1)):
            shapes[f"{name}.{cname}.weight"] = (co, ci, k, k)
            shapes[f"{name}.{cname}.bias"] = (co,)
        for n in ("norm1", "norm2"):
            shapes[f"{name}.{n}.weight"] = (d,)
            shapes[f"{name}.{n}.bias"] = (d,)

    shapes["conv_in.weight"] = (dim, in_channel, 3, 3)
    shapes["conv_in.bias"] = (dim,)
    d = dim
    enc, downs = OrderedDict(), OrderedDict()
    saved = shapes
    for i, num in enumerate(enc_blk_nums):
        shapes = enc
        for j in range(num):
            block(f"encoders.{i}.{j}", d)
        downs[f"downs.{i}.weight"] = (2 * d, d, 2, 2)
        downs[f"downs.{i}.bias"] = (2 * d,)
        d *= 2
    mid = OrderedDict()
    shapes = mid
    for j in range(mid_blk_num):
        block(f"mid_blks.{j}", d)
    ups, dec = OrderedDict(), OrderedDict()
    for i, num in enumerate(dec_blk_nums):
        ups[f"ups.{i}.0.weight"] = (2 * d, d, 1, 1)
        d //= 2
        shapes = dec
        for j in range(num):
            block(f"decoders.{i}.{j}", d)
    shapes = saved
    # registration order of the reference: conv_in, encoders, decoders, mid_blks, ups, downs, conv_out
    for part in (enc, dec, mid, ups, downs):
        shapes.update(part)
    shapes["conv_out.weight"] = (in_channel, d, 3, 3)
    shapes["conv_out.bias"] = (in_channel,)
    return shapes


def procedural_hfrm_state_dict(seed: int = 61, **kw):
    import torch
    return OrderedDict((k, torch.from_numpy(procedural_tensor("hfrm." + k, s, seed))) for k, s in hfrm_param_shapes(**kw).items())
