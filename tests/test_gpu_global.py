"""GPU parity of the optional `data.global_attn` model: DiffusionUNet_Global and one Attn_Global against the reference's
own outputs (tests/golden/global.npz, written by tests/golden/make_golden.py while asserting oracle == reference), and the `use_global`
branch of the stitched sampler against the oracle."""
import numpy as np
import pytest
import torch

from conftest import rel_linf

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TOL = {"f32": 1e-3, "f32x3": 1e-3, "bf16": 2e-2}       # bf16: the reduced 32-channel fixture (see test_gpu_unet.test_reduced_unet_forward)


def seeded(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float32)


@pytest.mark.parametrize("dtype", ["f32", "f32x3", "bf16"])
def test_unet_global_forward(golden, dtype):
    import wavedm_amd
    from wavedm_amd import procedural as P
    g = golden("global.npz")
    cfg = P.global_config()
    net = wavedm_amd.DiffusionUNet_Global(cfg, dtype=dtype)
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["names"]]           # the reference's state_dict order
    net.load_state_dict(P.procedural_global_state_dict(cfg, seed=61), strict=True)
    net = net.cuda()
    x, xg, t = seeded((2, 6, 16, 16), 710), seeded((2, 3, 32, 32), 711), torch.tensor([400.0, 30.0])
    y = net(x.cuda(), t, xg.cuda())
    e = rel_linf(y.cpu(), g["y"])
    print(f"DiffusionUNet_Global {dtype}: rel_linf vs the reference {e:.3e}")
    assert tuple(y.shape) == g["y"].shape and e <= TOL[dtype]


def test_attn_global_block(golden):
    import wavedm_amd
    from wavedm_amd import procedural as P
    g = golden("global.npz")
    cfg = P.global_config()
    net = wavedm_amd.DiffusionUNet_Global(cfg, dtype="f32")
    net.load_state_dict(P.procedural_global_state_dict(cfg, seed=61), strict=True)
    net = net.cuda()
    xp, xq = seeded((2, 32, 16, 16), 712).cuda(), seeded((2, 32, 32, 32), 713).cuda()
    y = net._attn_global(net._sd(), "down_global.0.attn", xp, xq)
    assert rel_linf(y.cpu(), g["attn"]) <= 1e-4


def test_global_operators_vs_torch():
    """The four fp32 operators of csrc/global_attn.hip on their own, against torch on the host: strided / transposed / depthwise convolutions
    with odd sizes, GroupNorm (+SiLU), cross attention with 1, 5 and 64 keys, nearest upsample + residual."""
    import ctypes as C
    import torch.nn.functional as F
    from wavedm_amd import _lib
    L, h = _lib.lib(), _lib.handle(0)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = _lib.stream_ptr
    for (B, cin, H, W, cout, k, s, pad, groups, tr) in [(2, 8, 10, 12, 6, 4, 2, 1, 1, 0), (1, 6, 7, 9, 4, 4, 2, 1, 1, 1), (3, 5, 8, 8, 7, 2, 2, 0, 1, 0),
                                                        (2, 16, 16, 24, 16, 8, 8, 0, 16, 0), (1, 3, 9, 9, 5, 3, 1, 1, 1, 0)]:
        x = seeded((B, cin, H, W), 1)
        w = seeded((cin, cout, k, k) if tr else (cout, cin // groups, k, k), 2) * 0.2
        b = seeded((cout,), 3)
        want = F.conv_transpose2d(x, w, b, stride=s, padding=pad) if tr else F.conv2d(x, w, b, stride=s, padding=pad, groups=groups)
        y = torch.empty(want.shape, device="cuda")
        xd, wd, bd = x.cuda(), w.cuda(), b.cuda()                 # named: a temporary's memory is recycled by the next .cuda()
        _lib.check(L.wdm_conv2d_direct(h, p(xd), p(wd), p(bd), B, cin, H, W, cout, k, s, pad, groups, tr, p(y), st()))
        assert rel_linf(y.cpu(), want) <= 1e-5, (cin, cout, k, s, tr)
    x, gw, gb = seeded((2, 64, 6, 10), 4) * 3 + 1, seeded((64,), 5), seeded((64,), 6)
    for silu in (0, 1):
        want = F.group_norm(x, 32, gw, gb, eps=1e-6)
        want = want * torch.sigmoid(want) if silu else want
        y = torch.empty_like(x, device="cuda")
        xd, gwd, gbd = x.cuda(), gw.cuda(), gb.cuda()
        _lib.check(L.wdm_groupnorm(h, p(xd), p(gwd), p(gbd), 2, 64, 6, 10, 1e-6, silu, p(y), st()))
        assert rel_linf(y.cpu(), want) <= 1e-5
    for (B, Cc, nq, nk) in [(2, 32, 100, 5), (1, 64, 64, 64), (3, 8, 7, 1), (2, 48, 130, 330), (1, 16, 64, 65)]:      # 330 keys: a 480 x 720 image's pooled tokens (key blocks of 64)
        q, k, v = seeded((B, Cc, nq), 7), seeded((B, Cc, nk), 8), seeded((B, Cc, nk), 9)
        wgt = F.softmax(torch.bmm(q.permute(0, 2, 1), k) * (Cc ** -0.5), dim=2)
        want = torch.bmm(v, wgt.permute(0, 2, 1))
        y = torch.empty(B, Cc, nq, device="cuda")
        qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
        _lib.check(L.wdm_cross_attention(h, p(qd), p(kd), p(vd), B, Cc, nq, nk, p(y), st()))
        assert rel_linf(y.cpu(), want) <= 1e-5, (Cc, nq, nk)
    x, hp = seeded((2, 4, 6, 8), 10), seeded((2, 4, 3, 4), 11)
    y = torch.empty_like(x, device="cuda")
    xd, hd = x.cuda(), hp.cuda()
    _lib.check(L.wdm_upsample_add(h, p(xd), p(hd), 2, 4, 6, 8, 2, p(y), st()))
    assert torch.equal(y.cpu(), x + F.interpolate(hp, scale_factor=2.0, mode="nearest"))


def test_sampler_use_global():
    """`sample_image(..., total=, use_global=True)` (ddm_wavelet.py:479-483): stitched DDIM over a 24x28 image with 16x16 patches every 4,
    every patch attending to the 32x32 whole-image map, against the oracle's restatement of the same loop."""
    from types import SimpleNamespace
    import wavedm_amd
    from oracle import wavedm_oracle as O
    from wavedm_amd import procedural as P
    cfg = P.global_config()
    cfg.device = torch.device("cuda", 0)
    args = SimpleNamespace(resume="", sampling_timesteps=4, local_rank=0, image_folder="/tmp/wdm_img", test_set="raindrop", grid_r=4)
    d = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="f32")
    sd = P.procedural_global_state_dict(cfg, seed=61)
    d.model.load_state_dict(sd, strict=True)
    x_cond, x_T, total = seeded((1, 3, 24, 28), 21), seeded((1, 3, 24, 28), 22), seeded((1, 3, 32, 32), 23)
    corners = [(i, j) for i in (0, 4, 8) for j in (0, 4, 8, 12)]
    xs, x0 = d.sample_image(x_cond.cuda(), x_T.cuda(), last=False, patch_locs=corners, patch_size=16, total=total.cuda(), use_global=True)
    # oracle: the same loop with the oracle's model (eta = 0)
    betas = O.beta_schedule(cfg)
    seq = O.timestep_seq(cfg.diffusion.num_diffusion_timesteps, 4)
    xt = x_T.clone()
    mask = O.overlap_count_mask(24, 28, 16, corners)
    for i_t, j_t in zip(reversed(seq), reversed([-1] + list(seq[:-1]))):
        at, an = O.compute_alpha(betas, i_t), O.compute_alpha(betas, j_t)
        et = torch.zeros_like(xt)
        for (hi, wi) in corners:
            inp = torch.cat([x_cond[:, :, hi:hi + 16, wi:wi + 16], xt[:, :, hi:hi + 16, wi:wi + 16]], dim=1)
            et[:, :, hi:hi + 16, wi:wi + 16] += O.unet_global_forward(sd, cfg, inp, torch.tensor([float(i_t)]), total)
        et = et / mask.float()
        x0_t = (xt - et * (1 - at).sqrt()) / at.sqrt()
        xt = an.sqrt() * x0_t + (1 - an).sqrt() * et
    e = rel_linf(xs[-1].cpu(), xt)
    print(f"use_global stitched sampler f32: rel_linf vs the oracle {e:.3e}")
    assert len(xs) == 5 and len(x0) == 4 and e <= 1e-3
