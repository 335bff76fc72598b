"""GPU parity of the whole path: UNet forward, DDIM sampler, stitched restoration -- against the
golden vectors produced by the reference (tests/golden) and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_linf

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

# f32 = the parity mode (north_star: 1e-3 max-norm-relative).  bf16 = the throughput mode: bounded at <= 2x what this suite measures on
# the MI355X (printed by every test; 3.5e-3 ... 4.6e-3 on the full model, worst case 6.0e-3 on the reduced one), so a regression shows.
TOL = {"f32": 1e-3, "f32x3": 1e-3, "f16": 1e-3, "bf16": 1e-2}
# f16 on ONE forward of the full-width UNet: three 11-bit roundings per conv (packed weight, stored tensor, staged activation) add up to 1.1e-3 ... 1.3e-3 of max|eps|
# on the procedural weights (scripts/f16_error_budget.py reproduces it on the CPU oracle: each site alone ~8e-4, no layer dominates), just above north_star's bound;
# what the reference's user sees -- the sampler's xs[-1] / x0_preds[-5] and the restored image -- is held to 1e-3 (TOL) in every sampler test below.
TOL_FWD = dict(TOL, f16=1.5e-3)


def seeded(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def build(cfg, dtype):
    import wavedm_amd
    from wavedm_amd import procedural as P
    net = wavedm_amd.DiffusionUNet(cfg, dtype=dtype)
    net.load_state_dict(P.procedural_state_dict(cfg), strict=True)
    return net.cuda()


def make_diffusion(cfg, dtype, S, generator=lambda x: x):
    from types import SimpleNamespace
    import wavedm_amd
    from wavedm_amd import procedural as P
    cfg.device = torch.device("cuda", 0)
    args = SimpleNamespace(resume="", sampling_timesteps=S, local_rank=0, image_folder="/tmp/wdm_img", test_set="raindrop", grid_r=16)
    d = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=generator, dtype=dtype)
    d.model.load_state_dict(P.procedural_state_dict(cfg), strict=True)
    return d, args


@pytest.mark.parametrize("dtype", ["f32", "f32x3", "f16", "bf16"])
def test_reduced_unet_forward(golden, dtype):
    from wavedm_amd import procedural as P
    r = golden("reduced.npz")
    net = build(P.reduced_config(), dtype)
    x96 = seeded((2, 96, 16, 16), 40).cuda()
    # the reduced model (32-channel levels, 8-channel GroupNorm groups; not a BASELINE config) is the noisiest case for 16-bit operands: 1.1e-2 measured on one
    # forward in bf16, 1.35e-3 in f16 (the eightfold smaller round-off); the BASELINE configs below hold f16 to 1e-3
    tol = {"bf16": 2e-2, "f16": 2.5e-3}.get(dtype, TOL[dtype])
    e1 = rel_linf(net(x96, torch.tensor([500.0])).cpu(), r["fwd_t500"])
    e2 = rel_linf(net(x96, torch.tensor([990.0, 10.0])).cpu(), r["fwd_t_per_image"])
    print(f"reduced forward {dtype}: rel_linf {e1:.3e} {e2:.3e}")
    assert e1 <= tol and e2 <= tol
    # batch-composition independence: image 1 alone == image 1 inside the batch (bit-for-bit)
    a = net(x96, torch.tensor([500.0]))
    b = net(x96[1:2].contiguous(), torch.tensor([500.0]))
    assert torch.equal(a[1:2], b)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_reduced_sampler(golden, dtype):
    from wavedm_amd import procedural as P
    r = golden("reduced.npz")
    d, _ = make_diffusion(P.reduced_config(), dtype, 10)
    rainy, x_T = P.synthetic_batch(2, patch_px=64)
    out, xs_last, x0m5 = d.restore_batch(rainy.cuda(), x_T.cuda())
    assert rel_linf(xs_last.cpu(), r["samp_xs_last"]) <= TOL[dtype]
    assert rel_linf(x0m5.cpu(), r["samp_x0_m5"]) <= TOL[dtype]
    assert out.shape == (2, 3, 64, 64) and float(out.min()) >= 0 and float(out.max()) <= 1
    # the reference call surface on one image == the batched path (bit-for-bit: same kernels, per-image order)
    xc = d.wavelet_dec(2 * rainy[:1].cuda() - 1)
    xs, x0 = d.sample_image(xc, x_T[:1].cuda(), x_other=xc[:, 3:].contiguous(), last=False, patch_locs=[(0, 0)], patch_size=16, use_other=True)
    assert len(xs) == 11 and len(x0) == 10
    assert torch.equal(xs[-1], xs_last[:1]) and torch.equal(x0[-5], x0m5[:1])
    # opt-in early stop: the 4 steps after x0_preds[-5] are skipped, the kept prediction is identical
    out_es, _, x0_es = d.restore_batch(rainy.cuda(), x_T.cuda(), early_stop=True)
    assert torch.equal(x0_es, x0m5) and torch.equal(out_es, out)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_stitched_restore(golden, dtype):
    """DiffusiveRestoration.restore on a 120x180 image (30x45 wavelet domain, 45 overlapping 16x16 patches, r=4)."""
    import wavedm_amd
    from wavedm_amd import procedural as P
    s = golden("stitch.npz")
    d, args = make_diffusion(P.reduced_config(), dtype, 6)
    g = torch.Generator().manual_seed(77)
    img = torch.rand(1, 3, 120, 180, generator=g)
    gt = torch.rand(1, 3, 120, 180, generator=g)
    rest = wavedm_amd.DiffusiveRestoration(d, args, d.config, save_images=False)
    x_T = torch.from_numpy(s["x_T"]).cuda()
    real_randn = torch.randn
    torch.randn = lambda *a, **k: x_T.clone()          # the draw restoration.py:177 made when the golden was written
    try:
        outs, psnrs = rest.restore([(torch.cat([img, gt], 1), "img0", torch.zeros(1))], validation="raindrop", r=4)
    finally:
        torch.randn = real_randn
    assert int(s["n_corners"]) == 45
    e = rel_linf(outs[0].cpu(), s["out"])
    print(f"stitched restore {dtype}: rel_linf of the clamped output vs the reference's {e:.3e}")
    if dtype == "f32":
        assert e <= TOL[dtype]
    else:
        # The clamped image is mostly saturated with untrained weights (a bf16-sized deviation flips pixels at the clamp), so the bf16
        # bound is taken BEFORE the clamp: x0_preds[-5] in the wavelet domain against the oracle on the same inputs.
        from oracle import wavedm_oracle as O
        sd = P.procedural_state_dict(d.config)
        _, _, x0_o = O.restore(sd, d.config, img, x_T.cpu(), 6, r=4)
        xc = d.wavelet_dec(2 * img.cuda() - 1)
        corners = O.grid_corners(30, 45, 16, 4)
        _, x0 = d.sample_image(xc, x_T, x_other=xc[:, 3:].contiguous(), last=False, patch_locs=corners, patch_size=16, use_other=True)
        e0 = rel_linf(x0[-5].cpu(), x0_o[-5])
        print(f"stitched restore bf16: rel_linf of x0_preds[-5] before the clamp {e0:.3e}")
        assert e0 <= TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "f32x3", "f16"])
def test_full_unet_forward_f32(golden, dtype):
    """The 156 M-parameter UNet against the reference's own output, in both parity modes (exact fp32 MFMA; fp32 tensors with every
    product as three bf16 MFMAs on hi/lo-split operands)."""
    from wavedm_amd import procedural as P
    f = golden("full.npz")
    cfg = P.raindrop_wavelet_config()
    net = build(cfg, dtype)
    assert sum(p.numel() for p in net.parameters()) == int(f["n_params"]) == 156492675
    rainy, x_T = P.synthetic_batch(4, patch_px=256)
    d = __import__("wavedm_amd").WaveletTransform(scale=2, dec=True)
    xc = d(2 * rainy.cuda() - 1)
    x96 = torch.cat([xc[:2], x_T[:2].cuda(), xc[:2, 3:]], dim=1)
    got = net(x96, torch.tensor([990.0]))
    e = rel_linf(got.cpu(), f["fwd_t990"])
    print(f"full UNet forward {dtype}: rel_linf vs the reference {e:.3e}")
    assert e <= TOL_FWD[dtype]


@pytest.mark.parametrize("dtype", ["f32", "f32x3", "f16", "bf16"])
def test_config0_sampler(golden, dtype):
    """BASELINE.json configs[0]: 4x64x64, 10 DDIM steps, full-width model, vs the reference's own output."""
    from wavedm_amd import procedural as P
    f = golden("full.npz")
    d, _ = make_diffusion(P.raindrop_wavelet_config(), dtype, 10)
    rainy, x_T = P.synthetic_batch(4, patch_px=256)
    out, xs_last, x0m5 = d.restore_batch(rainy.cuda(), x_T.cuda())
    e1, e2 = rel_linf(xs_last.cpu(), f["c0_xs_last"]), rel_linf(x0m5.cpu(), f["c0_x0_m5"])
    print(f"config0 {dtype}: rel_linf xs[-1] {e1:.3e}  x0[-5] {e2:.3e}")
    assert e1 <= TOL[dtype] and e2 <= TOL[dtype]
    assert torch.isfinite(out).all()


def test_c1_length_against_the_oracle():
    """BASELINE.json configs[1] at its real length: 100 DDIM steps on the full-width model, four crops.  The CPU oracle (pinned to the reference by the
    golden files; ~20 s of host time for 4 x 100 steps) is the yardstick for EVERY mode: f32 and f32x3 must stay inside north_star's 1e-3 over the whole
    trajectory -- not only over the 10 steps of config 0 -- and bf16 (the throughput mode of the headline number) inside 2x its measured deviation."""
    from oracle import wavedm_oracle as O
    from wavedm_amd import procedural as P
    cfg = P.raindrop_wavelet_config()
    sd = P.procedural_state_dict(cfg)                   # what make_diffusion loads
    rainy, x_T = P.synthetic_batch(4, patch_px=256)
    xc = O.dwt_fwd(2 * rainy - 1)
    xs_cpu, x0_cpu = O.ddim_batch(sd, cfg, x_T, xc, xc[:, 3:].contiguous(), 100, chunk=4)
    want_xs, want_x0 = xs_cpu[-1], x0_cpu[-5]
    res = {}
    for dtype in ("f32", "f32x3", "f16", "bf16"):
        d, _ = make_diffusion(cfg, dtype, 100)
        out, xs_last, x0m5 = d.restore_batch(rainy.cuda(), x_T.cuda())
        res[dtype] = (xs_last.cpu(), x0m5.cpu(), out.cpu())
        del d
        torch.cuda.empty_cache()
    err = {k: (rel_linf(v[0], want_xs), rel_linf(v[1], want_x0)) for k, v in res.items()}
    for k, (e1, e2) in err.items():
        print(f"C1 length (4 x 100 steps) {k} vs the oracle: rel_linf xs[-1] {e1:.3e}  x0[-5] {e2:.3e}")
    print(f"C1 length bf16 vs f32 (HIP): {rel_linf(res['bf16'][0], res['f32'][0]):.3e}")
    assert torch.isfinite(res["bf16"][2]).all()
    assert max(err["f32"]) <= 1e-3 and max(err["f32x3"]) <= 1e-3 and max(err["f16"]) <= 1e-3           # north_star's tolerance over the full trajectory
    assert max(err["bf16"]) <= 6e-3                                         # 2x the measured 2.7e-3 ... 3.0e-3


def test_full_size_properties_bf16():
    """BASELINE config-1 sizes (B=64, 64x64) through size-independent properties: determinism run to run,
    per-image independence of the batch, finiteness."""
    from wavedm_amd import procedural as P
    net = build(P.raindrop_wavelet_config(), "bf16")
    x = seeded((64, 96, 64, 64), 5).cuda()
    t = torch.tensor([730.0])
    a = net(x, t)
    b = net(x, t)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    c = net(x[17:25].contiguous(), t)
    assert torch.equal(a[17:25], c)


def test_config2_r128_forward():
    """BASELINE.json configs[2] geometry: 128x128 wavelet-domain patches (attention moves to the 768-channel level,
    N = 256 tokens, d = 768; 163.05 M params).  One forward of 2 patches vs the CPU oracle, both dtypes."""
    from oracle import wavedm_oracle as O
    from wavedm_amd import procedural as P
    cfg = P.raindrop_wavelet_config(image_size=128)
    sd = P.procedural_state_dict(cfg)
    assert sum(v.numel() for v in sd.values()) == 163053955
    x = seeded((2, 96, 128, 128), 7)
    t = torch.tensor([470.0])
    want = O.unet_forward(sd, cfg, x, t)
    for dtype in ("f32", "f32x3", "f16", "bf16"):
        import wavedm_amd
        net = wavedm_amd.DiffusionUNet(cfg, dtype=dtype)
        net.load_state_dict(sd, strict=True)
        got = net.cuda()(x.cuda(), t).cpu()
        e = rel_linf(got, want)
        print(f"config2 R=128 {dtype}: rel_linf {e:.3e}")
        assert e <= TOL_FWD[dtype]
        del net
        torch.cuda.empty_cache()


def test_config2_r128_sampler():
    """BASELINE.json configs[2] at SAMPLER level: 2 crops of 512x512 px -> 128x128 wavelet domain, 10 DDIM steps through
    restore_batch (DWT, the 163 M-parameter UNet with its N = 256 / d = 768 attention inside the loop -- models/unet.py:168-193, resolution
    assert :346-351 --, DDIM update, IDWT) against the CPU oracle's trajectory: xs[-1] and x0_preds[-5], f32 / f32x3 <= 1e-3, bf16 <= its bound."""
    from oracle import wavedm_oracle as O
    from wavedm_amd import procedural as P
    cfg = P.raindrop_wavelet_config(image_size=128)
    sd = P.procedural_state_dict(cfg)
    rainy, x_T = P.synthetic_batch(2, patch_px=512, seed=77)
    assert x_T.shape == (2, 3, 128, 128)
    xc = O.dwt_fwd(2 * rainy - 1)
    oxs, ox0 = O.ddim_batch(sd, cfg, x_T, xc, xc[:, 3:].contiguous(), 10, chunk=2)
    for dtype in ("f32", "f32x3", "f16", "bf16"):
        d, _ = make_diffusion(P.raindrop_wavelet_config(image_size=128), dtype, 10)
        out, xs_last, x0m5 = d.restore_batch(rainy.cuda(), x_T.cuda())
        e1, e2 = rel_linf(xs_last.cpu(), oxs[-1]), rel_linf(x0m5.cpu(), ox0[-5])
        print(f"config2 R=128 sampler {dtype}: rel_linf xs[-1] {e1:.3e}, x0_preds[-5] {e2:.3e}")
        assert e1 <= TOL[dtype] and e2 <= TOL[dtype]
        assert out.shape == (2, 3, 512, 512) and bool(torch.isfinite(out).all())
        del d
        torch.cuda.empty_cache()


def test_config2_b256_properties_bf16():
    """configs[2]'s full batch (256 patches of 128x128, bf16): the tile rules count workgroups, so what a batch of 2 exercises is not what a batch of
    256 runs (the B >= 100 rule of round 3 was such a case).  One UNet call at B = 256 twice (determinism), images 100-107 alone (an image's bits do
    not depend on the batch it sits in), finiteness; then 3 DDIM steps of all 256 through the sampler against the same 8 crops alone."""
    import wavedm_amd
    from wavedm_amd import procedural as P
    cfg = P.raindrop_wavelet_config(image_size=128)
    net = build(cfg, "bf16")
    x = seeded((256, 96, 128, 128), 123).cuda()
    t = torch.tensor([610.0])
    a = net(x, t)
    b = net(x, t)
    assert a.shape == (256, 3, 128, 128) and bool(torch.isfinite(a).all())
    assert torch.equal(a, b)
    c = net(x[100:108].contiguous(), t)
    assert torch.equal(a[100:108], c)
    del a, b, c, x, net
    torch.cuda.empty_cache()
    d, args = make_diffusion(P.raindrop_wavelet_config(image_size=128), "bf16", 3)
    rainy, x_T = P.synthetic_batch(256, patch_px=512, seed=78)
    rainy, x_T = rainy.cuda(), x_T.cuda()
    for mb in (64, 256):                                     # the bench's chunking, and one 256-patch UNet call per step
        args.max_batch = mb
        out, xs_last, x0 = d.restore_batch(rainy, x_T, keep=-1)
        assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(xs_last).all())
        o8, xs8, x08 = d.restore_batch(rainy[100:108].contiguous(), x_T[100:108].contiguous(), keep=-1)
        assert torch.equal(xs_last[100:108], xs8) and torch.equal(x0[100:108], x08) and torch.equal(out[100:108], o8)
        del out, xs_last, x0


@pytest.mark.parametrize("dtype", ["f32", "f32x3", "f16", "bf16"])
def test_config4_fullres_stitch(dtype):
    """BASELINE.json configs[4] geometry: one 480x720 image -> 120x180 wavelet domain -> 45 overlapping 64x64 patches
    (r = 16) through the full-width UNet, DiffusiveRestoration.restore end to end, 5 DDIM steps, vs the CPU oracle."""
    import wavedm_amd
    from oracle import wavedm_oracle as O
    from wavedm_amd import procedural as P
    cfg = P.raindrop_wavelet_config()
    sd = P.procedural_state_dict(cfg)
    d, args = make_diffusion(cfg, dtype, 5)
    g = torch.Generator().manual_seed(11)
    img = torch.rand(1, 3, 480, 720, generator=g)
    gt = torch.rand(1, 3, 480, 720, generator=g)
    x_T = torch.randn(1, 3, 120, 180, generator=g)
    rest = wavedm_amd.DiffusiveRestoration(d, args, d.config, save_images=False)
    real_randn = torch.randn
    x_T_dev = x_T.cuda()
    torch.randn = lambda *a, **k: x_T_dev.clone()
    try:
        outs, _ = rest.restore([(torch.cat([img, gt], 1), "full0", torch.zeros(1))], validation="raindrop", r=16)
    finally:
        torch.randn = real_randn
    assert outs[0].shape == (1, 3, 480, 720)
    want, xs, x0 = O.restore(sd, cfg, img, x_T, 5, r=16)
    assert len(O.grid_corners(120, 180, 64, 16)) == 45
    e = rel_linf(outs[0].cpu(), want)
    print(f"config4 480x720 {dtype}: rel_linf of the clamped output {e:.3e}")
    if dtype in ("f32", "f32x3"):
        assert e <= 1e-3
    else:
        # 16-bit operands: bound x0_preds[-5] before the clamp (see test_stitched_restore)
        xc = d.wavelet_dec(2 * img.cuda() - 1)
        _, x0g = d.sample_image(xc, x_T_dev, x_other=xc[:, 3:].contiguous(), last=False, patch_locs=O.grid_corners(120, 180, 64, 16), patch_size=64,
                                use_other=True)
        e0 = rel_linf(x0g[-5].cpu(), x0[-5])
        print(f"config4 480x720 {dtype}: rel_linf of x0_preds[-5] before the clamp {e0:.3e}")
        assert e0 <= TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("kind", ["no_other", "window", "wavelet_in_unet"])
def test_optional_unet_branches(golden, kind, dtype):
    """the three optional branches of models/unet.py against the reference's own outputs."""
    import wavedm_amd
    from wavedm_amd import procedural as P
    v = golden("variants.npz")
    cfg, shape = P.variant_config(kind)
    net = wavedm_amd.DiffusionUNet(cfg, dtype=dtype)
    net.load_state_dict(P.procedural_state_dict(cfg, seed=61), strict=False)
    net = net.cuda()
    y = net(seeded(shape, 700).cuda(), torch.tensor([400.0, 30.0]))
    assert tuple(y.shape) == v[kind].shape
    assert rel_linf(y.cpu(), v[kind]) <= TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_sampler_without_other_channels(dtype):
    """use_other=False (ddm_wavelet.py:471-478): the UNet input is [x_cond | x_t]; batched and stitched sampler vs the oracle."""
    from oracle import wavedm_oracle as O
    from wavedm_amd import procedural as P
    cfg, _ = P.variant_config("no_other")
    sd = P.procedural_state_dict(cfg, seed=61)
    d, _ = make_diffusion(cfg, dtype, 6)
    d.model.load_state_dict(sd, strict=True)
    rainy, x_T = P.synthetic_batch(2, patch_px=64)
    xc_cpu = O.dwt_fwd(2 * rainy - 1)
    xs_o, x0_o = O.ddim_batch(sd, cfg, x_T, xc_cpu, None, 6)
    xc = d.wavelet_dec(2 * rainy.cuda() - 1)
    xs, x0 = d.sample_image(xc, x_T.cuda(), x_other=None, last=False, patch_locs=[(0, 0)], patch_size=16, use_other=False)
    assert rel_linf(xs[-1].cpu(), xs_o[-1]) <= TOL[dtype] and rel_linf(x0[-5].cpu(), x0_o[-5]) <= TOL[dtype]
    # patch_locs=None: the reference falls through to utils.sampling.generalized_steps (ddm_wavelet.py:305-306) -- every image one patch, [x_cond | x_t]
    xs_n, x0_n = d.sample_image(xc, x_T.cuda(), last=False)
    assert rel_linf(xs_n[-1].cpu(), xs_o[-1]) <= TOL[dtype] and rel_linf(x0_n[-5].cpu(), x0_o[-5]) <= TOL[dtype]
    assert torch.equal(d.sample_image(xc, x_T.cuda()), xs_n[-1])             # last=True: xs[0][-1] -> the final x
    # stitched: one 30x45 wavelet-domain image, 16x16 patches every 4
    g = torch.Generator().manual_seed(5)
    img = torch.rand(1, 3, 120, 180, generator=g)
    xT = torch.randn(1, 3, 30, 45, generator=g)
    xc_cpu = O.dwt_fwd(2 * img - 1)
    corners = O.grid_corners(30, 45, 16, 4)
    xs_o, _ = O.ddim_overlapping(sd, cfg, xT, xc_cpu, None, corners, 16, 6)
    xs, _ = d.sample_image(d.wavelet_dec(2 * img.cuda() - 1), xT.cuda(), x_other=None, last=False, patch_locs=corners, patch_size=16,
                           use_other=False)
    assert rel_linf(xs[-1].cpu(), xs_o[-1]) <= TOL[dtype]


def test_checkpoint_formats(tmp_path):
    """--resume with the reference's checkpoint dict (utils/logging.py:15-18, ddm_wavelet.py:180-190, :282-292): DDP-prefixed or
    plain state_dict keys, the EMA shadow dict, strict key checking."""
    from types import SimpleNamespace
    import wavedm_amd
    from wavedm_amd import procedural as P
    from oracle import wavedm_oracle as O
    cfg = P.reduced_config()
    cfg.device = torch.device("cuda", 0)
    sd = P.procedural_state_dict(cfg, seed=61)
    ema = {k: v * 1.25 for k, v in sd.items()}
    path = str(tmp_path / "ckpt.pth.tar")
    torch.save({"epoch": 3, "step": 77, "state_dict": {"module." + k: v for k, v in sd.items()}, "optimizer": {}, "ema_helper": ema,
                "params": None, "config": None}, path)
    args = SimpleNamespace(resume=path, sampling_timesteps=5, local_rank=0, image_folder="/tmp/wdm_img", test_set="raindrop", grid_r=4)
    d = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="f32")
    assert (d.start_epoch, d.step) == (3, 77)
    x96 = seeded((1, 96, 16, 16), 40).cuda()
    t = torch.tensor([500.0])
    assert rel_linf(d.model(x96, t).cpu(), O.unet_forward(sd, cfg, x96.cpu(), t)) <= TOL["f32"]
    d.load_ddm_ckpt(path, ema=True)                                                  # EMAHelper.ema: the shadow weights replace the parameters
    assert rel_linf(d.model(x96, t).cpu(), O.unet_forward(ema, cfg, x96.cpu(), t)) <= TOL["f32"]
    bad = dict(sd)
    bad.pop("conv_out.bias")
    torch.save({"epoch": 0, "step": 0, "state_dict": bad}, path)
    with pytest.raises(RuntimeError):
        d.load_ddm_ckpt(path)                                                        # strict=True like the reference


def test_temb_table_gives_the_bits_of_per_step_embedding():
    """wdm_unet_temb_table + wdm_unet_forward_temb (the sampler's default: the timestep-dependent rows of a whole DDIM sequence in four launches) against
    wdm_unet_forward with the timestep itself: same rows, same UNet output, same 6-step trajectory (WAVEDM_TEMB_TABLE=0 switches the sampler back)."""
    import os
    from wavedm_amd import procedural as P
    net = build(P.raindrop_wavelet_config(), "bf16")
    x = seeded((3, 64, 64, 96), 5).cuda().to(torch.bfloat16).contiguous()
    ts = torch.tensor([990.0, 500.0, 10.0], device="cuda")
    tab = net.temb_table(ts, B=3)
    for k in range(3):
        a = net.forward_nhwc(x, ts[k:k + 1], torch.empty(3, 3, 64, 64, device="cuda"))
        b = net.forward_nhwc(x, None, torch.empty(3, 3, 64, 64, device="cuda"), temb_row=tab[k])
        assert torch.equal(a, b) and torch.isfinite(a).all()
    d, _ = make_diffusion(P.reduced_config(), "f32", 6)
    rainy, x_T = P.synthetic_batch(2, patch_px=64)
    got = d.restore_batch(rainy.cuda(), x_T.cuda())
    os.environ["WAVEDM_TEMB_TABLE"] = "0"
    try:
        want = d.restore_batch(rainy.cuda(), x_T.cuda())
    finally:
        del os.environ["WAVEDM_TEMB_TABLE"]
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_bits_do_not_depend_on_the_batch_size():
    """Image 0 of batches of 1, 7, 64, 100 and 128 crops through the full-width UNet: bit-identical in bf16 and f32x3.  (Tile choices that change an output's
    summation order are functions of the layer shape alone; the ones that follow the workgroup count -- 256-column tiles, the persistent form -- write the same
    bits.  Round 3 found one coupling: whether conv1 of a 16 x 16 ResnetBlock could also normalise for conv2 depended on the tile the batch size picked.)"""
    from wavedm_amd import procedural as P
    g = torch.Generator().manual_seed(5)
    x = torch.randn(128, 96, 64, 64, generator=g)
    t = torch.tensor([470.0])
    for dtype in ("bf16", "f16", "f32x3"):
        net = build(P.raindrop_wavelet_config(), dtype)
        ref = net(x[:1].cuda(), t).cpu()
        for B in (7, 64, 100, 128):
            y = net(x[:B].cuda(), t)[:1].cpu()
            assert torch.equal(y, ref), (dtype, B, float((y - ref).abs().max()))
        del net
        torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["bf16", "f32x3"])
def test_sampler_on_several_streams_gives_the_same_bits(dtype):
    """sampling.ddim_sample(streams=...): independent crops walk their trajectories in chunks on separate HIP streams (opt-in: WAVEDM_STREAMS).  Same kernels, per-image
    results independent of the batch an image sits in: the same bits as one stream, for even and ragged chunkings."""
    from wavedm_amd import procedural as P
    from wavedm_amd import sampling
    d, _ = make_diffusion(P.raindrop_wavelet_config(), dtype, 6)
    for nimg in (19, 64):
        rainy, x_T = P.synthetic_batch(nimg, patch_px=256, seed=3)
        rainy, x_T = rainy.cuda(), x_T.cuda()
        outs = []
        for ns in ("1", "4", "3"):
            os.environ["WAVEDM_STREAMS"] = ns
            try:
                outs.append(d.restore_batch(rainy, x_T)[0].clone())
            finally:
                os.environ.pop("WAVEDM_STREAMS", None)
        assert torch.isfinite(outs[0]).all()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (dtype, nimg)


def test_f16_weight_outside_the_fp16_range_is_refused():
    """WDM_F16 stores the packed weights as IEEE half: a value beyond +-65504 (or a NaN) would become inf in the matrix.  wdm_unet_load_param refuses it with WDM_EINVAL
    and names the parameter; the same checkpoint loads in bf16."""
    import wavedm_amd
    from wavedm_amd import procedural as P
    cfg = P.reduced_config()
    sd = P.procedural_state_dict(cfg)
    key = next(k for k in sd if k.endswith("conv1.weight"))
    bad = dict(sd)
    bad[key] = sd[key].clone()
    bad[key].view(-1)[7] = 7.0e4
    net = wavedm_amd.DiffusionUNet(cfg, dtype="f16")
    net.load_state_dict(bad, strict=True)
    with pytest.raises(RuntimeError, match="fp16 range"):
        net.cuda().pack_weights()
    ok = wavedm_amd.DiffusionUNet(cfg, dtype="bf16")
    ok.load_state_dict(bad, strict=True)
    ok.cuda().pack_weights()
    good = wavedm_amd.DiffusionUNet(cfg, dtype="f16")
    good.load_state_dict(sd, strict=True)
    good.cuda().pack_weights()
    assert torch.isfinite(good(seeded((1, 96, 16, 16), 3).cuda(), torch.tensor([10.0]))).all()
    # ... and so is a FOLDED AttnBlock operand (Wk^T Wq) that leaves the range although both factors are inside it
    kq = next(k for k in sd if k.endswith("attn.0.q.weight"))
    kk = kq.replace(".q.", ".k.")
    big = dict(sd)
    c = sd[kq].shape[0]
    big[kq] = (300.0 * torch.eye(c)).reshape(c, c, 1, 1)
    big[kk] = (300.0 * torch.eye(c)).reshape(c, c, 1, 1)
    net = wavedm_amd.DiffusionUNet(cfg, dtype="f16")
    net.load_state_dict(big, strict=True)
    with pytest.raises(RuntimeError, match="fp16 range"):
        net.cuda().pack_weights()


def test_attention_folding_follows_the_tensors_in_whatever_order_they_arrive():
    """16-bit modes run the AttnBlocks on folded operands (Wk^T Wq, Wp Wv: csrc/unet.hip: refold), rebuilt by wdm_unet_load_param from fp32 originals kept beside the
    packed matrices: the result does not depend on the order of the loads, and reloading ONE tensor of a block moves the folded operand with it."""
    import ctypes as C
    import wavedm_amd
    from wavedm_amd import _lib, procedural as P
    cfg = P.raindrop_wavelet_config()
    sd = P.procedural_state_dict(cfg)
    x, t = seeded((2, 96, 64, 64), 5).cuda(), torch.tensor([300.0, 20.0])
    a = build(cfg, "f16")
    ya = a(x, t)
    b = wavedm_amd.DiffusionUNet(cfg, dtype="f16")
    b.load_state_dict(sd, strict=True)
    b = b.cuda()
    b._names = list(reversed(b._names))
    yb = b(x, t)
    assert torch.isfinite(ya).all() and torch.equal(ya, yb)
    # one tensor of one block changes: load it alone into `a`, everything into a fresh model
    sd2 = dict(sd)
    for key in ("down.2.attn.0.k.weight", "up.2.attn.1.proj_out.bias", "up.2.attn.2.v.weight"):
        sd2[key] = sd[key] * 1.25 + 0.01
        src = sd2[key].cuda().contiguous()
        _lib.check(_lib.lib().wdm_unet_load_param(a._u, key.encode(), _lib.ptr(src), src.numel(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    c = wavedm_amd.DiffusionUNet(cfg, dtype="f16")
    c.load_state_dict(sd2, strict=True)
    yc = c.cuda()(x, t)
    a._packed_sig = a._signature()
    ya2 = a(x, t)
    assert torch.equal(ya2, yc) and not torch.equal(ya2, ya)


def test_sampling_loop_replayed_from_a_hipgraph_gives_the_same_bits():
    """WAVEDM_GRAPH=1 (opt-in): the whole DDIM loop captured once and replayed -- independent crops and a stitched patch list, first call (capture) and second call
    (replay with other inputs): bit-identical to direct launches."""
    from wavedm_amd import procedural as P
    from wavedm_amd import sampling
    from oracle import wavedm_oracle as O
    d, _ = make_diffusion(P.reduced_config(), "bf16", 6)
    outs = {}
    for mode in ("0", "1"):
        os.environ["WAVEDM_GRAPH"] = mode
        try:
            res = []
            for seed in (3, 4):                                   # second seed: same shapes, other inputs -> the cached graph is replayed
                rainy, x_T = P.synthetic_batch(5, patch_px=64, seed=seed)
                res.append(d.restore_batch(rainy.cuda(), x_T.cuda()))
                g = torch.Generator().manual_seed(seed)
                img, xT = torch.rand(1, 3, 120, 180, generator=g).cuda(), torch.randn(1, 3, 30, 45, generator=g).cuda()
                xc = d.wavelet_dec(2 * img - 1)
                xs, x0 = d.sample_image(xc, xT, x_other=xc[:, 3:].contiguous(), last=False, patch_locs=O.grid_corners(30, 45, 16, 4), patch_size=16, use_other=True)
                res.append((xs[-1], x0[-5], x0[0]))
            outs[mode] = res
        finally:
            os.environ.pop("WAVEDM_GRAPH", None)
    assert sampling._GRAPHS is not None and len(sampling._GRAPHS) == 2
    for a, b in zip(outs["0"], outs["1"]):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    sampling.graph_cache_clear()


def test_eta_nonzero_matches_the_reference(golden, monkeypatch):
    """eta != 0 in generalized_steps_overlapping (ddm_wavelet.py:500-502): c1 * randn_like(x) enters x_next.  The reference's run with eta = 0.5 (reduced model,
    24x28 image, 12 stitched patches, 6 steps) is the fixture; its per-step draws are fed to the device sampler in place of torch.randn_like."""
    from oracle import wavedm_oracle as O
    from wavedm_amd import procedural as P
    e = golden("eta.npz")
    S, eta = int(e["S"]), float(e["eta"])
    seq = range(0, 1000, 1000 // S)                               # (sample_image's sequence for sampling_timesteps = 6: SEVEN steps, 0, 166, ..., 996)
    nst = len(seq)
    assert nst == len(e["noises"])
    d, args = make_diffusion(P.reduced_config(), "f32", S)
    xc, xT = torch.from_numpy(e["x_cond"]).cuda(), torch.from_numpy(e["x_T"]).cuda()
    draws = [torch.from_numpy(z).cuda() for z in e["noises"]]
    real = torch.randn_like
    calls = []

    def fake(t, *a, **k):
        calls.append(tuple(t.shape))
        return draws[len(calls) - 1].clone()
    monkeypatch.setattr(torch, "randn_like", fake)
    corners = O.grid_corners(24, 28, 16, 4)
    xs, x0 = d.generalized_steps_overlapping(xT, xc, seq, d.model, d.betas, eta=eta, corners=corners, p_size=16,
                                             x_other=xc[:, 3:].contiguous(), use_other=True)
    monkeypatch.setattr(torch, "randn_like", real)
    assert calls == [(1, 3, 24, 28)] * nst                          # one draw of x's shape per step, like the reference
    for name, got in (("xs_last", xs[-1]), ("x0_last", x0[-1]), ("xs_2", xs[2])):
        err = rel_linf(got.cpu(), e[name])
        print(f"eta = {eta} {name}: rel_linf vs the reference {err:.3e}")
        assert err <= 1e-3
    # eta = 0 through the same entry stays the deterministic sampler (no draw)
    calls.clear()
    xs0, _ = d.generalized_steps_overlapping(xT, xc, seq, d.model, d.betas, eta=0., corners=corners, p_size=16,
                                             x_other=xc[:, 3:].contiguous(), use_other=True)
    assert not calls and not torch.equal(xs0[2], xs[2])
    # and against the oracle with draws of the device generator (batched crops: corners=None path of ddim_sample)
    from wavedm_amd import sampling
    torch.manual_seed(7)
    zs = [torch.randn(2, 3, 16, 16, device="cuda") for _ in range(nst)]
    it = iter(zs)
    monkeypatch.setattr(torch, "randn_like", lambda t, *a, **k: next(it).clone())
    xc2, xT2 = seeded((2, 48, 16, 16), 820).cuda(), seeded((2, 3, 16, 16), 821).cuda()
    xs_b, x0_b = sampling.ddim_sample(d.model, xT2, xc2, xc2[:, 3:].contiguous(), list(seq), d.betas, eta=0.3)
    monkeypatch.setattr(torch, "randn_like", real)
    sd = P.procedural_state_dict(P.reduced_config())
    for i in range(2):
        oxs, _ = O.ddim_overlapping(sd, P.reduced_config(), xT2[i:i + 1].cpu(), xc2[i:i + 1].cpu(), xc2[i:i + 1, 3:].cpu(), [(0, 0)], 16, S, eta=0.3,
                                    noises=[z[i:i + 1].cpu() for z in zs])
        assert rel_linf(xs_b[-1][i:i + 1].cpu(), oxs[-1]) <= 1e-3 and rel_linf(xs_b[3][i:i + 1].cpu(), oxs[3]) <= 1e-3
