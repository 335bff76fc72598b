"""CPU: the oracle restatement reproduces the committed golden vectors, which were produced by
running the reference itself (tests/golden/make_golden.py).  No GPU, no /root/reference."""
import numpy as np
import pytest
import torch

from conftest import rel_linf
from oracle import wavedm_oracle as O
from wavedm_amd import procedural as P

torch.set_grad_enabled(False)


def seeded(shape, seed, kind="randn"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn if kind == "randn" else torch.rand)(*shape, generator=g, dtype=torch.float32)


def blk_sd(prefix, shapes):
    return {prefix + "." + k: torch.from_numpy(P.procedural_tensor(prefix + "." + k, s)) for k, s in shapes.items()}


def resblock_shapes(cin, cout):
    s = {"norm1.weight": (cin,), "norm1.bias": (cin,), "conv1.weight": (cout, cin, 3, 3), "conv1.bias": (cout,),
         "temb_proj.weight": (cout, 512), "temb_proj.bias": (cout,), "norm2.weight": (cout,), "norm2.bias": (cout,),
         "conv2.weight": (cout, cout, 3, 3), "conv2.bias": (cout,)}
    if cin != cout:
        s.update({"nin_shortcut.weight": (cout, cin, 1, 1), "nin_shortcut.bias": (cout,)})
    return s


def attn_shapes(c):
    s = {"norm.weight": (c,), "norm.bias": (c,)}
    for p in ("q", "k", "v", "proj_out"):
        s[p + ".weight"] = (c, c, 1, 1)
        s[p + ".bias"] = (c,)
    return s


def conv_shapes(c):
    return {"conv.weight": (c, c, 3, 3), "conv.bias": (c,)}


def test_integer_tables(golden):
    t = golden("tables.npz")
    sign = torch.sign(O.haar_filters()).reshape(16, 16).to(torch.int8).numpy()
    assert np.array_equal(sign, t["rec4_sign"])
    assert np.array_equal(np.abs(O.haar_filters().numpy()), np.full((16, 4, 4), 0.25, np.float32))
    perm = t["subband_perm"]
    assert all(perm[j * 3 + c] == c * 16 + j for j in range(16) for c in range(3))
    for (h, w, p, r) in [(64, 64, 64, 16), (120, 180, 64, 16), (128, 128, 64, 16), (65, 70, 64, 16),
                         (30, 45, 16, 4), (16, 16, 16, 4)]:
        hl, wl = O.overlapping_grid_indices(h, w, p, r)
        assert hl == t[f"grid_h_{h}_{w}_{p}_{r}"].tolist()
        assert wl == t[f"grid_w_{h}_{w}_{p}_{r}"].tolist()
    corners = O.grid_corners(120, 180, 64, 16)
    assert len(corners) == 45
    assert np.array_equal(O.overlap_count_mask(120, 180, 64, corners).numpy(), t["mask_120_180_64_16"].astype(np.int32))
    for S in (10, 25, 50, 100):
        assert O.timestep_seq(1000, S) == t[f"seq_{S}"].tolist()
    betas = O.beta_schedule(P.raindrop_wavelet_config())
    abar = torch.stack([O.compute_alpha(betas, tt) for tt in range(-1, 1000)]).numpy()
    assert np.array_equal(abar, t["alpha_bar_m1_to_999"])
    assert abar[0] == 1.0


def test_dwt_known_answers(golden):
    d = golden("dwt.npz")
    x, y, xr = map(torch.from_numpy, (d["x"], d["y"], d["xr"]))
    assert rel_linf(O.dwt_fwd(x), y) <= 1e-6
    assert rel_linf(O.dwt_inv(y), xr) <= 1e-6
    assert rel_linf(O.dwt_inv(O.dwt_fwd(x)), x) <= 1e-6          # perfect reconstruction
    assert rel_linf(O.dwt_fwd(torch.from_numpy(d["x2"])), torch.from_numpy(d["y2"])) <= 1e-6


def test_blocks(golden):
    b = golden("blocks.npz")
    cases = [("rb_a", resblock_shapes(64, 128), (2, 64, 16, 16), 10, (2, 512), 11),
             ("rb_b", resblock_shapes(128, 128), (2, 128, 8, 8), 12, (1, 512), 13),
             ("rb_c", resblock_shapes(384, 128), (1, 384, 16, 16), 14, (1, 512), 15),
             ("rb_d", resblock_shapes(1280, 768), (1, 1280, 8, 8), 16, (1, 512), 17)]
    for name, shapes, xs, sx, ts, st in cases:
        y = O.resnet_block(blk_sd(name, shapes), name, seeded(xs, sx), seeded(ts, st))
        assert rel_linf(y, b[name]) <= 1e-5, name
    y = O.attn_block(blk_sd("at_a", attn_shapes(512)), "at_a", seeded((1, 512, 16, 16), 20))
    assert rel_linf(y.flatten()[::5], b["at_a_s5"]) <= 1e-5
    y = O.attn_block(blk_sd("at_b", attn_shapes(64)), "at_b", seeded((2, 64, 8, 8), 21))
    assert rel_linf(y, b["at_b"]) <= 1e-5
    y = O.attn_block(blk_sd("at_c", attn_shapes(768)), "at_c", seeded((1, 768, 8, 8), 22))
    assert rel_linf(y, b["at_c"]) <= 1e-5
    y = O.downsample(blk_sd("ds_a", conv_shapes(64)), "ds_a", seeded((2, 64, 16, 16), 30))
    assert rel_linf(y, b["ds_a"]) <= 1e-5
    y = O.upsample(blk_sd("us_a", conv_shapes(64)), "us_a", seeded((2, 64, 8, 8), 31))
    assert rel_linf(y, b["us_a"]) <= 1e-5
    for tt in (0, 10, 990):
        assert rel_linf(O.timestep_embedding(torch.tensor([float(tt)]), 128), b[f"temb_{tt}"]) <= 1e-6


def test_reduced_unet_and_sampler(golden):
    r = golden("reduced.npz")
    cfg = P.reduced_config()
    sd = P.procedural_state_dict(cfg)
    assert sum(v.numel() for v in sd.values()) == 1029667
    x96 = seeded((2, 96, 16, 16), 40)
    assert rel_linf(O.unet_forward(sd, cfg, x96, torch.tensor([500.0])), r["fwd_t500"]) <= 1e-5
    assert rel_linf(O.unet_forward(sd, cfg, x96, torch.tensor([990.0, 10.0])), r["fwd_t_per_image"]) <= 1e-5
    rainy, x_T = P.synthetic_batch(2, patch_px=64)
    xc = O.dwt_fwd(2 * rainy - 1)
    xs, x0 = O.ddim_batch(sd, cfg, x_T, xc, xc[:, 3:], 10)
    assert len(xs) == 11 and len(x0) == 10
    assert rel_linf(xs[-1], r["samp_xs_last"]) <= 1e-5
    assert rel_linf(x0[-5], r["samp_x0_m5"]) <= 1e-5


def test_stitched_restore(golden):
    s = golden("stitch.npz")
    cfg = P.reduced_config()
    sd = P.procedural_state_dict(cfg)
    g = torch.Generator().manual_seed(77)
    img = torch.rand(1, 3, 120, 180, generator=g)
    out, xs, x0 = O.restore(sd, cfg, img, torch.from_numpy(s["x_T"]), 6, r=4)
    assert int(s["n_corners"]) == len(O.grid_corners(30, 45, 16, 4))
    assert out.shape == (1, 3, 120, 180) and float(out.min()) >= 0 and float(out.max()) <= 1
    assert rel_linf(out, s["out"]) <= 1e-5


def test_eta_nonzero_sampler(golden):
    """ddm_wavelet.py:500-502 with eta = 0.5: the oracle fed the reference's own per-step draws reproduces its trajectory."""
    e = golden("eta.npz")
    cfg = P.reduced_config()
    sd = P.procedural_state_dict(cfg)
    xc, xT = torch.from_numpy(e["x_cond"]), torch.from_numpy(e["x_T"])
    xs, x0 = O.ddim_overlapping(sd, cfg, xT, xc, xc[:, 3:], O.grid_corners(24, 28, 16, 4), 16, int(e["S"]), eta=float(e["eta"]),
                                noises=[torch.from_numpy(z) for z in e["noises"]])
    assert rel_linf(xs[-1], e["xs_last"]) <= 1e-5 and rel_linf(x0[-1], e["x0_last"]) <= 1e-5 and rel_linf(xs[2], e["xs_2"]) <= 1e-5
    xs0, _ = O.ddim_overlapping(sd, cfg, xT, xc, xc[:, 3:], O.grid_corners(24, 28, 16, 4), 16, int(e["S"]))
    assert rel_linf(xs0[2], e["xs_2"]) > 1e-3                      # the draws matter


@pytest.mark.parametrize("kind", P.VARIANTS)
def test_optional_unet_branches(golden, kind):
    """use_other_channels False, data.use_window, data.wavelet_in_unet (unet.py:212, :347-350, :387-391)."""
    v = golden("variants.npz")
    cfg, shape = P.variant_config(kind)
    sd = P.procedural_state_dict(cfg, seed=61)
    y = O.unet_forward(sd, cfg, seeded(shape, 700), torch.tensor([400.0, 30.0]))
    assert tuple(y.shape) == v[kind].shape == (2, 3) + tuple(shape[2:])
    assert rel_linf(y, v[kind]) <= 1e-5
    if kind == "wavelet_in_unet":                  # the frozen (de)conv weights the reference keeps in its state_dict
        w = O.haar_filters().repeat(3, 1, 1).unsqueeze(1)
        assert np.array_equal(w.numpy(), v[kind + ":wavelet_dec.conv.weight"]) and np.array_equal(w.numpy(), v[kind + ":wavelet_rec.conv.weight"])


def test_param_layout_full():
    cfg = P.raindrop_wavelet_config()
    shapes = P.unet_param_shapes(cfg)
    assert len(shapes) == 332
    assert sum(int(np.prod(s)) for s in shapes.values()) == 156492675
    assert P.unet_in_channels(cfg) == 96


def test_hfrm_matches_reference_golden(golden):
    """oracle HFRM (models/arch.py restated) == the reference HFRM's outputs, procedural weights."""
    g = golden("hfrm.npz")
    sd = P.procedural_hfrm_state_dict(seed=61)
    assert int(g["n_params"]) == sum(v.numel() for v in sd.values()) == 15941667
    for tag in ("a", "b"):
        x = seeded(tuple(int(v) for v in g["shape_" + tag]), int(g["seed_" + tag]), "rand")
        assert rel_linf(O.hfrm_forward(sd, x), torch.from_numpy(g["y_" + tag])) <= 1e-5
    xb = seeded((2, 64, 16, 24), 93)
    assert rel_linf(O.hfrm_block(sd, "encoders.1.0", xb), torch.from_numpy(g["blk_y"])) <= 1e-5


def test_metrics_match_reference_golden(golden):
    """oracle PSNR restatements == utils/metrics.py run in the build container."""
    g = golden("io.npz")
    gt = seeded((2, 3, 32, 48), 301, "rand")
    out = (gt + 0.1 * seeded((2, 3, 32, 48), 302)).clamp(-0.2, 1.2)
    for k in range(2):
        g1, o1 = gt[k:k + 1], out[k:k + 1]
        assert abs(O.psnr_torch(g1, o1) - g["psnr"][k, 0]) < 1e-4
        assert abs(O.psnr_y(g1, o1) - g["psnr"][k, 1]) < 1e-4
        assert abs(O.psnr_y(g1, o1.clamp(0, 1)) - g["psnr"][k, 2]) < 1e-3     # numpy path: float32 Y on 0..255 data


def _sub(t):
    t = t.detach().flatten()
    return t[:: (1 if t.numel() <= 4096 else 13)]


def test_training_step_matches_reference_golden(golden):
    """loss, every gradient, one Adam step and one EMA update of the oracle == the reference's own
    noise_estimation_loss / autograd / torch.optim.Adam / EMAHelper on the reduced model."""
    g = golden("train.npz")
    cfg = P.reduced_config()
    sd = P.procedural_state_dict(cfg, seed=61)
    betas = O.beta_schedule(cfg)
    x0, e, t = seeded((4, 96, 16, 16), 401), seeded((4, 3, 16, 16), 402), torch.tensor([990, 9, 500, 499])
    loss, out, grads = O.train_grads(sd, cfg, x0, t, e, betas)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert rel_linf(out, torch.from_numpy(g["output"])) <= 1e-5
    names = [str(n) for n in g["grad_names"]]
    assert names == list(grads.keys())
    for k, amax, gsum in zip(names, g["grad_absmax"], g["grad_sum"]):
        assert abs(float(grads[k].abs().max()) - amax) <= 1e-4 * amax + 1e-12, k
    for key in g.files:
        if key.startswith("g:"):
            k = key[2:]
            assert rel_linf(_sub(grads[k]), torch.from_numpy(g[key])) <= 1e-5, k
            p1, m1, v1 = O.adam_step(sd[k], grads[k], torch.zeros_like(sd[k]), torch.zeros_like(sd[k]), 1)
            assert rel_linf(_sub(p1), torch.from_numpy(g["p1:" + k])) <= 1e-6, k
            assert rel_linf(_sub(O.ema_update(sd[k], p1)), torch.from_numpy(g["ema1:" + k])) <= 1e-6, k


def test_training_use_mse_matches_reference_golden(golden):
    """training.use_mse (ddm_wavelet.py:263-264): gradients of mse_loss == the reference's, on the same inputs as above."""
    g = golden("train.npz")
    cfg = P.reduced_config()
    sd = P.procedural_state_dict(cfg, seed=61)
    x0, e, t = seeded((4, 96, 16, 16), 401), seeded((4, 3, 16, 16), 402), torch.tensor([990, 9, 500, 499])
    loss, _, grads = O.train_grads(sd, cfg, x0, t, e, O.beta_schedule(cfg), use_mse=True)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))          # the reported loss is the noise-space one
    for k, amax in zip([str(n) for n in g["grad_names"]], g["gm_absmax"]):
        assert abs(float(grads[k].abs().max()) - amax) <= 1e-4 * amax + 1e-12, k
    for key in g.files:
        if key.startswith("gm:"):
            assert rel_linf(_sub(grads[key[3:]]), torch.from_numpy(g[key])) <= 1e-5, key


def test_global_attn_model_matches_the_reference(golden):
    """DiffusionUNet_Global / Attn_Global: the oracle's restatement against the reference's own outputs, and the parameter
    table against the reference's state_dict order."""
    from wavedm_amd import procedural as P
    g = golden("global.npz")
    cfg = P.global_config()
    sd = P.procedural_global_state_dict(cfg, seed=61)
    assert list(sd.keys()) == [str(k) for k in g["names"]]
    x, xg, t = seeded((2, 6, 16, 16), 710), seeded((2, 3, 32, 32), 711), torch.tensor([400.0, 30.0])
    assert rel_linf(O.unet_global_forward(sd, cfg, x, t, xg), g["y"]) <= 1e-6
    xp, xq = seeded((2, 32, 16, 16), 712), seeded((2, 32, 32, 32), 713)
    assert rel_linf(O.attn_global(sd, "down_global.0.attn", xp, xq), g["attn"]) <= 1e-6
