"""GPU parity of the training-step primitives through the C ABI against torch autograd over the oracle's ops."""
import ctypes as C

import pytest
import torch

from conftest import rel_linf
from gpu_util import DT, dev, scratch, seeded, _p
from oracle import wavedm_oracle as O
from wavedm_amd import _lib

pytestmark = pytest.mark.gpu
BTOL = {"f32": 1e-3, "bf16": 4e-2}


def conv_backward(w, mode, x, dy, dtype, want_dx=True):
    L, h = _lib.lib(), _lib.handle(0)
    wd, xd, dyd = w.to(dev()).contiguous(), x.to(dev()).contiguous(), dy.to(dev()).contiguous()
    B, cin, H, W = xd.shape
    cout = wd.shape[0]
    dx = torch.empty_like(xd) if want_dx else None
    dw = torch.empty_like(wd)
    db = torch.empty(cout, device=dev())
    sc = scratch(1 << 30)
    _lib.check(L.wdm_conv_backward(h, _p(wd), cin, cout, mode, _p(xd), _p(dyd), B, H, W, _p(dx), _p(dw), _p(db), DT[dtype], _p(sc), sc.numel(),
                                   _lib.stream_ptr()))
    torch.cuda.synchronize()
    return (dx.cpu() if want_dx else None), dw.cpu(), db.cpu()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("mode,cin,cout,B,H", [
    (0, 64, 128, 2, 16), (0, 128, 64, 3, 8), (0, 128, 3, 2, 16), (0, 96, 128, 1, 32), (0, 256, 256, 2, 32), (0, 192, 384, 1, 16), (0, 128, 128, 3, 64), (0, 72, 136, 2, 16), (0, 256, 192, 5, 8), (0, 64, 64, 2, 8),
    (1, 64, 64, 2, 16), (2, 64, 64, 2, 8), (3, 64, 128, 2, 16), (3, 384, 128, 1, 16), (3, 160, 64, 3, 8),
])
def test_conv_backward(dtype, mode, cin, cout, B, H):
    k = 1 if mode == 3 else 3
    w = seeded((cout, cin, k, k), 500 + mode) / (cin * k * k) ** 0.5
    b = seeded((cout,), 510 + mode) * 0.1
    x = seeded((B, cin, H, H), 520 + mode + H).requires_grad_(True)
    wl, bl = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    sd = {"c.conv.weight": wl, "c.conv.bias": bl, "c.weight": wl, "c.bias": bl}
    with torch.enable_grad():
        y = [lambda: O.conv(sd, "c", x, padding=1), lambda: O.downsample(sd, "c", x), lambda: O.upsample(sd, "c", x), lambda: O.conv(sd, "c", x)][mode]()
        dy = seeded(tuple(y.shape), 530 + mode)
        y.backward(dy)
    dx, dw, db = conv_backward(w, mode, x.detach(), dy, dtype)
    assert rel_linf(dx, x.grad) <= BTOL[dtype], ("dx", mode, cin, cout)
    assert rel_linf(dw, wl.grad) <= BTOL[dtype], ("dw", mode, cin, cout)
    assert rel_linf(db, bl.grad) <= BTOL[dtype], ("db", mode, cin, cout)


def gn_act_backward(x, C0, gamma, beta, dy, silu, dtype):
    L, h = _lib.lib(), _lib.handle(0)
    xd, dyd, gd, bd = x.to(dev()).contiguous(), dy.to(dev()).contiguous(), gamma.to(dev()).contiguous(), beta.to(dev()).contiguous()
    B, Cc, H, W = xd.shape
    dx, dg, db = torch.empty_like(xd), torch.empty(Cc, device=dev()), torch.empty(Cc, device=dev())
    sc = scratch(1 << 28)
    _lib.check(L.wdm_gn_act_backward(h, _p(xd), C0, Cc, _p(gd), _p(bd), _p(dyd), silu, B, H, W, _p(dx), _p(dg), _p(db), DT[dtype], _p(sc), sc.numel(),
                                     _lib.stream_ptr()))
    torch.cuda.synchronize()
    return dx.cpu(), dg.cpu(), db.cpu()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("C,C0,B,H,silu", [(64, 64, 2, 16, 1), (128, 128, 3, 8, 0), (384, 256, 2, 16, 1), (1280, 768, 1, 8, 1), (32, 32, 2, 16, 1)])
def test_gn_act_backward(dtype, C, C0, B, H, silu):
    x = (seeded((B, C, H, H), 600 + C) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1.0 + 0.1 * seeded((C,), 601)).requires_grad_(True)
    beta = (0.1 * seeded((C,), 602)).requires_grad_(True)
    if dtype == "bf16":            # the device sees bf16 activations: give autograd the same rounded inputs
        x = x.detach().bfloat16().float().requires_grad_(True)
    with torch.enable_grad():
        y = torch.nn.functional.group_norm(x, 32, gamma, beta, eps=1e-6)
        if silu:
            y = y * torch.sigmoid(y)
        dy = seeded(tuple(y.shape), 603)
        if dtype == "bf16":
            dy = dy.bfloat16().float()
        y.backward(dy)
    dx, dg, db = gn_act_backward(x.detach(), C0, gamma.detach(), beta.detach(), dy, silu, dtype)
    tol = 1e-3 if dtype == "f32" else 2e-2
    assert rel_linf(dx, x.grad) <= tol
    assert rel_linf(dg, gamma.grad) <= tol
    assert rel_linf(db, beta.grad) <= tol


def _sub(t):
    t = t.detach().flatten()
    return t[:: (1 if t.numel() <= 4096 else 13)]


def _trainer(dtype, **kw):
    from wavedm_amd import procedural as P
    from wavedm_amd.training import Trainer
    cfg = P.reduced_config()
    cfg.device = dev()
    tr = Trainer(cfg, dtype=dtype, lr=4e-5, eps=1e-8, **kw)
    tr.load_state_dict(P.procedural_state_dict(cfg, seed=61))
    return tr, cfg


def test_trainer_layout_matches_reference_state_dict():
    from wavedm_amd import procedural as P
    tr, cfg = _trainer("f32")
    want = P.unet_param_shapes(cfg)
    assert set(tr.layout) == set(want)
    assert all(tuple(tr.layout[k][1]) == tuple(want[k]) for k in want)
    sd = tr.state_dict()
    ref = P.procedural_state_dict(cfg, seed=61)
    assert all(torch.equal(sd[k].cpu(), ref[k]) for k in ref)


def test_training_step_matches_reference_golden(golden):
    """loss, network output, every gradient, one Adam step and one EMA update == the reference's own training step (f32 mode)."""
    g = golden("train.npz")
    tr, cfg = _trainer("f32")
    x0, e, t = seeded((4, 96, 16, 16), 401).to(dev()), seeded((4, 3, 16, 16), 402).to(dev()), torch.tensor([990, 9, 500, 499])
    loss, out = tr.loss_and_grads(x0, t, e, return_output=True)
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    assert rel_linf(out.cpu(), torch.from_numpy(g["output"])) <= 1e-3
    grads = tr.grad_dict()
    names = [str(n) for n in g["grad_names"]]
    # some gradients are mathematically zero (a bias in front of a one-channel-per-group GroupNorm, the k bias of the attention):
    # both sides hold rounding noise there, so errors are measured against the largest gradient of the model as a floor
    floor = 1e-4 * float(g["grad_absmax"].max())
    worst = 0.0
    for k, amax in zip(names, g["grad_absmax"]):
        got = float(grads[k].abs().max())
        worst = max(worst, abs(got - amax) / max(amax, floor))
    assert worst <= 5e-3, worst
    for key in g.files:
        if key.startswith("g:"):
            k = key[2:]
            want = torch.from_numpy(g[key])
            err = float((_sub(grads[k]).cpu() - want).abs().max()) / max(float(want.abs().max()), floor)
            assert err <= 2e-3, (k, err)
    tr.optimizer_step()
    sd, ema = tr.state_dict(), tr.ema_state_dict()
    for key in g.files:
        if key.startswith("p1:"):
            k = key[3:]
            if float(torch.from_numpy(g["g:" + k]).abs().max()) < floor:
                # a mathematically zero gradient: Adam's first step is lr * sign(rounding noise) on both sides
                assert float((_sub(sd[k]).cpu() - torch.from_numpy(g[key])).abs().max()) <= 2.1 * tr.lr, k
                continue
            assert rel_linf(_sub(sd[k]).cpu(), torch.from_numpy(g[key])) <= 1e-5, k
            assert rel_linf(_sub(ema[k]).cpu(), torch.from_numpy(g["ema1:" + k])) <= 1e-5, k


def test_training_use_mse_matches_reference_golden(golden):
    """training.use_mse: the x0-space objective's gradients == the reference's mse_loss.backward() (f32 mode); the reported loss stays the
    noise-space value."""
    g = golden("train.npz")
    tr, cfg = _trainer("f32", use_mse=True)
    x0, e, t = seeded((4, 96, 16, 16), 401).to(dev()), seeded((4, 3, 16, 16), 402).to(dev()), torch.tensor([990, 9, 500, 499])
    loss = tr.loss_and_grads(x0, t, e)
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    grads = tr.grad_dict()
    floor = 1e-4 * float(g["gm_absmax"].max())
    for k, amax in zip([str(n) for n in g["grad_names"]], g["gm_absmax"]):
        assert abs(float(grads[k].abs().max()) - amax) / max(amax, floor) <= 5e-3, k
    for key in g.files:
        if key.startswith("gm:"):
            want = torch.from_numpy(g[key])
            err = float((_sub(grads[key[3:]]).cpu() - want).abs().max()) / max(float(want.abs().max()), floor)
            assert err <= 2e-3, (key, err)
    assert float(g["gm_absmax"].max()) > 3 * float(g["grad_absmax"].max())          # the two objectives really differ on these inputs


def test_training_step_bf16_tracks_f32():
    trf, _ = _trainer("f32")
    trb, _ = _trainer("bf16")
    x0, e, t = seeded((4, 96, 16, 16), 401).to(dev()), seeded((4, 3, 16, 16), 402).to(dev()), torch.tensor([990, 9, 500, 499])
    lf, lb = float(trf.loss_and_grads(x0, t, e)), float(trb.loss_and_grads(x0, t, e))
    assert abs(lf - lb) <= 2e-2 * abs(lf)
    gf, gb = trf.grads, trb.grads
    cos = float((gf * gb).sum() / (gf.norm() * gb.norm()))
    assert cos >= 0.98, cos


def test_training_step_full_width_f32():
    """Full raindrop_wavelet UNet (156 M parameters, attention at 16x16), 2 samples: loss and a spread of gradients vs torch autograd over
    the oracle on the host."""
    from wavedm_amd import procedural as P
    from wavedm_amd.training import Trainer
    cfg = P.raindrop_wavelet_config()
    cfg.device = dev()
    sd = P.procedural_state_dict(cfg, seed=61)
    tr = Trainer(cfg, dtype="f32")
    tr.load_state_dict(sd)
    x0, e, t = seeded((2, 96, 64, 64), 411), seeded((2, 3, 64, 64), 412), torch.tensor([700, 120])
    loss = float(tr.loss_and_grads(x0.to(dev()), t, e.to(dev())))
    ol, _, og = O.train_grads(sd, cfg, x0, t, e, O.beta_schedule(cfg))
    assert abs(loss - float(ol)) <= 1e-4 * abs(float(ol))
    g = tr.grad_dict()
    floor = 1e-4 * max(float(v.abs().max()) for v in og.values())
    for k in ["conv_in.weight", "down.0.block.1.conv2.weight", "down.1.downsample.conv.weight", "down.2.attn.0.q.weight", "down.2.attn.1.v.bias",
              "down.3.block.0.nin_shortcut.weight", "mid.attn_1.proj_out.weight", "mid.block_2.norm1.weight", "up.3.block.0.conv1.weight",
              "up.2.block.2.nin_shortcut.weight", "up.2.attn.1.k.weight", "up.1.upsample.conv.weight", "up.0.block.2.temb_proj.weight",
              "temb.dense.0.weight", "temb.dense.1.bias", "norm_out.bias", "conv_out.weight"]:
        err = float((g[k].cpu() - og[k]).abs().max()) / max(float(og[k].abs().max()), floor)
        assert err <= 2e-3, (k, err)


def test_direct_weight_gradient_kernel_on_the_full_model_bf16():
    """conv_wgrad_kernel.h (bf16: NHWC operands in LDS, transposing reads, nine taps in registers) against the batched-GEMM form of the weight gradient
    (WDM_WGRAD_BG=2 keeps every layer on it) on the full-width model -- channel-concat inputs, the 8 x 8 maps, the upsample convs: the same bf16 operands and
    fp32 accumulation in another order, so every weight gradient agrees to ~1e-4 of its scale; loss and all other gradients too."""
    import os
    from wavedm_amd import _lib, procedural as P
    from wavedm_amd.training import Trainer
    cfg = P.raindrop_wavelet_config()
    cfg.device = dev()
    sd = P.procedural_state_dict(cfg, seed=61)
    x0, e, t = seeded((2, 96, 64, 64), 411).to(dev()), seeded((2, 3, 64, 64), 412).to(dev()), torch.tensor([700, 120])

    def run():
        tr = Trainer(cfg, dtype="bf16")
        tr.load_state_dict(sd)
        loss = float(tr.loss_and_grads(x0, t, e))
        return loss, {k: v.detach().float().cpu().clone() for k, v in tr.grad_dict().items()}
    old = os.environ.get("WDM_WGRAD_BG")
    try:
        os.environ.pop("WDM_WGRAD_BG", None)
        _lib.env_refresh()
        l1, g1 = run()
        os.environ["WDM_WGRAD_BG"] = "2"
        _lib.env_refresh()
        l0, g0 = run()
    finally:
        if old is None:
            os.environ.pop("WDM_WGRAD_BG", None)
        else:
            os.environ["WDM_WGRAD_BG"] = old
        _lib.env_refresh()
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    worst = 0.0
    for k in g0:
        scale = max(float(g0[k].abs().max()), 1e-12)
        err = float((g1[k] - g0[k]).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 2e-3, (k, err)
    print(f"direct vs GEMM-form weight gradient, worst relative difference over {len(g0)} tensors: {worst:.2e}")


def test_training_gradients_are_deterministic_bf16():
    """No atomics anywhere in the backward (the direct weight-gradient kernel reduces its split-K partials in a fixed order): two runs of the same full-width
    step give the same loss and the same gradient bits."""
    from wavedm_amd import procedural as P
    from wavedm_amd.training import Trainer
    cfg = P.raindrop_wavelet_config()
    cfg.device = dev()
    sd = P.procedural_state_dict(cfg, seed=61)
    x0, e, t = seeded((3, 96, 64, 64), 421).to(dev()), seeded((3, 3, 64, 64), 422).to(dev()), torch.tensor([900, 40, 511])
    tr = Trainer(cfg, dtype="bf16")
    tr.load_state_dict(sd)
    l0 = float(tr.loss_and_grads(x0, t, e))
    g0 = tr.grads.clone()
    for _ in range(2):
        l1 = float(tr.loss_and_grads(x0, t, e))
        assert l1 == l0
        assert torch.equal(tr.grads, g0)


def test_train_step_api_and_checkpoint_roundtrip(tmp_path):
    """DenoisingDiffusion_Wavelet.train_step on raw crops: the loss goes down over a few steps on a fixed batch, and the checkpoint the
    trainer writes loads back through --resume (reference dict format) with the EMA weights available."""
    from types import SimpleNamespace
    import wavedm_amd
    from wavedm_amd import procedural as P
    cfg = P.reduced_config()
    cfg.device = dev()
    cfg.optim = SimpleNamespace(lr=2e-3, eps=1e-8, weight_decay=0.0)
    args = SimpleNamespace(resume="", sampling_timesteps=5, local_rank=0, image_folder="/tmp/wdm_img", test_set="raindrop", grid_r=4)
    d = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="f32")
    d.model.load_state_dict(P.procedural_state_dict(cfg), strict=True)
    tr = d.make_trainer(dtype="f32")
    g = torch.Generator().manual_seed(3)
    x = torch.rand(4, 6, 64, 64, generator=g)
    x0 = d.assemble_training_sample(x)
    assert tuple(x0.shape) == (4, 96, 16, 16)
    want = O.dwt_fwd(2 * x[:, 3:] - 1)
    assert rel_linf(x0[:, 48:51].cpu(), want[:, :3]) <= 1e-5 and rel_linf(x0[:, 51:].cpu(), want[:, 3:]) <= 1e-5
    e, t = seeded((4, 3, 16, 16), 5).to(dev()), torch.tensor([800, 300, 50, 600])
    first = float(tr.loss_and_grads(x0, t, e))
    for _ in range(25):
        tr.loss_and_grads(x0, t, e)
        tr.optimizer_step()
    last = float(tr.loss_and_grads(x0, t, e))
    assert last < 0.7 * first, (first, last)                  # Adam on a fixed batch: the loss must fall
    path = str(tmp_path / "ck.pth.tar")
    tr.save_checkpoint(path, epoch=2)
    args2 = SimpleNamespace(resume=path, sampling_timesteps=5, local_rank=0, image_folder="/tmp/wdm_img", test_set="raindrop", grid_r=4)
    d2 = wavedm_amd.DenoisingDiffusion_Wavelet(args2, cfg, generator=lambda x: x, dtype="f32")
    assert d2.step == 25 and d2.start_epoch == 2
    sd = tr.state_dict()
    assert all(torch.equal(p.detach().cpu(), sd[k].cpu()) for k, p in d2.model.named_parameters())
    d2.load_ddm_ckpt(path, ema=True)
    ema = tr.ema_state_dict()
    assert all(torch.equal(p.detach().cpu(), ema[k].cpu()) for k, p in d2.model.named_parameters())


def test_resume_restores_adam_state(tmp_path):
    """save -> --resume -> step equals the uninterrupted run (weights, EMA shadow, Adam moments, step count), and the
    'optimizer' entry of the checkpoint is a torch.optim.Adam state_dict in the reference's parameter order (ddm_wavelet.py:186, :288)."""
    from types import SimpleNamespace
    import wavedm_amd
    from wavedm_amd import procedural as P
    cfg = P.reduced_config()
    cfg.device = dev()
    cfg.optim = SimpleNamespace(lr=1e-3, eps=1e-8, weight_decay=0.0)
    args = SimpleNamespace(resume="", sampling_timesteps=5, local_rank=0, image_folder="/tmp/wdm_img", test_set="raindrop", grid_r=4)
    sd0 = P.procedural_state_dict(cfg, seed=61)
    x0 = seeded((4, 96, 16, 16), 21).to(dev())
    e, t = seeded((4, 3, 16, 16), 22).to(dev()), torch.tensor([900, 40, 510, 333])

    def steps(tr, n):
        for _ in range(n):
            tr.loss_and_grads(x0, t, e)
            tr.optimizer_step()

    da = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="f32")
    da.model.load_state_dict(sd0, strict=True)
    ta = da.make_trainer(dtype="f32")
    steps(ta, 12)                                               # the uninterrupted run
    db = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="f32")
    db.model.load_state_dict(sd0, strict=True)
    tb = db.make_trainer(dtype="f32")
    steps(tb, 10)
    path = str(tmp_path / "resume.pth.tar")
    tb.save_checkpoint(path, epoch=1)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    names = [k for k, _ in da.model.named_parameters()]
    assert ck["optimizer"]["param_groups"][0]["params"] == list(range(len(names)))
    assert tuple(ck["optimizer"]["state"][3]["exp_avg"].shape) == tuple(dict(da.model.named_parameters())[names[3]].shape)
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros_like(p)) for _, p in da.model.named_parameters()], lr=1.0)
    opt.load_state_dict(ck["optimizer"])                        # torch accepts it: the layout is torch.optim.Adam's own
    assert float(opt.state_dict()["state"][0]["step"]) == 10 and opt.param_groups[0]["lr"] == 1e-3
    args_r = SimpleNamespace(resume=path, sampling_timesteps=5, local_rank=0, image_folder="/tmp/wdm_img", test_set="raindrop", grid_r=4)
    dc = wavedm_amd.DenoisingDiffusion_Wavelet(args_r, cfg, generator=lambda x: x, dtype="f32")
    tc = dc.make_trainer(dtype="f32")
    assert tc.step == 10 and torch.equal(tc.exp_avg, tb.exp_avg) and torch.equal(tc.exp_avg_sq, tb.exp_avg_sq) and torch.equal(tc.ema, tb.ema)
    steps(tc, 2)
    assert tc.step == 12
    assert torch.equal(tc.params, ta.params) and torch.equal(tc.ema, ta.ema)        # same kernels, same state: bit-identical


def test_train_loop_on_synthetic_dataset(tmp_path):
    """DenoisingDiffusion_Wavelet.train(DATASET): RainDrop training loader (random crops) -> DWT -> train steps -> checkpoint at step 1."""
    import os
    import random
    from types import SimpleNamespace
    import wavedm_amd
    from wavedm_amd import procedural as P
    from wavedm_amd.datasets import RainDrop
    O.synthetic_raindrop_dir(str(tmp_path), seed=303, sizes=((200, 140), (180, 120)))
    os.rename(tmp_path / "raindrop" / "raindrop_test", tmp_path / "raindrop" / "train")
    for sub in ("input", "gt"):
        os.makedirs(tmp_path / "raindrop" / "raindrop_test" / sub)
    cfg = P.reduced_config()
    cfg.device = dev()
    cfg.data.data_dir, cfg.data.patch_size = str(tmp_path), 64
    cfg.training = SimpleNamespace(patch_n=4, batch_size=1, n_epochs=2, snapshot_freq=1000)
    cfg.optim = SimpleNamespace(lr=1e-3, eps=1e-8, weight_decay=0.0)
    args = SimpleNamespace(resume="", sampling_timesteps=5, local_rank=0, image_folder="/tmp/wdm_img", test_set="raindrop", grid_r=4, world_size=1, rank=0)
    d = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="bf16")
    d.model.load_state_dict(P.procedural_state_dict(cfg), strict=True)
    random.seed(1)
    torch.manual_seed(1)
    d.train(RainDrop(args, cfg), max_steps=3)
    assert d.step == 3
    ck = tmp_path / "ckpts" / f"{cfg.data.dataset}_epoch1_ddpm.pth.tar"
    assert ck.is_file()                                        # written at step 1 (ddm_wavelet.py:282)
    saved = torch.load(ck, weights_only=False)
    assert saved["step"] == 1 and set(saved["state_dict"]) == set(P.unet_param_shapes(cfg))
    d.sync_from_trainer(ema=True)
    x96 = seeded((1, 96, 16, 16), 40).to(dev())
    assert bool(torch.isfinite(d.model(x96, torch.tensor([500.0]))).all())


def test_train_loop_validation_sheet(tmp_path):
    """The training loop's periodic validation restore (ddm_wavelet.py:273-278, :340-411): at step 10 (single process) rank 0 restores
    the first two validation images and writes one 4-column sheet [input | LL from the sampler + gt bands | output | gt]."""
    import os
    import random
    from types import SimpleNamespace
    import numpy as np
    from PIL import Image
    import wavedm_amd
    from wavedm_amd import procedural as P
    from wavedm_amd.datasets import RainDrop
    O.synthetic_raindrop_dir(str(tmp_path), seed=303, sizes=((200, 140), (200, 140), (200, 140)))
    import shutil
    shutil.copytree(tmp_path / "raindrop" / "raindrop_test", tmp_path / "raindrop" / "train")
    cfg = P.reduced_config()
    cfg.device = dev()
    cfg.data.data_dir, cfg.data.patch_size = str(tmp_path), 64
    cfg.training = SimpleNamespace(patch_n=2, batch_size=1, n_epochs=8, snapshot_freq=1000, validation_freq=1000)
    cfg.optim = SimpleNamespace(lr=1e-4, eps=1e-8, weight_decay=0.0)
    args = SimpleNamespace(resume="", sampling_timesteps=5, local_rank=0, image_folder=str(tmp_path / "img"), test_set="raindrop", grid_r=4,
                           world_size=1, rank=0)
    d = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="bf16")
    d.model.load_state_dict(P.procedural_state_dict(cfg), strict=True)
    random.seed(1)
    torch.manual_seed(1)
    d.train(RainDrop(args, cfg), max_steps=10)
    assert d.step == 10
    folder = tmp_path / "img" / cfg.data.dataset / "raindrop"
    sheets = sorted(os.listdir(folder))
    assert len(sheets) == 1 and sheets[0].endswith("_output_epoch3.png")          # 3 training items per epoch: step 10 is in epoch 3
    sheet = np.asarray(Image.open(folder / sheets[0]))
    _, val_loader = RainDrop(args, cfg).get_loaders(parse_patches=False, validation="raindrop")
    items = list(val_loader)
    H, W = items[0][0].shape[-2:]
    assert sheet.shape == (2 * (H + 2) + 2, 4 * (W + 2) + 2, 3)
    u8 = lambda t: (t[0].mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8)).numpy()
    seen = []
    for row in range(2):
        y0 = row * (H + 2) + 2
        col = lambda c: sheet[y0:y0 + H, c * (W + 2) + 2:c * (W + 2) + 2 + W]
        match = [k for k, (x, y, total) in enumerate(items) if np.array_equal(col(0), u8(x[:, :3]))]   # column 0: a degraded input ...
        assert len(match) == 1 and match[0] not in seen
        seen.append(match[0])
        assert np.array_equal(col(3), u8(items[match[0]][0][:, 3:]))                                     # ... column 3: its ground truth
        assert col(1).std() > 0 and col(2).std() > 0
    assert sheets[0].startswith(str(items[seen[-1]][1]))                                                 # named after the last item restored
    assert not sheet[:2].any() and not sheet[:, :2].any()                                              # frame = pad_value 0
