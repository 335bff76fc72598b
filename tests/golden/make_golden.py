#!/usr/bin/env python3
"""Generate the committed golden fixtures by RUNNING THE REFERENCE ITSELF (build container only).

    python tests/golden/make_golden.py            # needs /root/reference; writes tests/golden/*.npz

The reference (pure Python, /root/reference) has no tests or golden vectors of its own
(SURVEY.md §4), so parity is pinned by importing it here with three shims:
stub `torchvision` / `cv2` / `skimage` modules, cwd = /root/reference (relative pickle path,
wavelet.py:7), and an un-constructed `DenoisingDiffusion_Wavelet` instance (its ctor needs an
absent HFRM checkpoint + NCCL + CUDA DDP) whose attributes are set by hand so the *bound
reference methods* `sample_image` / `generalized_steps_overlapping` /
`overlapping_grid_indices` and `DiffusiveRestoration.restore` run unmodified on CPU.

Every fixture holds DATA only (inputs are regenerated from seeds; outputs are stored, large ones
sub-sampled with a fixed stride).  While writing them the script also asserts that
`oracle/wavedm_oracle.py` reproduces the reference on each case (<= 1e-5 max-norm relative for
floats, exact for integers); that is what "pins" the oracle.
"""
import argparse
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from wavedm_amd import procedural as P           # noqa: E402
from oracle import wavedm_oracle as O            # noqa: E402

SAVED_IMAGES = {}


def install_stubs():
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")

    def save_image(img, path, normalize=False, **kw):
        SAVED_IMAGES[os.path.basename(path)] = img.detach().clone()
    tvu.save_image = save_image
    tvu.make_grid = lambda x, **kw: x
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvf.crop = lambda img, t, l, h, w: img[..., t:t + h, l:l + w]
    tvt.functional = tvf
    tv.utils, tv.transforms = tvu, tvt
    tvm = types.ModuleType("torchvision.models")
    tv.models = tvm
    sys.modules.update({"torchvision": tv, "torchvision.utils": tvu, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf, "torchvision.models": tvm})
    cv2 = types.ModuleType("cv2")
    sk = types.ModuleType("skimage")
    skc = types.ModuleType("skimage.color")
    sk.color = skc
    sys.modules.update({"cv2": cv2, "skimage": sk, "skimage.color": skc})


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def check(name, got, want, tol=1e-5):
    e = rel_err(got, want)
    print(f"  oracle vs reference  {name:<28s} rel_linf = {e:.3e}")
    assert e <= tol, (name, e)


def sub(t, stride):
    """Fixed-stride subsample of a flattened tensor (keeps fixtures small)."""
    return t.detach().flatten()[::stride].contiguous().numpy()


def seeded(shape, seed, kind="randn"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn if kind == "randn" else torch.rand)(*shape, generator=g, dtype=torch.float32)


def block_sd(prefix, shapes, seed=61):
    return {k: torch.from_numpy(P.procedural_tensor(k, s, seed)) for k, s in shapes.items()
            if k.startswith(prefix)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true", help="skip the full-width (156 M) cases")
    ap.add_argument("--only-hfrm", action="store_true", help="regenerate hfrm.npz only")
    ap.add_argument("--only-io", action="store_true", help="regenerate io.npz only")
    ap.add_argument("--only-train", action="store_true", help="regenerate train.npz only")
    ap.add_argument("--only-variants", action="store_true", help="regenerate variants.npz only")
    ap.add_argument("--only-config", action="store_true", help="regenerate config.npz only")
    ap.add_argument("--only-global", action="store_true", help="regenerate global.npz only")
    ap.add_argument("--only-eta", action="store_true", help="regenerate eta.npz only")
    ap.add_argument("--only-timing", action="store_true", help="time the reference against the oracle on BASELINE configs[0] only (oracle_timing.json)")
    args = ap.parse_args()

    install_stubs()
    os.chdir(REF)
    sys.path.insert(0, REF)
    import models                                              # noqa: F401  (reference package)
    from models import unet as RU
    from models.wavelet import WaveletTransform
    from models.ddm_wavelet import DenoisingDiffusion_Wavelet, get_beta_schedule
    from models.restoration import DiffusiveRestoration
    from utils.sampling import compute_alpha
    torch.set_grad_enabled(False)

    out = lambda n: os.path.join(HERE, n)

    # ------------------------------------------------------------------ HFRM (models/arch.py)
    def golden_hfrm():
        print("[hfrm]")
        from models.arch import HFRM
        sd_h = P.procedural_hfrm_state_dict(seed=61)
        gen = HFRM(in_channel=3, dim=32, mid_blk_num=6, enc_blk_nums=[2, 2, 2, 4], dec_blk_nums=[2, 2, 2, 2]).eval()
        assert list(gen.state_dict().keys()) == list(sd_h.keys())
        gen.load_state_dict(sd_h, strict=True)
        hf = {"n_params": np.array(sum(v.numel() for v in sd_h.values()), dtype=np.int64)}
        for tag, shape, seed in (("a", (2, 3, 32, 48), 91), ("b", (1, 3, 64, 64), 92)):
            x = seeded(shape, seed, "rand")
            y = gen(x)
            check(f"hfrm forward {tag}", O.hfrm_forward(sd_h, x), y)
            hf["y_" + tag] = y.numpy()
            hf["shape_" + tag] = np.array(shape, dtype=np.int32)
            hf["seed_" + tag] = np.array(seed, dtype=np.int32)
        # one block in isolation (NAFBlock-style, arch.py:132-204) at d = 64
        blk = gen.encoders[1][0]
        xb = seeded((2, 64, 16, 24), 93)
        yb = blk(xb)
        check("hfrm block", O.hfrm_block(sd_h, "encoders.1.0", xb), yb)
        hf["blk_y"] = yb.numpy()
        np.savez_compressed(out("hfrm.npz"), **hf)

    # ------------------------------------------------------------------ metrics + data loading
    def golden_io():
        print("[io]")
        import random
        import tempfile
        import utils as RUT                                        # utils/metrics.py (cv2 / skimage stubbed: unused here)
        from PIL import Image
        io = {}
        gt = seeded((2, 3, 32, 48), 301, "rand")
        outp = (gt + 0.1 * seeded((2, 3, 32, 48), 302)).clamp(-0.2, 1.2)      # a little outside [0,1]: torchPSNR clamps
        ref = []
        for k in range(2):
            g1, o1 = gt[k:k + 1], outp[k:k + 1]
            to255 = lambda t: torch.clamp(t[0] * 255, 0, 255).numpy().transpose((1, 2, 0))
            ref.append([float(RUT.torchPSNR(g1, o1)), float(RUT.calculate_psnr_in_GPU(g1, o1, True)),
                        float(RUT.calculate_psnr(to255(g1), to255(o1.clamp(0, 1)), True))])
            assert abs(O.psnr_torch(g1, o1) - ref[-1][0]) < 1e-4 and abs(O.psnr_y(g1, o1) - ref[-1][1]) < 1e-4
        io["psnr"] = np.array(ref, dtype=np.float64)               # rows: image; cols: torchPSNR, GPU-Y, numpy-Y
        # eval-path items of the reference dataset class on two synthetic PNG pairs
        from datasets.raindrop import RainDropDataset
        tmp = tempfile.mkdtemp(prefix="wdm_golden_ds")
        sizes = O.synthetic_raindrop_dir(tmp, seed=303)
        random.seed(61)
        ds = RainDropDataset(dir=os.path.join(tmp, "raindrop", "raindrop_test"), patch_size=256, n=8, transforms=O.pil_to_tensor,
                             filelist=None, parse_patches=False)
        for i in range(len(ds)):
            x, img_id, total = ds[i]
            io[f"ds_{img_id}_shape"] = np.array(x.shape, dtype=np.int32)
            io[f"ds_{img_id}_sub"] = sub(x, 997)
            io[f"ds_{img_id}_sum"] = np.array(float(x.double().sum()))
            assert torch.equal(total, x[:3])
        io["ds_order"] = np.array([re_id for re_id in [ds[i][1] for i in range(len(ds))]])
        io["ds_sizes"] = np.array(sizes, dtype=np.int32)
        np.savez_compressed(out("io.npz"), **io)

    # ------------------------------------------------------------------ training step
    def golden_train():
        print("[train]")
        from models.ddm_wavelet import noise_estimation_loss, EMAHelper
        torch.set_grad_enabled(True)
        cfg_t = P.reduced_config()
        sd_t = P.procedural_state_dict(cfg_t, seed=61)
        net_t = RU.DiffusionUNet(cfg_t).train()
        net_t.load_state_dict(sd_t, strict=True)
        betas_t = torch.from_numpy(get_beta_schedule(beta_schedule="linear", beta_start=0.0001, beta_end=0.02, num_diffusion_timesteps=1000)).float()
        x0 = seeded((4, 96, 16, 16), 401)
        e = seeded((4, 3, 16, 16), 402)
        t = torch.tensor([990, 9, 500, 499])
        ema = EMAHelper()
        ema.register(net_t)
        opt = torch.optim.Adam(net_t.parameters(), lr=0.00004, weight_decay=0.0, betas=(0.9, 0.999), amsgrad=False, eps=0.00000001)
        loss, output, x0_pred, mse = noise_estimation_loss(net_t, x0, t, e, betas_t, inp_channels=48, pred_channels=3, use_other_channels=True)
        opt.zero_grad()
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in net_t.named_parameters()}
        o_loss, o_out, o_g = O.train_grads(sd_t, cfg_t, x0, t, e, betas_t)
        check("train loss", o_loss.reshape(1), loss.detach().reshape(1))
        check("train output", o_out, output.detach())
        worst = max(rel_err(o_g[k], grads[k]) for k in grads)
        print(f"  oracle vs reference  all {len(grads)} gradients: worst rel_linf = {worst:.3e}")
        assert worst <= 1e-4
        opt.step()
        ema.update(net_t)
        tr = {"loss": np.array(float(loss)), "mse": np.array(float(mse)), "output": output.detach().numpy(), "x0_pred": sub(x0_pred, 3),
              "grad_names": np.array(list(grads.keys())),
              "grad_absmax": np.array([float(g.abs().max()) for g in grads.values()]),
              "grad_sum": np.array([float(g.double().sum()) for g in grads.values()])}
        keep = ["conv_in.weight", "conv_out.bias", "temb.dense.0.weight", "down.0.block.0.conv1.weight", "down.0.block.0.norm1.weight",
                "down.1.block.0.nin_shortcut.weight", "down.1.attn.0.q.weight", "down.1.attn.0.proj_out.bias", "mid.block_1.temb_proj.weight",
                "mid.attn_1.k.bias", "up.0.block.2.conv2.weight", "up.1.block.0.norm2.bias", "up.1.upsample.conv.weight", "down.0.downsample.conv.weight"]
        newp = dict(net_t.named_parameters())
        for k in keep:
            st = 1 if grads[k].numel() <= 4096 else 13            # fixed subsampling rule (the tests apply the same)
            tr["g:" + k] = sub(grads[k], st)
            tr["p1:" + k] = sub(newp[k], st)
            tr["ema1:" + k] = sub(ema.shadow[k], st)
            pn, _, _ = O.adam_step(sd_t[k], grads[k], torch.zeros_like(sd_t[k]), torch.zeros_like(sd_t[k]), 1)
            assert rel_err(pn, newp[k].detach()) <= 1e-6, k
            assert rel_err(O.ema_update(sd_t[k], pn), ema.shadow[k]) <= 1e-6, k
        # training.use_mse: the same step differentiating mse_loss (ddm_wavelet.py:263-264) on a fresh copy of the model
        net_m = RU.DiffusionUNet(cfg_t).train()
        net_m.load_state_dict(sd_t, strict=True)
        _, _, _, mse_m = noise_estimation_loss(net_m, x0, t, e, betas_t, inp_channels=48, pred_channels=3, use_other_channels=True)
        mse_m.backward()
        grads_m = {k: p.grad.detach().clone() for k, p in net_m.named_parameters()}
        _, _, o_gm = O.train_grads(sd_t, cfg_t, x0, t, e, betas_t, use_mse=True)
        worst = max(rel_err(o_gm[k], grads_m[k]) for k in grads_m)
        print(f"  oracle vs reference  use_mse: all {len(grads_m)} gradients: worst rel_linf = {worst:.3e}")
        assert worst <= 1e-4
        tr["gm_absmax"] = np.array([float(g.abs().max()) for g in grads_m.values()])
        tr["gm_sum"] = np.array([float(g.double().sum()) for g in grads_m.values()])
        for k in keep:
            tr["gm:" + k] = sub(grads_m[k], 1 if grads_m[k].numel() <= 4096 else 13)
        np.savez_compressed(out("train.npz"), **tr)
        torch.set_grad_enabled(False)

    # ------------------------------------------------------------------ optional config branches
    def golden_variants():
        print("[variants]")
        va = {}
        for kind in P.VARIANTS:
            c, shape = P.variant_config(kind)
            c.device = torch.device("cpu")
            sd_v = P.procedural_state_dict(c, seed=61)
            net_v = RU.DiffusionUNet(c).eval()
            # wavelet_in_unet: the model also holds its two frozen Haar (de)conv weights (models/unet.py:204-206); they stay as built
            frozen = [k for k in net_v.state_dict().keys() if k.startswith("wavelet_")]
            assert [k for k in net_v.state_dict().keys() if k not in frozen] == list(sd_v.keys()), kind
            assert net_v.load_state_dict(sd_v, strict=False).unexpected_keys == []
            for k in frozen:
                va[kind + ":" + k] = net_v.state_dict()[k].numpy()
            x = seeded(shape, 700)
            y = net_v(x, torch.tensor([400.0, 30.0]))
            check(f"unet variant {kind}", O.unet_forward(sd_v, c, x, torch.tensor([400.0, 30.0])), y)
            va[kind] = y.numpy()
        np.savez_compressed(out("variants.npz"), **va)

    # ------------------------------------------------------------------ data.global_attn: DiffusionUNet_Global (models/unet.py:397-636)
    def golden_global():
        print("[global]")
        c = P.global_config()
        c.device = torch.device("cpu")
        sd_g = P.procedural_global_state_dict(c, seed=61)
        net_g = RU.DiffusionUNet_Global(c).eval()
        assert list(net_g.state_dict().keys()) == list(sd_g.keys()), "parameter order / names of DiffusionUNet_Global"
        assert [tuple(v.shape) for v in net_g.state_dict().values()] == [tuple(v.shape) for v in sd_g.values()]
        net_g.load_state_dict(sd_g, strict=True)
        x, xg, t = seeded((2, 6, 16, 16), 710), seeded((2, 3, 32, 32), 711), torch.tensor([400.0, 30.0])
        y = net_g(x, t, xg)
        check("unet global_attn", O.unet_global_forward(sd_g, c, x, t, xg), y)
        ga = {"y": y.numpy(), "names": np.array(list(sd_g.keys()))}
        # one Attn_Global on its own (32 channels, 16x16 patch map -> 64 queries, 32x32 whole-image map -> 16 key / value tokens)
        ag = RU.Attn_Global(32).eval()
        sd_a = {k[len("down_global.0.attn."):]: v for k, v in sd_g.items() if k.startswith("down_global.0.attn.")}
        ag.load_state_dict(sd_a, strict=True)
        xp, xq = seeded((2, 32, 16, 16), 712), seeded((2, 32, 32, 32), 713)
        ya = ag(xp, xq)
        check("Attn_Global", O.attn_global(sd_g, "down_global.0.attn", xp, xq), ya)
        ga["attn"] = ya.numpy()
        np.savez_compressed(out("global.npz"), **ga)

    # ------------------------------------------------------------------ config file (the YAML keys the drop-in must read)
    def golden_config():
        print("[config]")
        import hashlib
        import json
        import yaml
        from wavedm_amd.config import namespace2dict
        with open(os.path.join(REF, "configs", "raindrop_wavelet.yml")) as f:
            ref_cfg = yaml.safe_load(f)
        mine = namespace2dict(P.raindrop_wavelet_config())
        mine["data"]["data_dir"], mine["data"]["num_workers"] = ref_cfg["data"]["data_dir"], ref_cfg["data"]["num_workers"]   # deployment keys
        assert mine == ref_cfg, "procedural.raindrop_wavelet_config() differs from the reference's configs/raindrop_wavelet.yml"
        canon = json.dumps(ref_cfg, sort_keys=True)
        np.savez_compressed(out("config.npz"), sha256=np.array(hashlib.sha256(canon.encode()).hexdigest()),
                            n_keys=np.array(sum(len(v) for v in ref_cfg.values())), sections=np.array(sorted(ref_cfg)))
        print(f"  procedural config == reference YAML ({sum(len(v) for v in ref_cfg.values())} keys)")

    # ------------------------------------------------------------------ eta != 0 (ddm_wavelet.py:500-502): the stochastic term of generalized_steps_overlapping
    def golden_eta():
        print("[eta]")
        c = P.reduced_config()
        c.device = torch.device("cpu")
        sd_e = P.procedural_state_dict(c)
        net_e = RU.DiffusionUNet(c).eval()
        net_e.load_state_dict(sd_e, strict=True)
        d = object.__new__(DenoisingDiffusion_Wavelet)
        d.config, d.device, d.model = c, torch.device("cpu"), net_e
        d.betas = torch.from_numpy(get_beta_schedule(beta_schedule="linear", beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
        d.num_timesteps = 1000
        S, eta = 6, 0.5
        xc, xT = seeded((1, 48, 24, 28), 810), seeded((1, 3, 24, 28), 811)
        corners = O.grid_corners(24, 28, 16, 4)
        seq = range(0, 1000, 1000 // S)
        torch.manual_seed(812)
        xs, x0p = d.generalized_steps_overlapping(xT, xc, seq, net_e, d.betas, eta=eta, corners=corners, p_size=16, x_other=xc[:, 3:], use_other=True)
        torch.manual_seed(812)
        noises = [torch.randn_like(xT) for _ in range(len(list(seq)))]      # the draws ddm_wavelet.py:502 made, in order
        oxs, ox0 = O.ddim_overlapping(sd_e, c, xT, xc, xc[:, 3:], corners, 16, S, eta=eta, noises=noises)
        assert len(xs) == len(oxs)
        check("eta=0.5 xs[-1]", oxs[-1], xs[-1])
        check("eta=0.5 x0_preds[-1]", ox0[-1], x0p[-1])
        check("eta=0.5 xs[2]", oxs[2], xs[2])
        np.savez_compressed(out("eta.npz"), eta=np.array(eta, dtype=np.float32), S=np.array(S, dtype=np.int32), x_T=xT.numpy(), x_cond=xc.numpy(),
                            noises=torch.stack(noises).numpy(), xs_last=xs[-1].numpy(), x0_last=x0p[-1].numpy(), xs_2=xs[2].numpy())

    # ------------------------------------------------------------------ the oracle stands in for the reference as the CPU baseline on the GPU box
    # only if it runs the same work in the same time here -- config 0 (4 x 64 x 64, 10 DDIM steps, fp32, 8 threads), reference and oracle interleaved, best of 3
    def golden_timing():
        import json
        import time
        print("[timing]  reference vs oracle wall time on BASELINE configs[0] (156 M procedural weights)")
        torch.set_num_threads(8)
        c = P.raindrop_wavelet_config()
        c.device = torch.device("cpu")
        sd_f = P.procedural_state_dict(c, seed=61)
        net_f = RU.DiffusionUNet(c).eval()
        net_f.load_state_dict(sd_f, strict=True)
        dec_ = WaveletTransform(scale=2, dec=True)
        d = object.__new__(DenoisingDiffusion_Wavelet)
        d.config, d.device, d.model = c, torch.device("cpu"), net_f
        d.args = SimpleNamespace(sampling_timesteps=10, resume="", local_rank=0, image_folder="/tmp/x", test_set="raindrop", grid_r=16)
        d.betas = torch.from_numpy(get_beta_schedule(beta_schedule="linear", beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
        d.num_timesteps = 1000
        rainy, x_T = P.synthetic_batch(4, patch_px=256, seed=61)
        x_cond = dec_(2 * rainy - 1)
        x_other = x_cond[:, 3:]

        def run_ref():
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):            # (the reference prints one line per step)
                return torch.cat([d.sample_image(x_cond[i:i + 1], x_T[i:i + 1], x_other=x_other[i:i + 1], last=False, patch_locs=[(0, 0)], patch_size=64,
                                                 use_other=True)[0][-1] for i in range(4)])

        def run_oracle():
            return O.ddim_batch(sd_f, c, x_T, x_cond, x_other, 10, chunk=1)[0][-1]

        tr, to = [], []
        run_ref(); run_oracle()                                          # warm-up (oneDNN primitive caches)
        for _ in range(int(os.environ.get("WDM_TIMING_ROUNDS", "3"))):
            t0 = time.perf_counter(); a = run_ref(); tr.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); b = run_oracle(); to.append(time.perf_counter() - t0)
        check("timing runs agree", b, a)
        r, o = min(tr), min(to)
        delta = (o - r) / r
        rec_ = {"workload": "BASELINE.json configs[0]: 4 x 64x64 wavelet-domain crops, 10 DDIM steps, fp32, torch CPU, 8 threads",
                "reference_s": [round(v, 3) for v in tr], "oracle_s": [round(v, 3) for v in to], "reference_best_s": round(r, 3), "oracle_best_s": round(o, 3),
                "oracle_minus_reference_rel": round(delta, 4), "threads": torch.get_num_threads(), "host_cpus": os.cpu_count(),
                "reference_img_per_s_at_100_steps": round(4 / (r * 10), 5), "oracle_img_per_s_at_100_steps": round(4 / (o * 10), 5)}
        print(" ", json.dumps(rec_))
        assert abs(delta) <= 0.05, f"oracle wall time differs from the reference's by {100 * delta:+.1f} % (BASELINE.md §3.2 allows 5 %)"
        with open(out("oracle_timing.json"), "w") as f:
            json.dump(rec_, f, indent=1)

    if args.only_eta:
        golden_eta()
        return
    if args.only_timing:
        golden_timing()
        return
    if args.only_config:
        golden_config()
        return
    if args.only_global:
        golden_global()
        return
    if args.only_variants:
        golden_variants()
        return
    if args.only_train:
        golden_train()
        return
    if args.only_hfrm:
        golden_hfrm()
        return
    if args.only_io:
        golden_io()
        return
    golden_hfrm()
    golden_io()
    golden_train()
    golden_variants()
    golden_config()
    golden_eta()

    # ------------------------------------------------------------------ integer tables
    print("[tables]")
    dec = WaveletTransform(scale=2, dec=True)
    rec = WaveletTransform(scale=2, dec=False)
    w = dec.conv.weight.detach()                                # (48,1,4,4)
    assert torch.equal(w[:16], w[16:32]) and torch.equal(w[:16], w[32:])
    assert float(w.abs().min()) == 0.25 == float(w.abs().max())
    rec4_sign = torch.sign(w[:16, 0]).reshape(16, 16).to(torch.int8)
    assert torch.equal(torch.sign(O.haar_filters()).reshape(16, 16).to(torch.int8), rec4_sign)
    assert torch.equal(O.haar_filters(), w[:16, 0])
    # sub-band permutation: feed one-hot conv channels through the reference's view/transpose
    perm = np.zeros(48, dtype=np.int32)                          # out channel -> conv channel c*16+j
    probe = torch.arange(48, dtype=torch.float32).view(1, 48, 1, 1)
    osz = probe.size()
    perm[:] = probe.view(1, 3, -1, 1, 1).transpose(1, 2).contiguous().view(osz).flatten().numpy()
    assert all(perm[j * 3 + c] == c * 16 + j for j in range(16) for c in range(3))
    ns = object.__new__(DenoisingDiffusion_Wavelet)
    tables = {"rec4_sign": rec4_sign.numpy(), "subband_perm": perm}
    for (h, wd, p, r) in [(64, 64, 64, 16), (120, 180, 64, 16), (128, 128, 64, 16), (65, 70, 64, 16),
                          (30, 45, 16, 4), (16, 16, 16, 4)]:
        hl, wl = ns.overlapping_grid_indices(torch.zeros(1, 1, h, wd), output_size=p, r=r)
        ohl, owl = O.overlapping_grid_indices(h, wd, p, r)
        assert hl == ohl and wl == owl
        tables[f"grid_h_{h}_{wd}_{p}_{r}"] = np.array(hl, dtype=np.int32)
        tables[f"grid_w_{h}_{wd}_{p}_{r}"] = np.array(wl, dtype=np.int32)
    corners = O.grid_corners(120, 180, 64, 16)
    mask = torch.zeros(1, 1, 120, 180)
    for (hi, wi) in corners:                                     # ddm_wavelet.py:451-453
        mask[:, :, hi:hi + 64, wi:wi + 64] += 1
    assert torch.equal(mask[0, 0].to(torch.int32), O.overlap_count_mask(120, 180, 64, corners))
    tables["mask_120_180_64_16"] = mask[0, 0].to(torch.int16).numpy()
    for S in (10, 25, 50, 100):
        seq = list(range(0, 1000, 1000 // S))                    # ddm_wavelet.py:296-297
        assert seq == O.timestep_seq(1000, S)
        tables[f"seq_{S}"] = np.array(seq, dtype=np.int32)
    cfg_full = P.raindrop_wavelet_config()
    betas = torch.from_numpy(get_beta_schedule(beta_schedule="linear", beta_start=1e-4, beta_end=0.02,
                                               num_diffusion_timesteps=1000)).float()
    assert torch.equal(betas, O.beta_schedule(cfg_full))
    abar = torch.stack([compute_alpha(betas, torch.tensor([t])).flatten()[0] for t in range(-1, 1000)])
    oabar = torch.stack([O.compute_alpha(betas, t) for t in range(-1, 1000)])
    assert torch.equal(abar, oabar)
    tables["alpha_bar_m1_to_999"] = abar.numpy()
    np.savez_compressed(out("tables.npz"), **tables)

    # ------------------------------------------------------------------ DWT known answers
    print("[dwt]")
    x = seeded((2, 3, 16, 16), 1, "rand") * 2 - 1
    y = dec(x)
    xr = rec(y)
    check("dwt_fwd", O.dwt_fwd(x), y, 1e-6)
    check("dwt_inv", O.dwt_inv(y), xr, 1e-6)
    x2 = seeded((1, 3, 8, 12), 2, "randn")
    y2 = dec(x2)
    check("dwt_fwd ragged", O.dwt_fwd(x2), y2, 1e-6)
    np.savez_compressed(out("dwt.npz"), x=x.numpy(), y=y.numpy(), xr=xr.numpy(), x2=x2.numpy(), y2=y2.numpy())

    # ------------------------------------------------------------------ per-block outputs
    print("[blocks]")
    blocks = {}

    def load(mod, prefix, seed=61):
        sd = {}
        for k, v in mod.state_dict().items():
            sd[k] = torch.from_numpy(P.procedural_tensor(prefix + "." + k, tuple(v.shape), seed))
        mod.load_state_dict(sd, strict=True)
        return {prefix + "." + k: v for k, v in sd.items()}

    # ResnetBlock with shortcut 64->128 @16, B=2, n_t = B
    rb = RU.ResnetBlock(in_channels=64, out_channels=128, dropout=0.0, temb_channels=512).eval()
    sd = load(rb, "rb_a")
    xin, temb = seeded((2, 64, 16, 16), 10), seeded((2, 512), 11)
    yref = rb(xin, temb)
    check("resblock 64->128@16", O.resnet_block(sd, "rb_a", xin, temb), yref)
    blocks["rb_a"] = yref.numpy()
    # ResnetBlock no shortcut 128->128 @8, temb broadcast n_t = 1
    rb = RU.ResnetBlock(in_channels=128, out_channels=128, dropout=0.0, temb_channels=512).eval()
    sd = load(rb, "rb_b")
    xin, temb = seeded((2, 128, 8, 8), 12), seeded((1, 512), 13)
    yref = rb(xin, temb)
    check("resblock 128->128@8", O.resnet_block(sd, "rb_b", xin, temb), yref)
    blocks["rb_b"] = yref.numpy()
    # ResnetBlock on a concat input 384->128 @16 (group width 12, straddle-free) B=1
    rb = RU.ResnetBlock(in_channels=384, out_channels=128, dropout=0.0, temb_channels=512).eval()
    sd = load(rb, "rb_c")
    xin, temb = seeded((1, 384, 16, 16), 14), seeded((1, 512), 15)
    yref = rb(xin, temb)
    check("resblock 384->128@16", O.resnet_block(sd, "rb_c", xin, temb), yref)
    blocks["rb_c"] = yref.numpy()
    # ResnetBlock 1280->768 @8: GroupNorm group (40 ch) straddles the [h(768) | skip(512)] seam
    rb = RU.ResnetBlock(in_channels=1280, out_channels=768, dropout=0.0, temb_channels=512).eval()
    sd = load(rb, "rb_d")
    xin, temb = seeded((1, 1280, 8, 8), 16), seeded((1, 512), 17)
    yref = rb(xin, temb)
    check("resblock 1280->768@8", O.resnet_block(sd, "rb_d", xin, temb), yref)
    blocks["rb_d"] = yref.numpy()
    # AttnBlock at the real shape: C=512, 16x16 (N=256), B=1 (subsampled) and C=64 @8x8, B=2
    ab = RU.AttnBlock(512).eval()
    sd = load(ab, "at_a")
    xin = seeded((1, 512, 16, 16), 20)
    yref = ab(xin)
    check("attn 512@16", O.attn_block(sd, "at_a", xin), yref)
    blocks["at_a_s5"] = sub(yref, 5)
    ab = RU.AttnBlock(64).eval()
    sd = load(ab, "at_b")
    xin = seeded((2, 64, 8, 8), 21)
    yref = ab(xin)
    check("attn 64@8", O.attn_block(sd, "at_b", xin), yref)
    blocks["at_b"] = yref.numpy()
    # mid attention shape: C=768 @8x8 (N=64), B=1
    ab = RU.AttnBlock(768).eval()
    sd = load(ab, "at_c")
    xin = seeded((1, 768, 8, 8), 22)
    yref = ab(xin)
    check("attn 768@8", O.attn_block(sd, "at_c", xin), yref)
    blocks["at_c"] = yref.numpy()
    # Downsample / Upsample
    ds = RU.Downsample(64, True).eval()
    sd = load(ds, "ds_a")
    xin = seeded((2, 64, 16, 16), 30)
    yref = ds(xin)
    check("downsample 64@16", O.downsample(sd, "ds_a", xin), yref)
    blocks["ds_a"] = yref.numpy()
    us = RU.Upsample(64, True).eval()
    sd = load(us, "us_a")
    xin = seeded((2, 64, 8, 8), 31)
    yref = us(xin)
    check("upsample 64@8", O.upsample(sd, "us_a", xin), yref)
    blocks["us_a"] = yref.numpy()
    # timestep embedding
    for tt in (0, 10, 990):
        e = RU.get_timestep_embedding(torch.tensor([float(tt)]), 128)
        check(f"timestep_embedding t={tt}", O.timestep_embedding(torch.tensor([float(tt)]), 128), e, 1e-6)
        blocks[f"temb_{tt}"] = e.numpy()
    np.savez_compressed(out("blocks.npz"), **blocks)

    # ------------------------------------------------------------------ reduced UNet + sampler
    print("[reduced]")
    cfg = P.reduced_config()
    cfg.device = torch.device("cpu")
    sd_r = P.procedural_state_dict(cfg, seed=61)
    net = RU.DiffusionUNet(cfg).eval()
    assert list(net.state_dict().keys()) == list(sd_r.keys()), "procedural key order != reference"
    net.load_state_dict(sd_r, strict=True)
    red = {}
    x96 = seeded((2, 96, 16, 16), 40)
    for name, t in (("t500", torch.tensor([500.0])), ("t_per_image", torch.tensor([990.0, 10.0]))):
        yref = net(x96, t)
        check(f"reduced unet fwd {name}", O.unet_forward(sd_r, cfg, x96, t), yref)
        red["fwd_" + name] = yref.numpy()

    def ref_diffusion(cfg_, net_, S):
        d = object.__new__(DenoisingDiffusion_Wavelet)
        d.config, d.device, d.model = cfg_, torch.device("cpu"), net_
        d.args = SimpleNamespace(sampling_timesteps=S, resume="", local_rank=0, image_folder="/tmp/x",
                                 test_set="raindrop", grid_r=16)
        d.betas = torch.from_numpy(get_beta_schedule(
            beta_schedule="linear", beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
        d.num_timesteps = 1000
        d.wavelet_dec, d.wavelet_rec = dec, rec
        d.generator = lambda x_: x_                               # identity HFRM stand-in
        return d

    # 10-step sampler on 2 independent 16x16 patches through the bound reference methods
    rainy, x_T = P.synthetic_batch(2, patch_px=64, seed=61)
    x_cond = dec(2 * rainy - 1)
    x_other = x_cond[:, 3:]
    d = ref_diffusion(cfg, net, 10)
    xs_l, x0_l = [], []
    for i in range(2):
        xs, x0p = d.sample_image(x_cond[i:i + 1], x_T[i:i + 1], x_other=x_other[i:i + 1], last=False,
                                 patch_locs=[(0, 0)], patch_size=16, use_other=True)
        xs_l.append(xs[-1]); x0_l.append(x0p[-5])
    ref_xs, ref_x0 = torch.cat(xs_l), torch.cat(x0_l)
    oxs, ox0 = O.ddim_batch(sd_r, cfg, x_T, O.dwt_fwd(2 * rainy - 1), O.dwt_fwd(2 * rainy - 1)[:, 3:], 10)
    check("reduced sampler xs[-1]", oxs[-1], ref_xs)
    check("reduced sampler x0[-5]", ox0[-5], ref_x0)
    red["samp_xs_last"], red["samp_x0_m5"] = ref_xs.numpy(), ref_x0.numpy()
    np.savez_compressed(out("reduced.npz"), **red)

    # ------------------------------------------------------------------ C4-shaped stitch (reduced model)
    print("[stitch]")
    st = {}
    g = torch.Generator().manual_seed(77)
    img = torch.rand(1, 3, 120, 180, generator=g)                 # -> 30 x 45 wavelet domain, p=16, r=4
    gt = torch.rand(1, 3, 120, 180, generator=g)
    S = 6
    d = ref_diffusion(cfg, net, S)
    rargs = SimpleNamespace(resume="", image_folder="/tmp/wdm_golden", sampling_timesteps=S)
    import utils as RUT
    RUT.calculate_psnr = lambda *a, **k: 0.0                      # numpy/cv2 metric: not on the path
    RUT.calculate_psnr_in_GPU = lambda *a, **k: torch.tensor(0.0)
    restorer = DiffusiveRestoration(d, rargs, cfg)
    torch.manual_seed(123)
    SAVED_IMAGES.clear()
    restorer.restore([(torch.cat([img, gt], dim=1), "img0", torch.zeros(1))], validation="raindrop", r=4)
    ref_out = SAVED_IMAGES["img0_output.png"]
    torch.manual_seed(123)
    x_T_full = torch.randn(1, 3, 30, 45)                          # the draw restoration.py:177 made
    o_out, o_xs, o_x0 = O.restore(sd_r, cfg, img, x_T_full, S, r=4)
    check("restore() stitched output", o_out, ref_out)
    st["out"] = ref_out.numpy()
    st["x_T"] = x_T_full.numpy()
    st["n_corners"] = np.array(len(O.grid_corners(30, 45, 16, 4)), dtype=np.int32)
    np.savez_compressed(out("stitch.npz"), **st)

    # ------------------------------------------------------------------ full-width (156 M params)
    if not args.skip_full:
        print("[full]  (generating 156 M procedural weights)")
        cfg = P.raindrop_wavelet_config()
        cfg.device = torch.device("cpu")
        sd_f = P.procedural_state_dict(cfg, seed=61)
        net = RU.DiffusionUNet(cfg).eval()
        assert list(net.state_dict().keys()) == list(sd_f.keys())
        net.load_state_dict(sd_f, strict=True)
        nparams = sum(v.numel() for v in sd_f.values())
        print("  params:", nparams)
        full = {"n_params": np.array(nparams, dtype=np.int64)}
        rainy, x_T = P.synthetic_batch(4, patch_px=256, seed=61)
        x_cond = dec(2 * rainy - 1)
        x_other = x_cond[:, 3:]
        x96 = torch.cat([x_cond[:2], x_T[:2], x_other[:2]], dim=1)
        yref = net(x96, torch.tensor([990.0]))
        check("full unet fwd t=990", O.unet_forward(sd_f, cfg, x96, torch.tensor([990.0])), yref)
        full["fwd_t990"] = yref.numpy()
        d = ref_diffusion(cfg, net, 10)
        xs_l, x0_l = [], []
        for i in range(4):                                        # config 0: 4x64x64, 10 DDIM steps
            xs, x0p = d.sample_image(x_cond[i:i + 1], x_T[i:i + 1], x_other=x_other[i:i + 1], last=False,
                                     patch_locs=[(0, 0)], patch_size=64, use_other=True)
            xs_l.append(xs[-1]); x0_l.append(x0p[-5])
        ref_xs, ref_x0 = torch.cat(xs_l), torch.cat(x0_l)
        oxs, ox0 = O.ddim_batch(sd_f, cfg, x_T, O.dwt_fwd(2 * rainy - 1), O.dwt_fwd(2 * rainy - 1)[:, 3:], 10,
                                chunk=1)
        check("C0 sampler xs[-1]", oxs[-1], ref_xs)
        check("C0 sampler x0[-5]", ox0[-5], ref_x0)
        full["c0_xs_last"], full["c0_x0_m5"] = ref_xs.numpy(), ref_x0.numpy()
        np.savez_compressed(out("full.npz"), **full)
    if not args.skip_full:
        golden_timing()
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
