"""GPU: output side of DiffusiveRestoration.restore through the C ABI -- metrics, 8-bit conversion,
asynchronous PNG writes, several images per sampler call, and the loader -> restore pipeline end to end."""
import os
import random
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from gpu_util import dev, seeded
from oracle import wavedm_oracle as O
from wavedm_amd import imageio
from wavedm_amd import procedural as P

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def test_metrics_match_reference_golden(golden):
    g = golden("io.npz")
    gt = seeded((2, 3, 32, 48), 301, "rand")
    out = (gt + 0.1 * seeded((2, 3, 32, 48), 302)).clamp(-0.2, 1.2)
    m = imageio.psnr_from_sums(imageio.sqdiff(gt.to(dev()), out.to(dev())), 32, 48)
    mc = imageio.psnr_from_sums(imageio.sqdiff(gt.to(dev()), out.clamp(0, 1).to(dev())), 32, 48)
    for k in range(2):
        assert abs(m[k][0] - g["psnr"][k, 0]) < 1e-4          # torchPSNR
        assert abs(m[k][1] - g["psnr"][k, 1]) < 1e-4          # calculate_psnr_in_GPU(.., True)
        assert abs(mc[k][1] - g["psnr"][k, 2]) < 1e-3         # numpy calculate_psnr(.., True) on the clamped 0..255 images
    # full-size image, odd pixel count per thread
    a, b = seeded((1, 3, 480, 720), 7, "rand"), seeded((1, 3, 480, 720), 8, "rand")
    m = imageio.psnr_from_sums(imageio.sqdiff(a.to(dev()), b.to(dev())), 480, 720)
    assert abs(m[0][0] - O.psnr_torch(a, b)) < 1e-6 and abs(m[0][1] - O.psnr_y(a, b)) < 1e-6


def test_to_u8_hwc_exact():
    x = seeded((2, 3, 33, 47), 11) * 0.5 + 0.5            # values below 0 and above 1 included
    x[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.5 / 255, 254.5 / 255])
    got = imageio.to_u8_hwc(x.to(dev())).cpu()
    assert torch.equal(got, O.to_u8_hwc(x))
    y = seeded((1, 1, 16, 16), 12, "rand")
    assert torch.equal(imageio.to_u8_hwc(y.to(dev())).cpu(), O.to_u8_hwc(y))


def _diffusion(S, generator=lambda x: x):
    from test_gpu_unet import make_diffusion
    d, args = make_diffusion(P.reduced_config(), "f32", S, generator=generator)
    return d, args


def test_images_per_call_is_bit_identical_and_pngs_match(tmp_path):
    import wavedm_amd
    from PIL import Image
    d, args = _diffusion(6)
    g = torch.Generator().manual_seed(21)
    items = [(torch.rand(1, 6, 96, 112, generator=g), (f"im{k}",), torch.zeros(1)) for k in range(3)]
    items.append((torch.rand(1, 6, 64, 80, generator=g), ("im3",), torch.zeros(1)))       # a different size closes the group
    outs = {}
    for per_call in (1, 2, 4):
        args.images_per_call = per_call
        args.image_folder = str(tmp_path / f"n{per_call}")
        rest = wavedm_amd.DiffusiveRestoration(d, args, d.config, save_images=True)
        torch.manual_seed(5)
        o, psnr = rest.restore(items, validation="raindrop", r=4)
        rest.writer.close()
        outs[per_call] = [t.cpu() for t in o]
        assert len(o) == 4 and len(psnr) == 4 and all(np.isfinite(psnr))
    for k in range(4):
        assert torch.equal(outs[1][k], outs[2][k]) and torch.equal(outs[1][k], outs[4][k])
    # the PNGs hold exactly save_image's quantisation of the returned tensors, for every variant
    for per_call in (1, 4):
        folder = tmp_path / f"n{per_call}" / d.config.data.dataset / "raindrop"
        for k in range(4):
            png = np.asarray(Image.open(folder / f"im{k}_output.png"))
            assert np.array_equal(png, O.to_u8_hwc(outs[1][k])[0].numpy())
            for suffix in ("cond", "gt", "all_wdnet", "lrgt_hrwdnet", "lrgt_hrcond", "lrdiff_hrgt"):
                assert (folder / f"im{k}_{suffix}.png").is_file()
            gt_png = np.asarray(Image.open(folder / f"im{k}_gt.png"))
            assert np.array_equal(gt_png, O.to_u8_hwc(items[k][0][:, 3:])[0].numpy())


def test_restore_defaults_early_stop_and_auto_grouping_are_bit_identical(tmp_path):
    """restore() as a caller of the reference gets it (no images_per_call / early_stop in args): early stop at x0_preds[-5] (restoration.py:108 reads nothing behind
    it), as many same-sized images per sampler call as fill the UNet calls, groups pipelined two deep -- against the plain loop (every step, one image per call):
    the same bits per image, the same PSNRs, the same console order."""
    import wavedm_amd
    d, args = _diffusion(8)
    g = torch.Generator().manual_seed(33)
    items = [(torch.rand(1, 6, 96, 112, generator=g), (f"im{k}",), torch.zeros(1)) for k in range(7)]       # 24x28 wavelet domain: 3 x 4 = 12 patches of 16
    items.insert(4, (torch.rand(1, 6, 64, 80, generator=g), ("odd",), torch.zeros(1)))                        # a different size in the middle closes a group
    args.max_batch = 32                                                                                     # UNet calls of at most 32 patches: 48 = 32 + 16 -> two calls of 24, ...
    res = {}
    for tag, kw in (("plain", dict(images_per_call=1, early_stop=False)), ("default", {}), ("early1", dict(images_per_call=1)), ("late_auto", dict(early_stop=False))):
        a = SimpleNamespace(**vars(args))
        for k, v in kw.items():
            setattr(a, k, v)
        a.image_folder = str(tmp_path / tag)
        rest = wavedm_amd.DiffusiveRestoration(d, a, d.config, save_images=False)
        if tag == "default":
            assert rest.images_per_call_for(24, 28, 4) == 8 and rest.images_per_call_for(16, 20, 4) > 1          # several images per call, by default
        torch.manual_seed(5)
        o, psnr = rest.restore(items, validation="raindrop", r=4)
        res[tag] = ([t.cpu() for t in o], list(psnr), list(rest.last_psnrs_y))
    assert len(res["plain"][0]) == 8
    for tag in ("default", "early1", "late_auto"):
        for k in range(8):
            assert torch.equal(res["plain"][0][k], res[tag][0][k]), (tag, k)
        assert res[tag][1] == res["plain"][1] and res[tag][2] == res["plain"][2]
    # the early stop really stops: sample_image hands back None behind x0_preds[-5]
    x_cond = d.wavelet_dec.forward_affine(items[0][0][:, :3].to(dev()).contiguous())
    torch.manual_seed(1)
    xs, x0 = wavedm_amd.DiffusiveRestoration(d, args, d.config, save_images=False).diffusive_restoration(
        x_cond, x_other=x_cond[:, 3:].contiguous(), r=4, last=False, use_other=True, stop_at=-5)
    assert len(x0) == 8 and x0[-5] is not None and all(t is None for t in x0[-4:]) and all(t is None for t in xs[-4:])
    with pytest.raises(ValueError):
        d.sample_image(x_cond, torch.randn(1, 3, 24, 28, device=dev()), x_other=x_cond[:, 3:].contiguous(), last=True, patch_locs=[(0, 0)], patch_size=16,
                       use_other=True, stop_at=-5)


def test_restore_surfaces_loader_errors_and_short_schedules(tmp_path):
    import wavedm_amd
    d, args = _diffusion(6)
    args.image_folder = str(tmp_path)

    def bad_loader():
        yield torch.rand(1, 6, 64, 80), ("ok",), torch.zeros(1)
        raise OSError("truncated PNG")
    rest = wavedm_amd.DiffusiveRestoration(d, args, d.config, save_images=False)
    with pytest.raises(OSError, match="truncated PNG"):
        rest.restore(bad_loader(), validation="raindrop", r=4)
    assert rest.restore([], validation="raindrop", r=4) == ([], [])                      # an empty loader: nothing queued, nothing printed
    # a loader batch of several images, 5-D as a parse_patches loader yields them (restoration.py:72 flattens): split into images, the items keep their order
    g = torch.Generator().manual_seed(8)
    xb = torch.rand(1, 3, 6, 64, 80, generator=g)
    torch.manual_seed(3)
    o5, _ = rest.restore([(xb, ("p",), torch.zeros(1))], validation="raindrop", r=4)
    torch.manual_seed(3)
    o1, _ = rest.restore([(xb[0, k:k + 1], (f"p{k}",), torch.zeros(1)) for k in range(3)], validation="raindrop", r=4)
    assert len(o5) == 3 and all(torch.equal(a, b) for a, b in zip(o5, o1))
    d4, a4 = _diffusion(4)                                      # x0_preds[-5] of a 4-step run: the reference's IndexError (restoration.py:108)
    a4.image_folder = str(tmp_path)
    with pytest.raises(IndexError):
        wavedm_amd.DiffusiveRestoration(d4, a4, d4.config, save_images=False).restore([(torch.rand(1, 6, 64, 80), ("x",), torch.zeros(1))], r=4)


def test_loader_to_restore_pipeline(tmp_path):
    """datasets.RainDrop loaders -> DiffusiveRestoration.restore, two images per call, PSNR against the oracle's restore()."""
    import wavedm_amd
    from wavedm_amd.datasets import RainDrop
    O.synthetic_raindrop_dir(str(tmp_path), seed=303, sizes=((200, 140), (180, 120)))
    os.makedirs(tmp_path / "raindrop" / "train" / "input"); os.makedirs(tmp_path / "raindrop" / "train" / "gt")
    d, args = _diffusion(5)
    cfg = d.config
    cfg.data.data_dir = str(tmp_path)
    cfg.training = SimpleNamespace(patch_n=2, batch_size=1)
    args.world_size, args.rank = 1, 0
    random.seed(3)
    # keep the test small: the loader's fixed 720x480 resize is exercised by the CPU test; here the images are shrunk again
    _, val_loader = RainDrop(args, cfg).get_loaders(parse_patches=False, validation="raindrop")
    small = [(torch.nn.functional.interpolate(x, size=(96, 128), mode="bilinear"), y, t) for x, y, t in val_loader]
    assert len(small) == 2 and small[0][0].shape == (1, 6, 96, 128)
    args.images_per_call = 2
    args.image_folder = str(tmp_path / "out")
    rest = wavedm_amd.DiffusiveRestoration(d, args, cfg, save_images=False)
    torch.manual_seed(9)
    outs, psnrs = rest.restore(small, validation="raindrop", r=4)
    torch.manual_seed(9)
    xT = [torch.randn((1, 3, 24, 32), device=dev()).cpu() for _ in range(2)]
    sd = P.procedural_state_dict(cfg)
    for k in range(2):
        want, _, _ = O.restore(sd, cfg, small[k][0][:, :3], xT[k], 5, r=4)
        assert float((outs[k].cpu() - want).abs().max()) <= 1e-3
        assert abs(psnrs[k] - O.psnr_torch(small[k][0][:, 3:], want)) < 1e-2


def test_writer_last_save_to_a_path_wins(tmp_path):
    """Several encoder threads take the queue's FIFO order away: saves to the SAME path are ordered by sequence number -- whichever thread gets there
    first, the file holds the LAST image saved to it."""
    import numpy as np
    from PIL import Image
    from wavedm_amd.imageio import AsyncImageWriter
    w = AsyncImageWriter(workers=8)
    paths = [str(tmp_path / f"p{k}.png") for k in range(4)]
    for rep in range(6):
        for k, p in enumerate(paths):
            img = torch.full((1, 3, 480, 720), (rep * 4 + k) / 255.0, device="cuda")
            w.save(img, p)
    w.close()
    for k, p in enumerate(paths):
        assert int(np.asarray(Image.open(p))[0, 0, 0]) == 5 * 4 + k
