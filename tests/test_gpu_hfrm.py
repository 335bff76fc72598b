"""GPU parity of the HFRM through the C ABI (wdm_hfrm_*): HIP path vs the reference's own outputs
(tests/golden/hfrm.npz) and vs the oracle at sizes the golden file does not hold."""
import pytest
import torch

from conftest import rel_linf
from gpu_util import TOL, dev, seeded
from oracle import wavedm_oracle as O
from wavedm_amd import procedural as P
from wavedm_amd.arch import HFRM

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

HF_TOL = {"f32": 1e-3, "bf16": 6e-2}      # 36 blocks deep: bf16 activations drift more than in the UNet


def make(dtype):
    m = HFRM(in_channel=3, dim=32, mid_blk_num=6, enc_blk_nums=[2, 2, 2, 4], dec_blk_nums=[2, 2, 2, 2], dtype=dtype)
    sd = P.procedural_hfrm_state_dict(seed=61)
    m.load_state_dict(sd, strict=True)
    return m.to(dev()), sd


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_hfrm_matches_reference_golden(golden, dtype):
    g = golden("hfrm.npz")
    m, _ = make(dtype)
    for tag in ("a", "b"):
        x = seeded(tuple(int(v) for v in g["shape_" + tag]), int(g["seed_" + tag]), "rand")
        y = m(x.to(dev())).cpu()
        want = torch.from_numpy(g["y_" + tag])
        assert rel_linf(y, want) <= HF_TOL[dtype], (tag, rel_linf(y, want))


def test_hfrm_larger_and_batch_independent():
    m, sd = make("f32")
    x = seeded((3, 3, 128, 176), 17, "rand")
    y = m(x.to(dev())).cpu()
    want = O.hfrm_forward(sd, x)
    assert rel_linf(y, want) <= HF_TOL["f32"]
    y1 = m(x[1:2].to(dev())).cpu()
    assert torch.equal(y1, y[1:2])            # an image's result does not depend on its batch


def test_hfrm_rejects_bad_sizes():
    m, _ = make("bf16")
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 40, 64, device=dev()))      # 40 is not a multiple of 16
    with pytest.raises(TypeError):
        m(torch.zeros(1, 3, 64, 64))                    # CPU tensor: no CPU path


def test_restore_with_hfrm_matches_oracle():
    """restoration.py:88-134 end to end with the HFRM on the device: HFRM -> DWT -> x_other -> stitched DDIM -> IDWT."""
    import wavedm_amd
    from test_gpu_unet import make_diffusion
    cfg = P.reduced_config()
    S = 6
    d, args = make_diffusion(cfg, "f32", S, generator="procedural")
    sd_h = P.procedural_hfrm_state_dict(seed=61)
    g = torch.Generator().manual_seed(78)
    img = torch.rand(1, 3, 96, 112, generator=g)
    gt = torch.rand(1, 3, 96, 112, generator=g)
    x_T = torch.randn(1, 3, 24, 28, generator=g)
    want, _, _ = O.restore(P.procedural_state_dict(cfg), cfg, img, x_T, S, r=4, hfrm=lambda x: O.hfrm_forward(sd_h, x))
    rest = wavedm_amd.DiffusiveRestoration(d, args, d.config, save_images=False)
    xT = x_T.cuda()
    real_randn = torch.randn
    torch.randn = lambda *a, **k: xT.clone()
    try:
        outs, _ = rest.restore([(torch.cat([img, gt], 1), "img0", torch.zeros(1))], validation="raindrop", r=4)
    finally:
        torch.randn = real_randn
    assert rel_linf(outs[0].cpu(), want) <= TOL["f32"]
