"""The big tilings of the LDS-DMA 3x3 kernel (conv_dma256_kernel.h: 256 x 256 output tiles where Cout is a multiple of 256, 512 x 128 tiles -- 32 x 16 pixels --
on maps that are multiples of 32 rows) must produce the BITS of the 256 x 128 tile (conv_dma_kernel.h): same K order per pixel, same pixel sets and association per GroupNorm statistics slab -- the launcher picks
a tiling by workgroup count, so an image's result must not depend on it.  Checked through the C ABI on single convs (vs torch as well) and on
whole ResnetBlocks (statistics from the epilogue, temb, residual, fused 1x1 shortcut over a concat input)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def gu():
    import gpu_util
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return gpu_util


def _modes(f, modes=("0", "2", "1")):
    """f() under every tiling switch.  WDM_GN_TILE=1 throughout: whether conv1 of a 16 x 16 ResnetBlock can also normalise for conv2 depends on the tile it runs
    on (only the 256 x 128 one holds a whole image) -- these tests are about the tilings' bits, not about where the norm is computed."""
    from wavedm_amd import _lib
    old = os.environ.get("WDM_BN256")
    old_t = os.environ.get("WDM_GN_TILE")
    out = []
    try:
        os.environ["WDM_GN_TILE"] = "1"
        for m in modes:
            os.environ["WDM_BN256"] = m
            _lib.env_refresh()
            out.append(f())
    finally:
        if old is None:
            os.environ.pop("WDM_BN256", None)
        else:
            os.environ["WDM_BN256"] = old
        if old_t is None:
            os.environ.pop("WDM_GN_TILE", None)
        else:
            os.environ["WDM_GN_TILE"] = old_t
        _lib.env_refresh()
    return out


@pytest.mark.parametrize("cin,cout,B,H", [(128, 256, 2, 32), (256, 256, 3, 32), (768, 256, 1, 32), (512, 512, 3, 16), (1280, 512, 2, 16), (96, 256, 1, 48),
                                          (128, 128, 2, 64), (96, 128, 1, 64), (384, 128, 1, 64), (128, 128, 3, 32), (64, 384, 1, 32)])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_conv_bits(gu, dtype, cin, cout, B, H):
    w = gu.seeded((cout, cin, 3, 3), 100) / (cin * 9) ** 0.5
    b = gu.seeded((cout,), 200) * 0.1
    x = gu.seeded((B, cin, H, H), 316)
    ys = _modes(lambda: gu.conv(w, b, 0, x, dtype))
    for y in ys[1:]:
        assert torch.equal(ys[0], y)
    ref = torch.nn.functional.conv2d(x, w, b, padding=1)
    assert float((ys[1] - ref).abs().max() / ref.abs().max()) <= gu.TOL[dtype]


@pytest.mark.parametrize("c0,c1,cout,B,H", [(256, 0, 256, 2, 32), (256, 256, 256, 2, 32), (512, 256, 256, 1, 32), (512, 0, 512, 3, 16), (512, 512, 512, 2, 16),
                                            (768, 512, 512, 1, 16), (128, 0, 128, 2, 64), (128, 128, 128, 1, 64), (256, 128, 128, 1, 64), (128, 0, 128, 3, 32)])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_resblock_bits(gu, dtype, c0, c1, cout, B, H):
    """conv1 (prologue over the concat, temb, statistics) and conv2 (statistics, residual or the fused 1x1 shortcut) through every tiling."""
    cin = c0 + c1
    shapes = {"norm1.weight": (cin,), "norm1.bias": (cin,), "conv1.weight": (cout, cin, 3, 3), "conv1.bias": (cout,), "temb_proj.weight": (cout, 512),
              "temb_proj.bias": (cout,), "norm2.weight": (cout,), "norm2.bias": (cout,), "conv2.weight": (cout, cout, 3, 3), "conv2.bias": (cout,)}
    if cin != cout:
        shapes["nin_shortcut.weight"] = (cout, cin, 1, 1)
        shapes["nin_shortcut.bias"] = (cout,)
    sd = gu.blk_sd("rb", shapes)
    x0 = gu.seeded((B, c0, H, H), 5)
    x1 = gu.seeded((B, c1, H, H), 7) if c1 else None
    t = gu.seeded((B, 512), 6)
    ys = _modes(lambda: gu.resblock(sd, "rb", x0, x1, t, dtype))
    assert torch.isfinite(ys[0]).all()
    for y in ys[1:]:
        assert torch.equal(ys[0], y)


@pytest.mark.parametrize("c0,c1,cout,B,H", [(128, 0, 128, 2, 64), (128, 128, 128, 1, 64), (96, 0, 128, 1, 32), (256, 0, 128, 3, 32), (256, 0, 256, 2, 32),
                                            (256, 256, 256, 1, 32), (128, 0, 256, 3, 16)])
def test_resblock_bits_f32x3(gu, c0, c1, cout, B, H):
    """The f32x3 mode's 512 x 128 and 256 x 256 tiles (conv_dmax3t_kernel.h; they need the block's pre-split weight copy, so it is reached through the ResnetBlock, not the single
    conv): the bits of conv_dmax3_kernel.h."""
    cin = c0 + c1
    shapes = {"norm1.weight": (cin,), "norm1.bias": (cin,), "conv1.weight": (cout, cin, 3, 3), "conv1.bias": (cout,), "temb_proj.weight": (cout, 512),
              "temb_proj.bias": (cout,), "norm2.weight": (cout,), "norm2.bias": (cout,), "conv2.weight": (cout, cout, 3, 3), "conv2.bias": (cout,)}
    if cin != cout:
        shapes["nin_shortcut.weight"] = (cout, cin, 1, 1)
        shapes["nin_shortcut.bias"] = (cout,)
    sd = gu.blk_sd("rb", shapes)
    x0 = gu.seeded((B, c0, H, H), 5)
    x1 = gu.seeded((B, c1, H, H), 7) if c1 else None
    t = gu.seeded((B, 512), 6)
    ys = _modes(lambda: gu.resblock(sd, "rb", x0, x1, t, "f32x3"), modes=("0", "2"))
    assert torch.isfinite(ys[0]).all()
    assert torch.equal(ys[0], ys[1])
