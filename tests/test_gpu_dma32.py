"""The experimental forms of the LDS-DMA 3x3 kernel (off by default) must produce the bits of the shipped one: the 512 x 128 tile
(WDM_DMA32=2: same K order, same pixel sets and association per GroupNorm statistics slab) and the K loop that reads the next
sub-stage's fragments behind its barrier (WDM_DMA_PF=1) -- conv_dma_kernel.h."""
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


def _both(f, var="WDM_DMA32", on="2"):
    from wavedm_amd import _lib
    old = os.environ.get(var)
    try:
        os.environ[var] = "0"
        _lib.env_refresh()
        y0 = f()
        os.environ[var] = on
        _lib.env_refresh()
        y1 = f()
    finally:
        if old is None:
            os.environ.pop(var, None)
        else:
            os.environ[var] = old
        _lib.env_refresh()
    return y0, y1


@pytest.mark.parametrize("cin,cout,B,H", [(128, 128, 2, 32), (256, 128, 3, 64), (96, 256, 2, 32), (768, 256, 1, 32)])
def test_conv_bits(gu, cin, cout, B, H):
    w = gu.seeded((cout, cin, 3, 3), 100) / (cin * 9) ** 0.5
    b = gu.seeded((cout,), 200) * 0.1
    x = gu.seeded((B, cin, H, H), 316)
    y0, y1 = _both(lambda: gu.conv(w, b, 0, x, "bf16"))
    assert torch.equal(y0, y1)
    ref = torch.nn.functional.conv2d(x, w, b, padding=1)
    assert float((y1 - ref).abs().max() / ref.abs().max()) <= gu.TOL["bf16"]


@pytest.mark.parametrize("cin,cout,B,H", [(128, 128, 2, 32), (256, 128, 2, 64), (128, 256, 2, 32)])
def test_resblock_bits(gu, cin, cout, B, H):
    """GroupNorm statistics from the epilogue, temb, residual and the fused 1x1 shortcut (cin != cout) through both tilings."""
    shapes = {"norm1.weight": (cin,), "norm1.bias": (cin,), "conv1.weight": (cout, cin, 3, 3), "conv1.bias": (cout,), "temb_proj.weight": (cout, 512),
              "temb_proj.bias": (cout,), "norm2.weight": (cout,), "norm2.bias": (cout,), "conv2.weight": (cout, cout, 3, 3), "conv2.bias": (cout,)}
    if cin != cout:
        shapes["nin_shortcut.weight"] = (cout, cin, 1, 1)
        shapes["nin_shortcut.bias"] = (cout,)
    sd = gu.blk_sd("rb", shapes)
    x = gu.seeded((B, cin, H, H), 5)
    t = gu.seeded((B, 512), 6)
    y0, y1 = _both(lambda: gu.resblock(sd, "rb", x, None, t, "bf16"))
    assert torch.isfinite(y1).all() and torch.equal(y0, y1)


@pytest.mark.parametrize("cin,cout,B,H", [(128, 128, 2, 32), (256, 128, 3, 64), (96, 256, 2, 32), (512, 512, 3, 16)])
def test_prefetching_k_loop_bits(gu, cin, cout, B, H):
    w = gu.seeded((cout, cin, 3, 3), 100) / (cin * 9) ** 0.5
    b = gu.seeded((cout,), 200) * 0.1
    x = gu.seeded((B, cin, H, H), 316)
    y0, y1 = _both(lambda: gu.conv(w, b, 0, x, "bf16"), "WDM_DMA_PF", "1")
    assert torch.equal(y0, y1)
