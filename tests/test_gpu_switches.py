"""Alternative code paths behind the (ten) environment switches must agree with the default ones: bit for bit where the arithmetic is the same (tilings, proj_out fused into the attention core, in-tile GroupNorm), within the bf16 bound where it is reordered (register-staged instead of LDS-DMA
kernels, GroupNorm finalised in the consumer's prologue)."""
import os

import pytest
import torch

from conftest import rel_linf

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def gu():
    import gpu_util
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return gpu_util


def _with(env, f):
    from wavedm_amd import _lib
    old = {k: os.environ.get(k) for k in env}
    try:
        os.environ.update(env)
        _lib.env_refresh()
        return f()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _lib.env_refresh()


@pytest.mark.parametrize("cin,cout,B", [(768, 768, 5), (512, 768, 2), (1536, 768, 3), (96, 384, 4)])
def test_8x8_conv_on_the_lds_dma_and_the_register_staged_kernel(gu, cin, cout, B):
    """WDM_CONV_DMA=0 takes every LDS-DMA 3x3 kernel away: the 8x8 layers then run on conv_kernel.h, which walks K in another order -- the same bound
    against torch, other bits.  (Round 3's WDM_WSM / WDM_DMA8_BN / WDM_DMA8 switches are gone: slab-major weights and the Cout-determined N tile are fixed.)"""
    w = gu.seeded((cout, cin, 3, 3), 31) / (cin * 9) ** 0.5
    b = gu.seeded((cout,), 32) * 0.1
    x = gu.seeded((B, cin, 8, 8), 33)
    ref = torch.nn.functional.conv2d(x, w, b, padding=1)
    from wavedm_amd import _lib

    def run():
        _lib.prof_enable(True)
        out = gu.conv(w, b, 0, x, "bf16")
        names = [e["kernel"].split("|")[0] for e in _lib.prof_report()]
        _lib.prof_enable(False)
        return out, names
    y, k1 = run()
    y0, k0 = _with({"WDM_CONV_DMA": "0"}, run)
    assert rel_linf(y, ref) <= gu.TOL["bf16"] and rel_linf(y0, ref) <= gu.TOL["bf16"] and rel_linf(y0, y) <= gu.TOL["bf16"]
    assert any("convdma8" in n and ("bn48" if cout % 48 == 0 else "bn64") in n for n in k1), k1
    assert any(n.startswith("conv_3x3s1_t8x8") for n in k0) and not any("convdma8" in n for n in k0), k0      # the switch really switched
    assert torch.equal(y, run()[0])


@pytest.mark.parametrize("c,H,B", [(256, 32, 3), (512, 16, 5), (128, 32, 2)])
def test_groupnorm_finalised_in_the_conv_prologue_agrees_with_gn_finalize(gu, c, H, B):
    """gn_inline.h: conv2 of a ResnetBlock finalises its GroupNorm from conv1's group-level partials inside its own prologue (maps up to 32 x 32) instead of
    a gn_finalize launch.  Same statistics summed in another association (fp32 group merge in the producer, fp64 over the slabs in the consumer):
    agreement far inside the bf16 bound, and the f32 oracle is the yardstick for both."""
    shapes = {"norm1.weight": (c,), "norm1.bias": (c,), "conv1.weight": (c, c, 3, 3), "conv1.bias": (c,), "temb_proj.weight": (c, 512),
              "temb_proj.bias": (c,), "norm2.weight": (c,), "norm2.bias": (c,), "conv2.weight": (c, c, 3, 3), "conv2.bias": (c,)}
    sd = gu.blk_sd("rb", shapes)
    x, t = gu.seeded((B, c, H, H), 5), gu.seeded((B, 512), 6)
    # (WDM_GN_TILE=1: on 16 x 16 maps conv1 would otherwise normalise for conv2 itself -- test_conv1_normalises_for_conv2_on_16x16_maps)
    y = _with({"WDM_GN_TILE": "1"}, lambda: gu.resblock(sd, "rb", x, None, t, "bf16"))          # in-prologue finalize for conv2
    y_fin = _with({"WDM_GN_INLINE": "0", "WDM_GN_TILE": "1"}, lambda: gu.resblock(sd, "rb", x, None, t, "bf16"))
    y_f32 = gu.resblock(sd, "rb", x, None, t, "f32")
    assert not torch.equal(y, y_fin)                                                # two different paths really ran
    # measured: 0.0004 % ... 0.06 % of the outputs differ, each by ONE bf16 ulp (a conv input that rounded the other way): <= 2^-7 of the largest output
    assert rel_linf(y, y_fin) <= 8e-3 and float(((y - y_fin).abs() > 0).float().mean()) <= 1e-2
    assert rel_linf(y, y_f32) <= gu.TOL["bf16"] and rel_linf(y_fin, y_f32) <= gu.TOL["bf16"]
    assert torch.equal(y, _with({"WDM_GN_TILE": "1"}, lambda: gu.resblock(sd, "rb", x, None, t, "bf16")))                # deterministic
    for env in ({"WDM_BN256": "0"}, {"WDM_BN256": "2"}):
        env = dict(env, WDM_GN_TILE="1")
        assert torch.equal(y, _with(env, lambda: gu.resblock(sd, "rb", x, None, t, "bf16"))), env      # every tiling finalises alike


def test_proj_out_fused_into_the_attention_core_gives_the_same_bits(gu):
    """attn_fused_kernel<PROJ>: proj_out (+ bias, + the block's input, + the statistics of the result) as a third phase of the attention kernel ==
    the stand-alone GEMM (WDM_ATTN_FUSED=1): the same MFMA sequence per output and the same epilogue."""
    for C, B in ((512, 5), (256, 3), (128, 9)):
        shapes = {"norm.weight": (C,), "norm.bias": (C,)}
        for k in ("q", "k", "v", "proj_out"):
            shapes[k + ".weight"] = (C, C, 1, 1)
            shapes[k + ".bias"] = (C,)
        sd = gu.blk_sd("at", shapes)
        x = gu.seeded((B, C, 16, 16), 9)
        y = gu.attn(sd, "at", x, "bf16")
        y0 = _with({"WDM_ATTN_FUSED": "1"}, lambda: gu.attn(sd, "at", x, "bf16"))
        assert torch.isfinite(y).all() and torch.equal(y, y0), C
        assert rel_linf(y, gu.attn(sd, "at", x, "f32")) <= gu.TOL["bf16"]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_query_projection_inside_the_attention_core_gives_the_same_bits(gu, dtype):
    """attn_fused_kernel<PROJ, T, VTOK, QPROJ> (WDM_ATTN_FUSED=3, the default): the folded block's one remaining GEMM, q' = (Wk^T Wq) h + Wk^T bq, as phase 0 of the core on the
    workgroup's own 64 queries == the stand-alone GEMM (WDM_ATTN_FUSED=2): the same MFMA sequence per output, the same single rounding to 16 bits -- and one launch per AttnBlock."""
    from wavedm_amd import _lib
    for C, B in ((512, 5), (256, 3), (128, 9), (384, 2)):
        shapes = {"norm.weight": (C,), "norm.bias": (C,)}
        for k in ("q", "k", "v", "proj_out"):
            shapes[k + ".weight"] = (C, C, 1, 1)
            shapes[k + ".bias"] = (C,)
        sd = gu.blk_sd("at", shapes)
        x = gu.seeded((B, C, 16, 16), 19)

        def run():
            _lib.prof_enable(True)
            out = gu.attn(sd, "at", x, dtype)
            names = [e["kernel"].split("|")[0] for e in _lib.prof_report()]
            _lib.prof_enable(False)
            return out, names
        y, k3 = run()
        y2, k2 = _with({"WDM_ATTN_FUSED": "2"}, run)
        assert any(n.startswith("attn_fused_n256tq") for n in k3) and not any(n.startswith("gemm_1x1") or n.startswith("conv_1x1") for n in k3), k3
        assert any(n.startswith("gemm_1x1") or n.startswith("conv_1x1") for n in k2) and not any("n256tq" in n for n in k2), k2
        assert torch.isfinite(y).all() and torch.equal(y, y2), C
        assert torch.equal(y, _with({"WDM_ATTN_SM": "0"}, run)[0]), C          # slab-major or plain copies of the folded matrices: the same numbers
        assert rel_linf(y, gu.attn(sd, "at", x, "f32")) <= gu.TOL[dtype]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_8x8_attention_block_on_the_block_diagonal_fused_core(gu, dtype):
    """The AttnBlock of the 8 x 8 maps (64 tokens, C = 768 in the model): four images share one 256-row "image" of the fused core, a query block sees its own image's keys only
    (the other three key blocks' scores are -inf before the softmax: attn_fused_kernel.h, bdiag), on the folded operands -- q' GEMM, core, proj_out GEMM instead of seven launches.
    Any batch size: a ragged last group is skipped per query block, and an image's bits do not depend on the batch or on its place in a group."""
    from wavedm_amd import _lib
    for C in (768, 256):
        shapes = {"norm.weight": (C,), "norm.bias": (C,)}
        for k in ("q", "k", "v", "proj_out"):
            shapes[k + ".weight"] = (C, C, 1, 1)
            shapes[k + ".bias"] = (C,)
        sd = gu.blk_sd("at8", shapes)
        x = gu.seeded((6, C, 8, 8), 29)

        def run(xx=x):
            _lib.prof_enable(True)
            out = gu.attn(sd, "at8", xx, dtype)
            names = [e["kernel"].split("|")[0] for e in _lib.prof_report() for _ in range(int(e["launches"]))]
            _lib.prof_enable(False)
            return out, names
        y, k1 = run()
        y0, k0 = _with({"WDM_ATTN_FOLD": "0"}, run)
        ref = gu.attn(sd, "at8", x, "f32")
        assert any(n.startswith("attn_fused_n64x4t") for n in k1) and not any(n.startswith("attn_fused") for n in k0), (k1, k0)
        assert len(k1) < len(k0), (k1, k0)
        assert torch.isfinite(y).all() and rel_linf(y, ref) <= gu.TOL[dtype] and rel_linf(y0, ref) <= gu.TOL[dtype]
        assert torch.equal(y, run()[0])                                       # deterministic
        for sl in (slice(0, 1), slice(4, 6), slice(5, 6), slice(1, 4)):       # alone, as the ragged group, at another place of a group
            assert torch.equal(y[sl], run(x[sl].contiguous())[0]), (C, sl)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_fused_attention_core_is_bit_reproducible_at_every_width(gu, dtype):
    """Regression (round 5): the token-major-V core's transposing LDS reads sat behind an `if (fragment exists) read; else zero` -- a control-flow merge between an
    asynchronous read and its wait, where the compiler may copy a register the read has not filled yet.  C = 1024 in f16 (the only width using the fourth fragment without
    proj_out) went non-deterministic, ~1e-2 off, when an unrelated line changed the schedule.  The reads are branch-free now; every width must repeat bit for bit."""
    for C, B in ((1024, 3), (768, 2), (512, 5), (384, 3), (128, 2)):
        shapes = {"norm.weight": (C,), "norm.bias": (C,)}
        for k in ("q", "k", "v", "proj_out"):
            shapes[k + ".weight"] = (C, C, 1, 1)
            shapes[k + ".bias"] = (C,)
        sd = gu.blk_sd("atr", shapes)
        x = gu.seeded((B, C, 16, 16), 31)
        y = gu.attn(sd, "atr", x, dtype)
        for _ in range(4):
            assert torch.equal(y, gu.attn(sd, "atr", x, dtype)), C
        assert rel_linf(y, gu.attn(sd, "atr", x, "f32")) <= gu.TOL[dtype], C


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_attention_block_on_folded_operands(gu, dtype):
    """16-bit modes (blocks.hip: run_attn): softmax_j((Wq h_i + bq).(Wk h_j + bk)) = softmax_j((Wk^T Wq h_i + Wk^T bq).h_j) and proj_out(P.(Wv h + bv)) = (Wp Wv)(P.h) + Wp bv + bp, so
    the block runs ONE projection GEMM and the fused core with the normalised input as K and as (token-major, transposing-read) V.  WDM_ATTN_FOLD=0 keeps the k / v
    projections: both forms meet the mode's bound against exact fp32, and the folded one launches fewer kernels."""
    from wavedm_amd import _lib
    for C, B in ((512, 5), (256, 3), (128, 9), (768, 2), (1024, 1)):
        shapes = {"norm.weight": (C,), "norm.bias": (C,)}
        for k in ("q", "k", "v", "proj_out"):
            shapes[k + ".weight"] = (C, C, 1, 1)
            shapes[k + ".bias"] = (C,)
        sd = gu.blk_sd("at", shapes)
        x = gu.seeded((B, C, 16, 16), 9)

        def run():
            _lib.prof_enable(True)
            out = gu.attn(sd, "at", x, dtype)
            names = [e["kernel"].split("|")[0] for e in _lib.prof_report() for _ in range(int(e["launches"]))]
            _lib.prof_enable(False)
            return out, names
        (y, k1), (y0, k0) = run(), _with({"WDM_ATTN_FOLD": "0"}, run)
        ref = gu.attn(sd, "at", x, "f32")
        e1, e0 = rel_linf(y, ref), rel_linf(y0, ref)
        print(f"attn {dtype} C={C}: folded {e1:.2e}  k / v projections {e0:.2e}")
        assert torch.isfinite(y).all() and e1 <= gu.TOL[dtype] and e0 <= gu.TOL[dtype]
        assert any(n.startswith("attn_fused_n256t") for n in k1) and not any(n.startswith("attn_fused_n256t") for n in k0), (k1, k0)      # ("n256tq": with the query projection inside)
        assert len(k1) == len(k0) - (2 if C <= 512 else 1), (k1, k0)      # q' instead of q|k and V^T -- and q' inside the core where proj_out is (C <= 512)
        assert torch.equal(y, run()[0])


def test_in_tile_groupnorm_of_the_producing_conv_gives_the_bits_of_the_pass(gu):
    """gn_group.h: gn_out_tail -- conv1 of an 8x8 ResnetBlock writes act(norm2(h)) itself (two whole images x two whole groups per 128 x 48 tile of
    conv_dma8_kernel.h) instead of a gn_finalize_apply launch (WDM_GN_TILE=0).  Same reduction (gn_group_stats over the same float4 partials), same
    elementwise arithmetic (gn_apply_vec): the same bits; odd batches exercise the half-empty last tile."""
    from wavedm_amd import _lib
    for cin, cout, B, nin in ((768, 768, 3, False), (512, 768, 2, True), (768, 768, 64, False)):
        shapes = {"norm1.weight": (cin,), "norm1.bias": (cin,), "conv1.weight": (cout, cin, 3, 3), "conv1.bias": (cout,), "temb_proj.weight": (cout, 512),
                  "temb_proj.bias": (cout,), "norm2.weight": (cout,), "norm2.bias": (cout,), "conv2.weight": (cout, cout, 3, 3), "conv2.bias": (cout,)}
        if nin:
            shapes.update({"nin_shortcut.weight": (cout, cin, 1, 1), "nin_shortcut.bias": (cout,)})
        sd = gu.blk_sd("rb", shapes)
        x, t = gu.seeded((B, cin, 8, 8), 5), gu.seeded((B, 512), 6)

        def run():
            _lib.prof_enable(True)
            out = gu.resblock(sd, "rb", x, None, t, "bf16")
            names = [e["kernel"].split("|")[0] for e in _lib.prof_report() for _ in range(int(e["launches"]))]
            _lib.prof_enable(False)
            return out, names
        y, k1 = run()
        y0, k0 = _with({"WDM_GN_TILE": "0"}, run)
        assert torch.isfinite(y).all() and torch.equal(y, y0), (cin, cout, B)
        n1, n0 = sum("gn_finalize_apply" in k for k in k1), sum("gn_finalize_apply" in k for k in k0)
        assert (n1, n0) == (1, 2), (k1, k0)                                          # norm2's pass is gone, norm1's (no producer here) stays
        assert rel_linf(y, gu.resblock(sd, "rb", x, None, t, "f32")) <= gu.TOL["bf16"]


def test_in_tile_groupnorm_whole_unet_same_bits_fewer_launches():
    """The raindrop_wavelet UNet with and without the in-tile GroupNorm: conv2 of the ResnetBlocks in front of the AttnBlocks (16 x 16 maps: one image x
    128 columns per tile of conv_dma_kernel.h) and of the 8 x 8 ResnetBlocks also writes the consumer's norm.  Bit-identical output, 15 launches fewer."""
    from test_gpu_unet import build, seeded
    from wavedm_amd import procedural as P, _lib
    net = build(P.raindrop_wavelet_config(), "bf16")
    for B in (3, 64):
        x = seeded((B, 64, 64, 96), 5).cuda().to(torch.bfloat16).contiguous()
        ts = torch.tensor([500.0], device="cuda")

        def run():
            _lib.prof_enable(True)
            out = net.forward_nhwc(x, ts, torch.empty(B, 3, 64, 64, device="cuda")).clone()
            n = {}
            for e in _lib.prof_report():
                k = e["kernel"].split("|")[0]
                n[k] = n.get(k, 0) + int(e["launches"])
            _lib.prof_enable(False)
            return out, n
        y, n1 = _with({"WDM_GN_TILE": "1"}, run)
        y0, n0 = _with({"WDM_GN_TILE": "0"}, run)
        assert torch.isfinite(y).all() and torch.equal(y, y0), B
        assert n0["gn_finalize_apply_kernel"] - n1.get("gn_finalize_apply_kernel", 0) == 15, (n0, n1)
        # the default (2) also lets conv1 of the 16 x 16 ResnetBlocks normalise for conv2: another association of the statistics, same bound
        y2, _ = run()
        assert rel_linf(y2.float().cpu(), y.float().cpu()) <= 2e-2 and torch.isfinite(y2).all()


@pytest.mark.parametrize("cin,cout,B,H", [(128, 128, 5, 64), (256, 256, 3, 32), (128, 128, 2, 32), (96, 192, 2, 32), (512, 512, 2, 64)])
def test_downsample_on_the_lds_dma_kernel(gu, cin, cout, B, H):
    """conv_s2_kernel.h (the stride-2 conv as four stride-1 convs over the input's phases: 2x2 + 1x2 + 2x1 + 1x1 taps) against torch and against the
    register-staged kernel (WDM_CONV_DMA=0): another K order, the same bf16 bound; zero padding at the right / bottom edge only; ragged batches.  (The N tile --
    64 / 128 columns -- is a function of the layer's shape; both write the same bits: tools/conv_bench256.hip's kind of check was run on them in round 3.)"""
    from wavedm_amd import _lib
    w = gu.seeded((cout, cin, 3, 3), 41) / (cin * 9) ** 0.5
    b = gu.seeded((cout,), 42) * 0.1
    x = gu.seeded((B, cin, H, H), 43)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (0, 1, 0, 1)), w, b, stride=2)

    def run():
        _lib.prof_enable(True)
        out = gu.conv(w, b, 1, x, "bf16")
        names = [e["kernel"].split("|")[0] for e in _lib.prof_report()]
        _lib.prof_enable(False)
        return out, names
    y, k = run()
    assert any(n.startswith("convs2_") for n in k), k
    assert rel_linf(y, ref) <= gu.TOL["bf16"]
    y0, k0 = _with({"WDM_CONV_DMA": "0"}, run)
    assert any(n.startswith("conv_3x3s2") for n in k0) and rel_linf(y, y0) <= gu.TOL["bf16"]
    assert torch.equal(y, run()[0])
    # an edge the padding must not leak across: the last input row / column is read by tap rows / columns 0 and 1 only
    xz = x.clone(); xz[:, :, -1, :] = 0; xz[:, :, :, -1] = 0
    refz = torch.nn.functional.conv2d(torch.nn.functional.pad(xz, (0, 1, 0, 1)), w, b, stride=2)
    assert rel_linf(gu.conv(w, b, 1, xz, "bf16"), refz) <= gu.TOL["bf16"]


@pytest.mark.parametrize("cin,cout,B,H,cat", [(128, 128, 3, 32, 0), (256, 128, 2, 64, 128), (512, 512, 2, 16, 0), (160, 224, 2, 16, 64), (768, 768, 3, 8, 0),
                                                (512, 768, 5, 8, 0), (1280, 768, 2, 8, 512), (256, 320, 2, 8, 0)])
def test_f32x3_resblock_on_the_lds_dma_kernel(gu, cin, cout, B, H, cat):
    """conv_dmax3_kernel.h (f32x3 mode: hi / lo split once per staged element, in LDS) against the register-staged f32x3 kernel (WDM_CONV_DMA=0) and the exact
    fp32 path: a ResnetBlock with the GroupNorm prologue, temb, residual / 1x1 shortcut, optionally a concat input; Cout not a multiple of the N tile."""
    from wavedm_amd import _lib
    shapes = {"norm1.weight": (cin,), "norm1.bias": (cin,), "conv1.weight": (cout, cin, 3, 3), "conv1.bias": (cout,), "temb_proj.weight": (cout, 512),
              "temb_proj.bias": (cout,), "norm2.weight": (cout,), "norm2.bias": (cout,), "conv2.weight": (cout, cout, 3, 3), "conv2.bias": (cout,)}
    if cin != cout:
        shapes.update({"nin_shortcut.weight": (cout, cin, 1, 1), "nin_shortcut.bias": (cout,)})
    sd = gu.blk_sd("rb", shapes)
    x = gu.seeded((B, cin, H, H), 5)
    x0, x1 = (x[:, :cin - cat].contiguous(), x[:, cin - cat:].contiguous()) if cat else (x, None)
    t = gu.seeded((B, 512), 6)

    def run():
        _lib.prof_enable(True)
        out = gu.resblock(sd, "rb", x0, x1, t, "f32x3")
        names = [e["kernel"].split("|")[0] for e in _lib.prof_report()]
        _lib.prof_enable(False)
        return out, names
    y, k = run()
    y0, k0 = _with({"WDM_CONV_DMA": "0"}, run)
    tag = "convdma8x3" if H == 8 else "convdmax3"                                    # 8 x 8 maps: conv_dma8x3_kernel.h (48- or 64-column tiles)
    assert any(n.startswith(tag) for n in k) and not any(n.startswith("convdma") for n in k0), (k, k0)
    ref = gu.resblock(sd, "rb", x0, x1, t, "f32")
    e, e0 = rel_linf(y, ref), rel_linf(y0, ref)
    print(f"f32x3 resblock {cin}->{cout} @{H}: dma {e:.2e}  register-staged {e0:.2e}")
    assert e <= 2e-5 and e0 <= 2e-5 and rel_linf(y, y0) <= 2e-5
    assert torch.equal(y, run()[0])


@pytest.mark.parametrize("cin,cout,B,H", [(256, 256, 2, 32), (512, 512, 3, 16), (768, 768, 5, 8), (128, 192, 2, 16)])
def test_f32x3_upsample_in_sub_pixel_form(gu, cin, cout, B, H):
    """conv_up4x3_kernel.h (f32x3 mode: four 2x2-tap phase convs with pre-summed, pre-split weights) against the 9-tap register-staged form (WDM_UP4=0) and
    torch fp32: the pre-summing reassociates the taps, everything stays inside the f32x3 bound."""
    from wavedm_amd import _lib
    w = gu.seeded((cout, cin, 3, 3), 51) / (cin * 9) ** 0.5
    b = gu.seeded((cout,), 52) * 0.1
    x = gu.seeded((B, cin, H, H), 53)
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)

    def run():
        _lib.prof_enable(True)
        out = gu.conv(w, b, 2, x, "f32x3")
        names = [e["kernel"].split("|")[0] for e in _lib.prof_report()]
        _lib.prof_enable(False)
        return out, names
    y, k = run()
    y0, k0 = _with({"WDM_UP4": "0"}, run)
    assert any(n.startswith("convup4x3") for n in k) and any(n.startswith("conv_3x3ups") for n in k0), (k, k0)
    e, e0 = rel_linf(y, ref), rel_linf(y0, ref)
    print(f"f32x3 upsample {cin}->{cout} @{H}: sub-pixel {e:.2e}  9-tap {e0:.2e}")
    assert e <= 2e-5 and e0 <= 2e-5
    assert torch.equal(y, run()[0])


def test_f32x3_gemms_on_the_lds_dma_kernel(gu):
    """conv_gemmx3_kernel.h (f32x3 mode: both operands split hi / lo in LDS, two 16x16x32 MFMAs per product) against the register-staged f32x3 kernel
    (WDM_GEMM=0) and exact fp32: plain 1x1 convs (one and two K stages' worth of odd sizes, Cout not a multiple of the tile) and a whole AttnBlock
    (projections, Q.K^T and P.V with per-image operands, proj_out with residual)."""
    from wavedm_amd import _lib

    def run(f):
        _lib.prof_enable(True)
        out = f()
        names = [e["kernel"].split("|")[0] for e in _lib.prof_report() for _ in range(int(e["launches"]))]
        _lib.prof_enable(False)
        return out, names
    for cin, cout, B, H in ((512, 512, 3, 16), (160, 224, 2, 32), (1280, 512, 2, 16), (128, 128, 1, 64)):
        w = gu.seeded((cout, cin, 1, 1), 61) / cin ** 0.5
        b = gu.seeded((cout,), 62) * 0.1
        x = gu.seeded((B, cin, H, H), 63)
        ref = torch.nn.functional.conv2d(x, w, b)
        (y, k), (y0, k0) = run(lambda: gu.conv(w, b, 3, x, "f32x3")), _with({"WDM_GEMM": "0"}, lambda: run(lambda: gu.conv(w, b, 3, x, "f32x3")))
        assert any(n.startswith("gemmx3") for n in k) and not any(n.startswith("gemmx3") for n in k0), (k, k0)
        assert rel_linf(y, ref) <= 2e-5 and rel_linf(y0, ref) <= 2e-5, (cin, cout, rel_linf(y, ref))
    for C, B in ((512, 3), (256, 2)):
        shapes = {"norm.weight": (C,), "norm.bias": (C,)}
        for kk in ("q", "k", "v", "proj_out"):
            shapes[kk + ".weight"] = (C, C, 1, 1)
            shapes[kk + ".bias"] = (C,)
        sd = gu.blk_sd("at", shapes)
        x = gu.seeded((B, C, 16, 16), 9)
        (y, k), (y0, k0) = run(lambda: gu.attn(sd, "at", x, "f32x3")), _with({"WDM_GEMM": "0"}, lambda: run(lambda: gu.attn(sd, "at", x, "f32x3")))
        ref = gu.attn(sd, "at", x, "f32")
        assert sum(n.startswith("gemmx3") for n in k) >= 1 and not any(n.startswith("gemmx3") for n in k0), (k, k0)
        print(f"f32x3 attn C={C}: dma {rel_linf(y, ref):.2e}  register-staged {rel_linf(y0, ref):.2e}")
        assert rel_linf(y, ref) <= 2e-5 and rel_linf(y0, ref) <= 2e-5


@pytest.mark.parametrize("cin,cout,B,cat", [(512, 512, 5, 0), (768, 512, 2, 256), (256, 512, 3, 0)])
def test_conv1_normalises_for_conv2_on_16x16_maps(gu, cin, cout, B, cat):
    """WDM_GN_TILE=2 (default): on 16 x 16 maps conv1's tile is the whole image, so its epilogue also writes act(norm2(h)) (gn_group.h: gn_out_tail) and conv2
    runs WITHOUT the GroupNorm+SiLU prologue its four N tiles would each repeat.  Against WDM_GN_TILE=1 (prologue, finalised in place or by gn_finalize):
    the statistics are summed in another association -- one-ulp differences on a small fraction of the outputs --, both inside the bf16 bound."""
    from wavedm_amd import _lib
    shapes = {"norm1.weight": (cin,), "norm1.bias": (cin,), "conv1.weight": (cout, cin, 3, 3), "conv1.bias": (cout,), "temb_proj.weight": (cout, 512),
              "temb_proj.bias": (cout,), "norm2.weight": (cout,), "norm2.bias": (cout,), "conv2.weight": (cout, cout, 3, 3), "conv2.bias": (cout,)}
    if cin != cout:
        shapes.update({"nin_shortcut.weight": (cout, cin, 1, 1), "nin_shortcut.bias": (cout,)})
    sd = gu.blk_sd("rb", shapes)
    x = gu.seeded((B, cin, 16, 16), 5)
    x0, x1 = (x[:, :cin - cat].contiguous(), x[:, cin - cat:].contiguous()) if cat else (x, None)
    t = gu.seeded((B, 512), 6)

    def run():
        _lib.prof_enable(True)
        out = gu.resblock(sd, "rb", x0, x1, t, "bf16")
        names = [e["kernel"] for e in _lib.prof_report()]
        _lib.prof_enable(False)
        return out, names
    y, k = run()
    y1, k1 = _with({"WDM_GN_TILE": "1"}, run)
    conv2 = [n for n in k if f"{cout}->{cout}" in n and n.startswith("convdma")]
    assert any(" gn" not in n.split("|")[1] for n in conv2), k                      # conv2 ran without the prologue ...
    assert all(" gn" in n.split("|")[1] for n in k1 if n.startswith("convdma")), k1   # ... and with it under WDM_GN_TILE=1
    ref = gu.resblock(sd, "rb", x0, x1, t, "f32")
    assert rel_linf(y, y1) <= 8e-3 and float(((y - y1).abs() > 0).float().mean()) <= 2e-2
    assert rel_linf(y, ref) <= gu.TOL["bf16"] and rel_linf(y1, ref) <= gu.TOL["bf16"]
    assert torch.equal(y, run()[0])
    assert torch.equal(y[1:2], gu.resblock(sd, "rb", x0[1:2].contiguous(), x1[1:2].contiguous() if cat else None, t[1:2].contiguous(), "bf16"))     # batch-independent


def test_in_tile_groupnorm_in_the_f32x3_mode(gu):
    """The f32x3 LDS-DMA kernels keep their epilogue tiles too (gn_out_tail<float>): conv1 -> norm2 of an 8x8 ResnetBlock bit-identical to the gn_finalize_apply
    pass (WDM_GN_TILE=0), and on a 16x16 map conv2 without its prologue (WDM_GN_TILE=2) inside the f32x3 bound."""
    from wavedm_amd import _lib
    for c, H, B in ((768, 8, 3), (512, 16, 2)):
        shapes = {"norm1.weight": (c,), "norm1.bias": (c,), "conv1.weight": (c, c, 3, 3), "conv1.bias": (c,), "temb_proj.weight": (c, 512),
                  "temb_proj.bias": (c,), "norm2.weight": (c,), "norm2.bias": (c,), "conv2.weight": (c, c, 3, 3), "conv2.bias": (c,)}
        sd = gu.blk_sd("rb", shapes)
        x, t = gu.seeded((B, c, H, H), 5), gu.seeded((B, 512), 6)

        def run():
            _lib.prof_enable(True)
            out = gu.resblock(sd, "rb", x, None, t, "f32x3")
            names = [e["kernel"].split("|")[0] for e in _lib.prof_report() for _ in range(int(e["launches"]))]
            _lib.prof_enable(False)
            return out, names
        y, k = run()
        y0, k0 = _with({"WDM_GN_TILE": "0"}, run)
        ref = gu.resblock(sd, "rb", x, None, t, "f32")
        if H == 8:
            assert torch.equal(y, y0) and sum("gn_finalize_apply" in n for n in k) + 1 == sum("gn_finalize_apply" in n for n in k0), (k, k0)
        assert rel_linf(y, ref) <= 2e-5 and rel_linf(y0, ref) <= 2e-5 and rel_linf(y, y0) <= 2e-5


@pytest.mark.parametrize("cin,cout,B,H", [(128, 128, 3, 64), (256, 256, 2, 32), (96, 192, 2, 32)])
def test_f32x3_downsample_on_the_lds_dma_kernel(gu, cin, cout, B, H):
    """conv_s2x3_kernel.h (f32x3 mode: the four-phase Downsample kernel with halo and weights split in LDS) against torch fp32 and the register-staged f32x3 form."""
    from wavedm_amd import _lib
    w = gu.seeded((cout, cin, 3, 3), 41) / (cin * 9) ** 0.5
    b = gu.seeded((cout,), 42) * 0.1
    x = gu.seeded((B, cin, H, H), 43)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (0, 1, 0, 1)), w, b, stride=2)

    def run():
        _lib.prof_enable(True)
        out = gu.conv(w, b, 1, x, "f32x3")
        names = [e["kernel"].split("|")[0] for e in _lib.prof_report()]
        _lib.prof_enable(False)
        return out, names
    y, k = run()
    y0, k0 = _with({"WDM_CONV_DMA": "0"}, run)
    assert any(n.startswith("convs2x3") for n in k) and any(n.startswith("conv_3x3s2") for n in k0), (k, k0)
    assert rel_linf(y, ref) <= 2e-5 and rel_linf(y0, ref) <= 2e-5
    assert torch.equal(y, run()[0])


def test_groupnorm_finalised_by_the_producers_last_workgroups_gives_the_bits_of_gn_finalize(gu):
    """gn_arrive.h: every conv whose consumer would otherwise need a gn_finalize launch (the 64 x 64 maps, every channel-concat input of the up path, norm_out --
    17 per UNet call) can finalise that norm itself (WDM_GN_INLINE=2; opt-in: exact but measured slower, gn_arrive.h): its workgroups announce their tiles on a per-image counter, the last one runs gn_finalize_kernel's reduction over
    the image's partials (read past the caches) and writes the consumer's scale / shift rows.  Same instruction sequence over the same partials in the same order,
    whoever arrives last: the full-width UNet's output must equal WDM_GN_INLINE=1 (gn_finalize launches) BIT FOR BIT -- at several batch sizes (tile walks and
    last arrivers differ), repeatedly (arrival order differs from run to run), and no gn_finalize launch may be left."""
    import wavedm_amd
    from wavedm_amd import _lib, procedural as P
    cfg = P.raindrop_wavelet_config()
    net = wavedm_amd.DiffusionUNet(cfg, dtype="bf16")
    net.load_state_dict(P.procedural_state_dict(cfg), strict=True)
    net = net.cuda()
    t = torch.tensor([470.0])
    for B in (3, 64, 17):
        x = gu.seeded((B, 96, 64, 64), 100 + B).cuda()

        def run():
            _lib.prof_enable(True)
            y = net(x, t)
            names = [e["kernel"].split("|")[0] for e in _lib.prof_report() for _ in range(int(e["launches"]))]
            _lib.prof_enable(False)
            return y, names
        y, k = _with({"WDM_GN_INLINE": "2"}, run)
        y1, k1 = _with({"WDM_GN_INLINE": "1"}, run)
        n_fin, n_fin1 = sum(n == "gn_finalize_kernel" for n in k), sum(n == "gn_finalize_kernel" for n in k1)
        print(f"B={B}: gn_finalize launches {n_fin1} -> {n_fin}")
        assert n_fin1 == 17 and n_fin == 0, (n_fin1, n_fin)
        assert torch.isfinite(y).all() and torch.equal(y, y1), B
        for _ in range(3):
            assert torch.equal(y, _with({"WDM_GN_INLINE": "2"}, lambda: net(x, t)))
