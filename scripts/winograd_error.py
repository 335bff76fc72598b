#!/usr/bin/env python3
"""Precision of Winograd F(2x2, 3x3) on 16-bit operands against the direct 3x3 conv on the same operands (the part that can be measured without the
kernel): one layer of the 64 x 64 level (128 -> 128), activations ~ the GroupNorm + SiLU output of a normalised tensor, weights ~ the procedural initialisation's scale.
Both forms round their MULTIPLIER operands to the 16-bit type (Winograd: the transformed input tiles B^T d B and the transformed filters G g G^T, which is what an MFMA kernel
would feed the matrix pipe) and accumulate in fp32/fp64; the reference is the exact fp64 conv of the unrounded operands.   usage: python scripts/winograd_error.py"""
import torch

torch.manual_seed(3)
Cin, Cout, H = 128, 128, 32
x = torch.nn.functional.silu(torch.randn(2, Cin, H, H, dtype=torch.float64))
w = torch.randn(Cout, Cin, 3, 3, dtype=torch.float64) / (Cin * 9) ** 0.5
ref = torch.nn.functional.conv2d(x, w, padding=1)

Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def rnd(t, dt):
    return t.to(dt).to(torch.float64) if dt is not None else t


def winograd(x, w, dt):
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    n, c, hp, wp = xp.shape
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                        # [n, c, th, tw, 4, 4]
    V = rnd(torch.einsum("ij,nctujk,lk->nctuil", Bt, tiles, Bt), dt)   # B^T d B, rounded as an MFMA operand
    U = rnd(torch.einsum("ij,ocjk,lk->ocil", G, w, G), dt)             # G g G^T
    M = torch.einsum("nctuil,ocil->notuil", V, U)                      # 16 GEMMs over the channels, exact accumulation
    Y = torch.einsum("ij,notujk,lk->notuil", At, M, At)                # A^T m A: [n, o, th, tw, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], H, H)


def rel(a):
    return float((a - ref).abs().max() / ref.abs().max())


print(f"layer {Cin}->{Cout}, {H}x{H}, exact fp64 reference")
assert rel(winograd(x, w, None)) < 1e-12
for name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
    direct = torch.nn.functional.conv2d(rnd(x, dt), rnd(w, dt), padding=1)
    e_d, e_w = rel(direct), rel(winograd(x, w, dt))
    print(f"{name}: direct conv on rounded operands {e_d:.2e}   Winograd F(2x2,3x3) with rounded transformed operands {e_w:.2e}   ratio {e_w / e_d:.2f}")
