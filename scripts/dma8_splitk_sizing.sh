#!/bin/bash
# Round 6: what ONE launch of a cross-workgroup split-K design for the 8x8 level would cost, measured with the kernel that already exists for
# its workgroup shape -- the 256-pixel x 256-column tile of conv_dma256_kernel.h (8 waves of 64 x 128, 0.31 fragment reads per MFMA), no GroupNorm transform, K = one slice.
#   768 -> 768 @ 8x8, batch 64: M = 4096 rows = 16 tiles x 3 column tiles = 48; split 4 -> 192 workgroups of 6 slabs; split 5 (5,5,5,5,4) -> 240 of <= 5 slabs
#   stand-in with the same rows, columns and K per workgroup: 16x16 maps (one 256-pixel tile per image), Cout 768, Cin = 32 x slabs, batch = workgroups / 3
# plus the layers as they run today (tools/abl_dma8_n2_0 of scripts/d8_ab.sh when built) -- the split-K launch still needs a reduction pass over 4 (5) fp32 partial tiles.
export ROUNDS=3 IT=20 ONLY=t256x256
echo "== 192 workgroups x 6 slabs (split 4 of 768->768)";   tools/abl_conv_bench256 64 16 192 768 0 | tail -4
echo "== 255 workgroups x 5 slabs (split 5 of 768->768)";   tools/abl_conv_bench256 85 16 160 768 0 | tail -4
echo "== 192 workgroups x 12 slabs (split 4 of 1536->768)"; tools/abl_conv_bench256 64 16 384 768 0 | tail -4
echo "== 256 workgroups x 8 slabs, Cout 256 (the 32x32 layer 256->256 without transform, for scale)"; tools/abl_conv_bench256 64 32 256 256 0 | tail -4
