#!/bin/bash
# XCD order of the 8 x 8 kernel's workgroups on the current tile (128 x 48, four waves, slab-major weights): N-tile groups per XCD = 1 / 2 / 4 / 8
# (conv_kernel.h: conv_decode_tile), warm and with the weights cold (COLD=24: 24 copies of the tensor in rotation, as inside the UNet).
# usage (GPU box): bash scripts/dma8_xcd_sweep.sh > gpurun_out/dma8_xcd_sweep.log
set -e
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBN8=48 -DNI8=2 -I wavedm_amd/csrc -I include tools/dma8_ablate.hip -o /tmp/abl_dma8
for shape in "64 768 768" "64 1536 768"; do
  for gn in 1 2 4 8; do
    echo "== B Cin Cout = $shape  GN=$gn warm"; SM=1 GN=$gn /tmp/abl_dma8 $shape | tail -2
    echo "== B Cin Cout = $shape  GN=$gn cold"; SM=1 GN=$gn COLD=24 /tmp/abl_dma8 $shape | tail -2
  done
done
