#!/bin/bash
# The board's MFMA roofs and the shipped tiles stand-alone in ONE call on ONE box (the boxes of the pool differ by ~10 % in what their power limit gives):
# usage (GPU box): bash scripts/roof_same_box.sh > gpurun_out/roof_same_box.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in reg lds; do HOLD=2 MODE=$m tools/abl_mfma_power; done
for shape in "64 32 768 256 1" "64 32 512 256 1" "64 32 256 256 1" "64 16 512 512 0" "64 64 384 128 1" "64 64 128 128 1"; do
  echo "== B H Cin Cout pro = $shape"; NOREF=1 ROUNDS=2 tools/abl_conv_bench256 $shape 2>&1 | grep -E "t256x256 |t256x128P |t512x128 " | head -4
done
for m in reg lds; do HOLD=2 MODE=$m tools/abl_mfma_power; done
