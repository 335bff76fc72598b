#!/bin/bash
# K-split 8 x 8 kernel (conv_dma8k_kernel.h) against the shipped 128 x 48 tile, stand-alone: warm, cold weights, with a fused shortcut, and the ablations
# usage (GPU box): bash scripts/dma8k_bench.sh > gpurun_out/dma8k_bench.log 2>&1
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="hipcc --offload-arch=gfx950 -O3 -std=c++17 -I wavedm_amd/csrc -I include -I tools tools/dma8k_bench.hip"
$B -o /tmp/d8k || exit 1
for shape in "64 768 768" "64 1536 768" "64 1280 768" "8 768 768"; do
  echo "== $shape warm"; /tmp/d8k $shape
  echo "== $shape cold"; COLD=24 /tmp/d8k $shape
done
echo "== 64 768 768 + shortcut over 768 channels"; SC=768 /tmp/d8k 64 768 768
echo "== 64 1536 768 + shortcut over 1536 channels, cold"; SC=1536 COLD=24 /tmp/d8k 64 1536 768
for gnk in 1 2 8; do echo "== GNK=$gnk"; GNK=$gnk /tmp/d8k 64 768 768 | grep dma8k; done
for m in 18 12 30; do
  $B -DWDM_D8KABL=$m -DWDM_D8ABL=$m -o /tmp/d8k_$m && { echo "== ablation mask $m (2|16: no MFMAs / fragment reads, 4|8: no DMA)"; /tmp/d8k_$m 64 768 768 | grep -v outputs; }
done
