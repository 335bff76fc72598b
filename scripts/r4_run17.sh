# 512 x 128 tile (conv_dma256_kernel<8,1,4,8,32>) against the 256 x 128 tile: bits and time
cd ${GRAFT_REPO_ROOT:-/root/repo}
ONLY=t256x128P,t512x128,t512x128F,t256x256 ROUNDS=4 IT=10 timeout 600 tools/abl_conv_bench256 > gpurun_out/r4_t512.log 2>&1
cat gpurun_out/r4_t512.log | cut -c1-330
