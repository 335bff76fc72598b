# 512 x 128 tile: micro-benchmark (bits + time), GPU tests touching the tilings, whole-model A/B against the previous build
cd ${GRAFT_REPO_ROOT:-/root/repo}
ONLY=t256x128P,t512x128,t512x128F,t512x128S ROUNDS=4 IT=10 timeout 600 tools/abl_conv_bench256 > gpurun_out/r4_t512b.log 2>&1
grep '64x64' gpurun_out/r4_t512b.log | cut -c1-330; tail -1 gpurun_out/r4_t512b.log
timeout 900 python -m pytest tests/test_gpu_bn256.py tests/test_gpu_switches.py tests/test_gpu_unet.py -x -q -m gpu 2>&1 | tail -4
bash scripts/ab.sh WAVEDM_LIB=tools/abl_lib_prev.so WDM_X=1 WAVEDM_LIB=tools/abl_lib_prev.so WDM_X=1 2>&1 | grep '^=='
