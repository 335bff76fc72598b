cd ${GRAFT_REPO_ROOT:-/root/repo}
ONLY=t256x128P,t512x128,t512x128F,t512x128FS ROUNDS=4 IT=10 timeout 600 tools/abl_conv_bench256 > gpurun_out/r4_t512c.log 2>&1
grep '64x64\|identical\|MISM' gpurun_out/r4_t512c.log | cut -c1-330
timeout 1200 python -m pytest tests/test_gpu_bn256.py -x -q -m gpu 2>&1 | tail -2
bash scripts/ab.sh WAVEDM_LIB=tools/abl_lib_prev.so WDM_X=1 WAVEDM_LIB=tools/abl_lib_prev.so WDM_X=1 2>&1 | grep '^=='
grep '^\[shape\].*64x64' gpurun_out/ab_3.log | cut -c9-150 | sort -k1,1
