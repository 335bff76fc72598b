cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python -m pytest tests/test_gpu_bn256.py -x -q -m gpu 2>&1 | tail -4
bash scripts/ab.sh WAVEDM_LIB=tools/abl_lib_prev.so WDM_X=1 WAVEDM_LIB=tools/abl_lib_prev.so WDM_X=1 2>&1 | grep '^=='
grep '^\[shape\].*64x64' gpurun_out/ab_3.log | cut -c1-150
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
