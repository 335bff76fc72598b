cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/ab.sh WAVEDM_LIB=tools/abl_lib_prev.so WDM_X=1 WAVEDM_LIB=tools/abl_lib_prev.so WDM_X=1 WAVEDM_LIB=tools/abl_lib_prev.so WDM_X=1 2>&1 | grep '^=='
grep -c gn_finalize gpurun_out/ab_1.log
