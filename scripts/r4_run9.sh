mkdir -p gpurun_out
ONLY=persist1,persistF IT=20 tools/abl_conv_bench256 2>&1 | cut -c1-330 | tail -24 > gpurun_out/r4_chunk_bench.log; tail -3 gpurun_out/r4_chunk_bench.log
( python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r4_pytest4.log 2>&1
cat gpurun_out/r4_pytest4.log
bash scripts/ab.sh "WAVEDM_LIB=tools/abl_lib_r3.so" "WDM_X=1" "WDM_GN_INLINE=0" "WAVEDM_LIB=tools/abl_lib_r3.so" "WDM_X=1" "WDM_GN_INLINE=0"
