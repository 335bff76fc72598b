mkdir -p gpurun_out
( python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r4_pytest5.log 2>&1
cat gpurun_out/r4_pytest5.log
bash scripts/ab.sh "WAVEDM_LIB=tools/abl_lib_r3.so" "WDM_X=1" "WAVEDM_LIB=tools/abl_lib_r3.so" "WDM_X=1" "WAVEDM_LIB=tools/abl_lib_r3.so" "WDM_X=1"
