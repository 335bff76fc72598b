mkdir -p gpurun_out
( python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/r4_pytest3.log 2>&1
cat gpurun_out/r4_pytest3.log
python scripts/switch_fuzz.py 24 > gpurun_out/r4_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r4_fuzz.log
bash scripts/ab.sh "WAVEDM_LIB=tools/abl_lib_r3.so" "WDM_X=1" "WAVEDM_LIB=tools/abl_lib_r3.so" "WDM_X=1"
