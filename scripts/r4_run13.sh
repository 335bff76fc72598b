mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_switches.py -x -q -k "producers_last" -s 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -5
bash scripts/ab.sh "WDM_GN_INLINE=1" "WDM_GN_INLINE=2" "WDM_GN_INLINE=1" "WDM_GN_INLINE=2"
