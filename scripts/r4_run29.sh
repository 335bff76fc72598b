cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_bn256.py tests/test_gpu_switches.py -x -q -m gpu 2>&1 | tail -4
WDM_BN256=0 timeout 600 python bench.py --dtype f32x3 --steps 1 --warmup 1 --ddim-steps 10 --no-extras --no-cpu-baseline 2> gpurun_out/x3_a.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BN256=0', d['value'])"
timeout 600 python bench.py --dtype f32x3 --steps 1 --warmup 1 --ddim-steps 10 --no-extras --no-cpu-baseline 2> gpurun_out/x3_b.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'])"
grep '^\[shape\].*64x64' gpurun_out/x3_a.log | cut -c9-140 | sort | head -8
grep '^\[shape\].*64x64' gpurun_out/x3_b.log | cut -c9-140 | sort | head -8
