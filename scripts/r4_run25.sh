cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python scripts/train_bench.py --batch 64 --iters 10 --prof 2> gpurun_out/train_shapes.log | tail -1
grep '^\[shape\]' gpurun_out/train_shapes.log | head -40 | cut -c1-170
