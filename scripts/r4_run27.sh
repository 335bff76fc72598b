cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python scripts/train_bench.py --batch 64 --iters 10 2>&1 | tail -1
WDM_WGRAD_BG=64 timeout 300 python scripts/train_bench.py --batch 64 --iters 10 2>&1 | tail -1
