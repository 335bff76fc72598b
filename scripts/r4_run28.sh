cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python scripts/train_bench.py --batch 8 --iters 10 2>&1 | tail -1
bash scripts/prof_train.sh --batch 64 > gpurun_out/prof_train.out 2>&1; tail -3 gpurun_out/prof_train.out | cut -c1-200
