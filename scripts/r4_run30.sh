cd ${GRAFT_REPO_ROOT:-/root/repo}
WAVEDM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --ddim-steps 10 > gpurun_out/b2.out 2> gpurun_out/b2.err; echo rc=$? lines=$(wc -l < gpurun_out/b2.out); cut -c1-120 gpurun_out/b2.out
timeout 900 python bench.py --steps 1 --warmup 1 --ddim-steps 10 > gpurun_out/b1.out 2> gpurun_out/b1.err; echo rc=$? lines=$(wc -l < gpurun_out/b1.out); cut -c1-120 gpurun_out/b1.out
python bench.py --gpus 3 > gpurun_out/b3.out 2> gpurun_out/b3.err; echo rc=$? lines=$(wc -l < gpurun_out/b3.out); cut -c1-200 gpurun_out/b3.out
