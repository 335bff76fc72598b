# round-4 profile refresh: kernel stats (bf16, f32x3), whole-path PMC, training step, full bench with extras
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 bash scripts/prof_r04.sh > gpurun_out/prof_r04.out 2>&1
timeout 300 bash scripts/prof_train.sh --batch 64 > gpurun_out/prof_train.out 2>&1
cd $R
timeout 900 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.log
tail -5 gpurun_out/prof_r04.out; tail -3 gpurun_out/prof_train.out; cut -c1-600 gpurun_out/r04_bench.json
