# Everything profiles/r04_* is made of, in one gpurun call (run on the GPU box):
#   scripts/prof_r04.sh   kernel stats bf16 + f32x3, whole-path PMC passes          -> gpurun_out/kernel_stats*.md, gpurun_out/raw/unet_pmc_*.csv.gz
#   scripts/prof_train.sh kernel stats of the training step at 64 samples            -> gpurun_out/train_stats.md
#   bench.py              the default run, as the driver launches it                 -> gpurun_out/r04_bench.json
# afterwards, here: copy the three .md files and the JSON line to profiles/r04_*, run `python scripts/traffic_from_unet.py r04`.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 bash scripts/prof_r04.sh > gpurun_out/prof_r04.out 2>&1
timeout 300 bash scripts/prof_train.sh --batch 64 > gpurun_out/prof_train.out 2>&1
cd $R
timeout 900 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.log
tail -5 gpurun_out/prof_r04.out; tail -3 gpurun_out/prof_train.out; cut -c1-400 gpurun_out/r04_bench.json
