#!/bin/bash
# rocprofv3 kernel statistics of the training step (scripts/train_bench.py): gpurun_out/train_stats.md  (copy to profiles/ to keep)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/train_prof
rocprofv3 --kernel-trace --stats -M --output-format csv -d $R/gpurun_out/train_prof -- python $R/scripts/train_bench.py --iters 5 "$@" > $R/gpurun_out/train_prof.log 2>&1
tail -1 $R/gpurun_out/train_prof.log
python $R/scripts/summarize_rocprof.py stats $R/gpurun_out/train_prof $R/gpurun_out/train_stats.md
rm -rf $R/gpurun_out/train_prof
head -34 $R/gpurun_out/train_stats.md
