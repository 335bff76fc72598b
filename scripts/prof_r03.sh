# Round-3 profiling recipe (run on the GPU box via gpurun): kernel trace + stats of the default bench command (informational legs skipped under the tracer).
# PMC passes are taken on the stand-alone launchers (scripts/pmc_*.sh, tools/unet_run.hip): rocprofv3 --pmc crashes on the python process in this image.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/raw
cd $R
rocprofv3 --kernel-trace --stats -M --output-format csv -d /tmp/prof_stats -- python bench.py --no-extras ${BENCH_ARGS} > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
cp /tmp/prof_stats/*/*kernel_stats.csv gpurun_out/raw/kernel_stats.csv
gzip -c /tmp/prof_stats/*/*kernel_trace.csv > gpurun_out/raw/kernel_trace.csv.gz
python scripts/summarize_rocprof.py stats /tmp/prof_stats gpurun_out/kernel_stats.md
head -40 gpurun_out/kernel_stats.md
