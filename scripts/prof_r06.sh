# Round-6 profiling recipe (run on the GPU box via gpurun):
#   1. rocprofv3 kernel trace + stats of the default bench command (bf16 headline; informational legs skipped under the tracer) -> gpurun_out/kernel_stats.md
#   2. the same for the tolerance-conformant modes:  bench.py --dtype f16 / f32x3  -> gpurun_out/kernel_stats_f16.md / _f32x3.md  (what parity_mode.roofline quotes)
#   3. whole-path PMC passes on the Python-free driver (scripts/pmc_unet.sh): FETCH_SIZE / WRITE_SIZE / GRBM / SQ, one counter set per pass, never with tracing
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/raw
cd $R
rm -rf /tmp/prof_stats /tmp/prof_x3
rocprofv3 --kernel-trace --stats -M --output-format csv -d /tmp/prof_stats -- python bench.py --no-extras > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
cp /tmp/prof_stats/*/*kernel_stats.csv gpurun_out/raw/kernel_stats.csv
gzip -c /tmp/prof_stats/*/*kernel_trace.csv > gpurun_out/raw/kernel_trace.csv.gz
python scripts/summarize_rocprof.py stats /tmp/prof_stats gpurun_out/kernel_stats.md
head -24 gpurun_out/kernel_stats.md | cut -c1-150
for dt in f16 f32x3; do
  rm -rf /tmp/prof_$dt
  rocprofv3 --kernel-trace --stats -M --output-format csv -d /tmp/prof_$dt -- python bench.py --dtype $dt --no-extras --no-cpu-baseline > gpurun_out/bench_prof_$dt.json 2> gpurun_out/bench_prof_$dt.err
  python scripts/summarize_rocprof.py stats /tmp/prof_$dt gpurun_out/kernel_stats_$dt.md
  head -14 gpurun_out/kernel_stats_$dt.md | cut -c1-150
done
bash scripts/gap_unet.sh 2>&1 | tail -12
bash scripts/pmc_unet.sh
python scripts/traffic_from_unet.py r06 2>&1 | tail -30
