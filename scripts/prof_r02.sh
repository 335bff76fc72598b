# Round-2 profiling recipe (same as scripts/prof_r01.sh; the informational legs of bench.py are skipped under the tracer) (run on the GPU box via gpurun): kernel trace + stats of the default bench command, then
# separate PMC passes (restricted to the conv / GEMM kernels: unrestricted, this rocprofv3 build segfaults at the first elementwise launch since the library grew)
# separate PMC passes (counters never combined with tracing, per the pool's rules) on a 5-step version of the workload.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/raw
cd $R
if [ -z "$PMC_ONLY" ]; then
rocprofv3 --kernel-trace --stats -M --output-format csv -d /tmp/prof_stats -- python bench.py --no-extras > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
cp /tmp/prof_stats/*/*kernel_stats.csv gpurun_out/raw/kernel_stats.csv
gzip -c /tmp/prof_stats/*/*kernel_trace.csv > gpurun_out/raw/kernel_trace.csv.gz
fi
PMC_CMD="python bench.py --ddim-steps 5 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline"
for C in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $C --kernel-include-regex "conv|gemm" -M --output-format csv -d /tmp/prof_$C -- $PMC_CMD > /dev/null 2> gpurun_out/pmc_$C.err
  gzip -c /tmp/prof_$C/*/*counter_collection.csv > gpurun_out/raw/pmc_$C.csv.gz
done
rocprofv3 --kernel-include-regex "conv|gemm" --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -M --output-format csv -d /tmp/prof_sq -- $PMC_CMD > /dev/null 2> gpurun_out/pmc_sq.err
gzip -c /tmp/prof_sq/*/*counter_collection.csv > gpurun_out/raw/pmc_sq.csv.gz
ls -la gpurun_out/raw
