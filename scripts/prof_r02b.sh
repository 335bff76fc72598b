# Round-2 profile of the final build (run on the GPU box via gpurun): kernel trace + stats of the default bench command (no PMC over the python
# process: that crashes inside this rocprofv3 build, see scripts/prof_r02.sh), then FETCH_SIZE / WRITE_SIZE passes over the stand-alone launchers of
# the two LDS-DMA conv kernels (scripts/pmc_dma_shapes.sh; tools/dma8_ablate.hip for the 8x8 kernel).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/raw
cd $R
timeout 900 rocprofv3 --kernel-trace --stats -M --output-format csv -d /tmp/prof_stats -- python bench.py --no-extras > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
cp /tmp/prof_stats/*/*kernel_stats.csv gpurun_out/raw/kernel_stats.csv
gzip -c /tmp/prof_stats/*/*kernel_trace.csv > gpurun_out/raw/kernel_trace.csv.gz
bash scripts/pmc_dma_shapes.sh > /dev/null 2>&1
OUT=$R/gpurun_out/pmc_dma8_shapes.csv
echo "Cin,Cout,counter,dispatches,mean_kib" > $OUT
for SH in "768 768" "1536 768" "1280 768" "512 768"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_pmc
    SM=1 GN=4 timeout 90 rocprofv3 --pmc $C -M --output-format csv -d /tmp/p_pmc -- $R/tools/abl_dma8_0 64 $SH > /dev/null 2>&1
    python3 - "$C" "$SH" >> $OUT <<'P'
import glob, csv, sys
c, key = sys.argv[1], sys.argv[2].replace(" ", ",")
vals = []
for f in glob.glob("/tmp/p_pmc/*/*counter_collection.csv"):
    vals += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
print(f"{key},{c},{len(vals)},{sum(vals) / max(len(vals), 1):.1f}")
P
  done
done
cat $OUT
tail -2 gpurun_out/bench_prof.json | cut -c1-600
