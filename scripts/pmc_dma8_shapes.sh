cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_dma8_shapes.csv
echo "Cin,Cout,counter,dispatches,mean_kib" > $OUT
for g in 1 4; do for i in 1 2; do SM=1 GN=$g timeout 30 $R/tools/abl_dma8_0 64 768 768; done; done
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
