# HBM traffic of the dominant kernel (convdma_3x3s1_t16x16x1_bn128w8_bf16) per layer shape of the raindrop_wavelet UNet at batch 64:
# separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with tracing) over tools/dma_ablate.hip, the stand-alone launcher
# of exactly that kernel.  (Round 2: the PMC passes over the whole `bench.py` process segfault inside this rocprofv3 build at the first
# elementwise launch -- gpurun_out/pmc_*.err -- so the per-shape launcher is profiled instead and weighted by the model's launch mix.)
# usage (on the GPU box): bash scripts/pmc_dma_shapes.sh   -> gpurun_out/pmc_dma_shapes.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_dma_shapes.csv
echo "H,Cin,Cout,pro,count_per_unet_call,counter,dispatches,mean_kib" > $OUT
#      H  Cin Cout pro count
SHAPES="64 128 128 1 7
32 256 256 1 6
16 512 512 1 6
64 256 128 1 2
64 384 128 1 1
32 768 256 1 1
16 1280 512 1 1
32 512 256 1 1
16 1024 512 1 1
32 384 256 1 1
16 768 512 1 1
64 96 128 0 1
32 128 256 1 1
16 256 512 1 1"
echo "$SHAPES" | while read H CIN COUT PRO CNT; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_pmc
    SM=1 timeout 90 rocprofv3 --pmc $C -M --output-format csv -d /tmp/p_pmc -- $R/tools/abl_dma_0 64 $H $CIN $COUT $PRO > /dev/null 2>&1
    python3 - "$C" "$H,$CIN,$COUT,$PRO,$CNT" >> $OUT <<'P'
import glob, csv, sys
c, key = sys.argv[1], sys.argv[2]
vals = []
for f in glob.glob("/tmp/p_pmc/*/*counter_collection.csv"):
    vals += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
print(f"{key},{c},{len(vals)},{sum(vals) / max(len(vals), 1):.1f}")
P
  done
done
cat $OUT
