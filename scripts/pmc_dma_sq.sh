# SQ counters of the dominant kernel (stand-alone launcher tools/dma_ablate.hip) on three layer shapes at batch 64: matrix-pipe busy cycles,
# wave cycles and their wait / issue split, LDS bank conflicts.  Separate rocprofv3 --pmc pass (never combined with tracing).
# usage (on the GPU box): bash scripts/pmc_dma_sq.sh -> gpurun_out/pmc_dma_sq.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_dma_sq.csv
echo "shape,counter,dispatches,mean" > $OUT
: > $R/gpurun_out/pmc_dma_sq_times.log
for SH in "16 512 512" "64 128 128" "32 256 256"; do
  set -- $SH
  for CS in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/p_sq
    timeout 90 rocprofv3 --pmc $CS -M --output-format csv -d /tmp/p_sq -- $R/tools/abl_dma_0 64 $1 $2 $3 1 2>/dev/null | sed "s/^/[$CS] /" >> $R/gpurun_out/pmc_dma_sq_times.log
    python3 - "$1x$1 $2->$3" >> $OUT <<'P'
import glob, csv, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/p_sq/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{sys.argv[1]},{k},{len(v)},{sum(v) / len(v):.1f}")
dur = []
for f in glob.glob("/tmp/p_sq/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "GRBM_GUI_ACTIVE" and r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
if dur:
    print(f"{sys.argv[1]},kernel_ns_in_GRBM_pass,{len(dur)},{sum(dur) / len(dur):.1f}")
P
  done
done
cat $OUT
