# Whole-path PMC passes (run on the GPU box via gpurun): rocprofv3 --pmc over tools/abl_unet_run, the Python-free driver of one UNet call at batch 64 --
# every kernel of the sampling path under its real launch sequence (cold weights, residuals, fused shortcuts, statistics).  One counter set per pass,
# never combined with tracing.  Outputs: gpurun_out/raw/unet_pmc_<set>.csv.gz and gpurun_out/unet_prof.jsonl (the library's algorithmic flops / bytes).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/raw
B=${B:-64}
PROF=1 $R/tools/abl_unet_run $B 3 > $R/gpurun_out/unet_prof.jsonl 2>&1
for C in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rm -rf /tmp/p_$C
  timeout 300 rocprofv3 --pmc $C -M --output-format csv -d /tmp/p_$C -- $R/tools/abl_unet_run $B 2 > $R/gpurun_out/unet_pmc_$C.log 2>&1
  gzip -c /tmp/p_$C/*/*counter_collection.csv > $R/gpurun_out/raw/unet_pmc_$C.csv.gz
done
rm -rf /tmp/p_sq
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -M --output-format csv -d /tmp/p_sq -- $R/tools/abl_unet_run $B 2 > $R/gpurun_out/unet_pmc_sq.log 2>&1
gzip -c /tmp/p_sq/*/*counter_collection.csv > $R/gpurun_out/raw/unet_pmc_sq.csv.gz
ls -la $R/gpurun_out/raw | tail -6
head -3 $R/gpurun_out/unet_prof.jsonl; tail -2 $R/gpurun_out/unet_pmc_FETCH_SIZE.log
