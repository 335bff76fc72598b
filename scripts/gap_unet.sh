# Kernel-to-kernel gaps of one UNet forward (run on the GPU box via gpurun): rocprofv3 --kernel-trace over tools/abl_unet_run, then
# scripts/gap_from_trace.py sums kernel durations against the span of each forward and lists the gaps by the kernel that precedes them.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
rm -rf /tmp/gap
$R/tools/abl_unet_run 64 20 | tee $R/gpurun_out/gap_plain.log
timeout 300 rocprofv3 --kernel-trace -M --output-format csv -d /tmp/gap -- $R/tools/abl_unet_run 64 6 > $R/gpurun_out/gap_run.log 2>&1
python3 $R/scripts/gap_from_trace.py /tmp/gap/*/*kernel_trace.csv | tee $R/gpurun_out/gap_report.txt
