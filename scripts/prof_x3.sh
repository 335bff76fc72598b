# rocprofv3 kernel trace + stats of the UNet in the f32x3 mode (run on the GPU box via gpurun): tools/abl_unet_run with DTYPE=2, 6 forward calls at batch 64
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_x3
DTYPE=2 rocprofv3 --kernel-trace --stats -M --output-format csv -d /tmp/prof_x3 -- $R/tools/abl_unet_run 64 6 > $R/gpurun_out/prof_x3.log 2>&1
cd $R && python scripts/summarize_rocprof.py stats /tmp/prof_x3 gpurun_out/kernel_stats_f32x3.md > /dev/null; head -16 gpurun_out/kernel_stats_f32x3.md | cut -c1-140
