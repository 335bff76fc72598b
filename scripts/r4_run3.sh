mkdir -p gpurun_out
IT=20 tools/abl_conv_bench256 > gpurun_out/r4_packed_bench.log 2>&1
cat gpurun_out/r4_packed_bench.log
