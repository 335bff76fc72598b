mkdir -p gpurun_out
for pro in 1 0; do
  ONLY=persist1,two80 IT=20 tools/abl_conv_bench256 64 64 128 128 $pro
  ONLY=persist1,two80 IT=20 tools/abl_conv_bench256 64 64 256 128 $pro
  ONLY=persist1,two80 IT=20 tools/abl_conv_bench256 64 64 384 128 $pro
  ONLY=t256x256,two80 IT=20 tools/abl_conv_bench256 64 32 256 256 $pro
  ONLY=t256x256,two80 IT=20 tools/abl_conv_bench256 64 32 512 256 $pro
  ONLY=persist1,two80 IT=20 tools/abl_conv_bench256 64 16 512 512 $pro
done > gpurun_out/r4_two80.log 2>&1
cat gpurun_out/r4_two80.log
