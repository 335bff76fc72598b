mkdir -p gpurun_out
{
for rep in 1 2; do
for sh in "64 64 128 128 1" "64 64 128 128 0" "64 64 256 128 1" "64 64 384 128 1" "64 64 96 128 0"; do
  ONLY=persist1,two80 IT=20 tools/abl_conv_bench256 $sh
done; done
} > gpurun_out/r4_two80b.log 2>&1
grep -v "^all" gpurun_out/r4_two80b.log | cut -c1-330
