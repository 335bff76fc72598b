mkdir -p gpurun_out
{
for rep in 1 2; do
for sh in "64 64 128 128 1" "64 64 128 128 0" "64 64 256 128 1" "64 64 384 128 1" "64 64 96 128 0" "64 64 128 128 1 256" "64 32 256 256 1"; do
  ONLY=persist1,persistF,two80,two80F IT=20 timeout 120 tools/abl_conv_bench256 $sh
done; done
ONLY=two80,two80F IT=5 timeout 300 tools/abl_conv_bench256
} > gpurun_out/r4_two80c.log 2>&1
grep -v "^all" gpurun_out/r4_two80c.log | cut -c1-400 | tail -40
