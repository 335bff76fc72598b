mkdir -p gpurun_out
{
for rep in 1 2; do
for sh in "64 64 128 128 1" "64 64 128 128 0" "64 64 256 128 1" "64 64 384 128 1" "64 64 96 128 0" "64 32 256 256 1" "64 32 512 256 1" "64 32 768 256 1" "64 16 512 512 1" "64 16 1024 512 1"; do
  echo "== pack"; IT=20 tools/abl_conv_bench256 $sh
  echo "== nopack"; IT=20 tools/abl_conv_bench256_nopack $sh
done; done
} > gpurun_out/r4_pack_ab.log 2>&1
grep -v "^all" gpurun_out/r4_pack_ab.log | cut -c1-330
