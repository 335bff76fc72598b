#!/bin/bash
# What the K-independent ~21 us of a big 3x3 launch are made of: the shipped 256 x 256 / 512 x 128 tiles stand-alone, complete / without the statistics pass / without the global
# stores of the output (-DWDM_EABL=1) / without any epilogue (-DWDM_EABL=2) / without the GroupNorm + SiLU prologue (pro = 0)
# build: tools/abl_conv_bench256 (see tools/conv_bench256.hip), and the same with -DWDM_EABL=1 / 2 as tools/abl_conv_bench256_e1 / _e2
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { NOREF=1 ROUNDS=2 ONLY=$1 "${@:2}" 2>&1 | grep -o "$1 wg *[0-9]* *[0-9.]* us *[0-9]* TF" ; }
for shape in "64 32 256 256" "64 32 768 256" "64 64 128 128"; do
  set -- $shape; T=t256x256; [ "$2" = "64" ] && T=t512x128
  echo "== B H Cin Cout = $shape ($T)"
  echo -n "complete           : "; run $T tools/abl_conv_bench256 $shape 1
  echo -n "no statistics pass : "; NOSTATS=1 run $T tools/abl_conv_bench256 $shape 1
  echo -n "no output stores   : "; run $T tools/abl_conv_bench256_e1 $shape 1
  echo -n "no epilogue at all : "; run $T tools/abl_conv_bench256_e2 $shape 1
  echo -n "no prologue (pro=0): "; run $T tools/abl_conv_bench256 $shape 0
  echo -n "neither            : "; run $T tools/abl_conv_bench256_e2 $shape 0
done
