#!/bin/bash
# Compile every stand-alone tool under tools/ (ablation / calibration harnesses and the retired kernels of tools/experiments/ they include) against the CURRENT headers of
# wavedm_amd/csrc -- nothing else builds them, so this is what keeps them from rotting.  hipcc cross-compiles without a GPU; ~30 s per file.
#   scripts/build_tools.sh [pattern]      -> tools/abl_<name> per tools/<name>.hip (git-ignored), a PASS / FAIL line each, exit 1 on any failure
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
fail=0
for src in tools/*${1:-}*.hip; do
  name=$(basename "$src" .hip)
  out=tools/abl_$name
  case $name in
    mfma_power_ubench) out=tools/abl_mfma_power ;;
    unet_run) extra="-L wavedm_amd/csrc -lwavedm_hip -Wl,-rpath,\$ORIGIN/../wavedm_amd/csrc" ;;
    dmap_timeline) extra="-DWDM_EPI_TS=8" ;;          # (the timeline tools need their stamp macro: see the build line at the top of each file)
    *) extra="" ;;
  esac
  if eval $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -I wavedm_amd/csrc -I tools/experiments -I include "$src" $extra -o "$out" > /tmp/build_tool_$name.log 2>&1; then
    echo "PASS $src"
  else
    echo "FAIL $src  (/tmp/build_tool_$name.log)"; tail -5 /tmp/build_tool_$name.log; fail=1
  fi
done
exit $fail
