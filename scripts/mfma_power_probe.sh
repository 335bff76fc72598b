#!/bin/bash
# Socket power and shader clock while tools/mfma_power_ubench.hip holds one variant for a few seconds: is the LDS-fed MFMA roof a power limit?
# usage (GPU box): bash scripts/mfma_power_probe.sh > gpurun_out/mfma_power_probe.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_power_ubench.hip -o /tmp/mfma_pw || exit 1
for cfg in "reg 0" "reg 1" "lds 0" "lds 1"; do
  set -- $cfg
  if [ "$2" = "1" ]; then Z="ZERO=1"; else Z="X=1"; fi
  env $Z HOLD=4 MODE=$1 /tmp/mfma_pw > /tmp/pw_run.log 2>&1 &
  BP=$!
  : > /tmp/pw.smi
  sleep 1.0
  while kill -0 $BP 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" >> /tmp/pw.smi
    sleep 0.15
  done
  wait $BP
  python3 - "$1" "$2" <<'PY'
import re,sys
s=open('/tmp/pw.smi').read()
p=[float(x) for x in re.findall(r'Power.*?:\s*([\d.]+)',s)]
c=[float(x) for x in re.findall(r'sclk.*?\((\d+)Mhz\)',s)]
p=p[1:-1] if len(p)>4 else p; c=c[1:-1] if len(c)>4 else c
run=open('/tmp/pw_run.log').read().strip().splitlines()[-1]
print(f"{'zeros ' if sys.argv[2]=='1' else 'random'} operands | {run} | {sum(p)/max(1,len(p)):7.1f} W  {sum(c)/max(1,len(c)):6.0f} MHz  ({len(p)} samples)")
PY
done
