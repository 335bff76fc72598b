#!/bin/bash
# Socket power and shader clock while ONE kernel variant of tools/abl_conv_bench256 runs back to back for a few seconds: energy per launch = power x time.
# usage (GPU box): scripts/power_probe.sh <variant> B H Cin Cout [pro] [sc]     -> one line
V=$1; shift
ONLY=$V NOREF=1 ROUNDS=1 IT=${IT:-15000} tools/abl_conv_bench256 "$@" > /tmp/pp_$V.log 2>&1 &
BP=$!
: > /tmp/pp_$V.smi
sleep 0.7
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" >> /tmp/pp_$V.smi
  sleep 0.15
done
wait $BP
python3 - "$V" <<'PY'
import re,sys
v=sys.argv[1]
s=open(f'/tmp/pp_{v}.smi').read()
p=[float(x) for x in re.findall(r'Power.*?:\s*([\d.]+)',s)]
c=[float(x) for x in re.findall(r'sclk.*?\((\d+)Mhz\)',s)]
log=open(f'/tmp/pp_{v}.log').read()
m=re.search(rf'{v} wg\s+\d+\s+([\d.]+) us',log)
us=float(m.group(1)) if m else float('nan')
n=len(p)
p=p[2:-2] if n>6 else p; c=c[2:-2] if len(c)>6 else c
P=sum(p)/max(1,len(p)); C=sum(c)/max(1,len(c))
print(f"{v:12s} {us:7.1f} us/launch  {P:7.1f} W  {C:6.0f} MHz  {P*us/1e3:7.2f} mJ/launch  ({len(p)} samples)")
PY
