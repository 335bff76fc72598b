#!/bin/bash
# Same-box A/B of two builds of the library (WAVEDM_LIB): scripts/ab_libs.sh <tag> <libA> <libB> [rounds]   -- alternating, headline passes only; per-kernel lines kept
tag=$1; A=$2; B=$3; R=${4:-3}
mkdir -p gpurun_out
out=gpurun_out/${tag}.log; : > $out
for r in $(seq 1 $R); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    WAVEDM_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python - "$v" "$lib" >> $out <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
k8 = [l for l in open('gpurun_out/ab_tmp.err') if l.startswith('[bench] convdma8')]
print(f"{sys.argv[1]} {sys.argv[2]:<40s} {d['value']:8.3f} img/s  ms/pass {d['ms_per_step']:8.2f}  | " + (k8[0].split('launches')[1].strip()[:60] if k8 else ''))
PY
  done
done
cat $out
