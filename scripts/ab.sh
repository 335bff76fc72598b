#!/bin/bash
# A/B of environment-selected variants inside one gpurun call: scripts/ab.sh "VAR=a" "VAR=b" ...   (each runs bench.py briefly)
# prints the JSON value and the per-shape lines of every variant into gpurun_out/ab_<i>.log
mkdir -p gpurun_out
i=0
for v in "$@"; do
  env $v python bench.py --steps 1 --warmup 1 --ddim-steps ${AB_DDIM:-20} --no-cpu-baseline ${AB_EXTRAS:---no-extras} > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.log
  echo "== $v: $(python -c "import json,sys; d=json.loads(open('gpurun_out/ab_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['roofline']['achieved'])")"
  i=$((i+1))
done
