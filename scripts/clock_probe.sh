#!/bin/bash
# Samples the GPU's shader clock and socket power while bench.py runs (how far the chip is from its nominal 2.4 GHz under the conv load).
# Usage (on the GPU box): scripts/clock_probe.sh [bench args...]   -> gpurun_out/clock_probe.log, gpurun_out/clock_probe_bench.json
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/clock_probe_bench.json 2> gpurun_out/clock_probe_bench.err &
BP=$!
: > gpurun_out/clock_probe.log
while kill -0 $BP 2>/dev/null; do
  echo "t=$(date +%s.%N)" >> gpurun_out/clock_probe.log
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" >> gpurun_out/clock_probe.log
  sleep 0.25
done
wait $BP
tail -1 gpurun_out/clock_probe_bench.json
grep -E "sclk" gpurun_out/clock_probe.log | sort | uniq -c | sort -rn | head -12
grep -E "Power" gpurun_out/clock_probe.log | sort | uniq -c | sort -rn | head -12
