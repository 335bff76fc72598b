#!/bin/bash
# Round 6: the driver's 8-rank scaling run rehearsed on ONE device (gloo; all ranks share cuda:0).  Logs under gpurun_out/.
mkdir -p gpurun_out
export WAVEDM_BENCH_BACKEND=gloo
python bench.py --gpus 8 --steps 1 --warmup 0 --ddim-steps 5 --no-extras > gpurun_out/r06_rehearse8_c1.json 2> gpurun_out/r06_rehearse8_c1.err; echo "c1 rc=$?"
python bench.py --gpus 8 --steps 1 --warmup 0 --ddim-steps 5 --no-extras --workload c4 --batch 2 --no-roofline > gpurun_out/r06_rehearse8_c4.json 2> gpurun_out/r06_rehearse8_c4.err; echo "c4 rc=$?"
python bench.py --gpus 8 --steps 1 --warmup 0 --ddim-steps 5 --no-extras --workload c4 --patch-sharded > gpurun_out/r06_rehearse8_c4ps.json 2> gpurun_out/r06_rehearse8_c4ps.err; echo "c4ps rc=$?"
tail -c 1500 gpurun_out/r06_rehearse8_c1.err; cat gpurun_out/r06_rehearse8_c1.json | cut -c1-600
