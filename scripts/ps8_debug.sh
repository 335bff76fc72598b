#!/bin/bash
# the 8-rank patch-sharded restore() worker of tests/test_gpu_dist.py, stand-alone, with its stderr kept
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0 WDM_TEST_BACKEND=gloo WDM_TEST_MODE=patch8
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 tests/dist_worker.py /tmp/ps8.pt > gpurun_out/r06_ps8.out 2> gpurun_out/r06_ps8.err
echo "rc=$?"; grep -v "^$" gpurun_out/r06_ps8.err | grep "rank0\]\|Error" | head -30
