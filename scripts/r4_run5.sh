mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r4_pytest1.log 2>&1
python bench.py --no-extras > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err
cat gpurun_out/r4_pytest1.log
python -c "
import json;d=json.loads(open('gpurun_out/r4_bench1.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_us'],[ (k['kernel'][:40],k['avg_launch_us'],k['frac']) for k in d['roofline']['family']['kernels']])"
