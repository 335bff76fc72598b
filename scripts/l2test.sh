# L2 hit-rate experiment ("XCD tile order variants"): TCC hit / miss counters for the tile-order variants of the main conv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/raw
for g in 1 4; do
  for t in bn128 0; do
    GN=$g rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -M --output-format csv -d /tmp/l2_${t}_$g -- ./tools/abl_$t 64 16 1280 512 1 > /dev/null 2>&1
    python - <<PY
import pandas as pd, glob
f=glob.glob('/tmp/l2_${t}_$g/**/*counter_collection.csv', recursive=True)[0]
d=pd.read_csv(f); d=d[d.Kernel_Name.str.contains('conv_kernel')]
p=d.pivot_table(index='Dispatch_Id',columns='Counter_Name',values='Counter_Value',aggfunc='sum')
m=p.mean(); print('cfg ${t} gn=$g', {k: float(v) for k,v in m.items()}, 'hit rate %.3f' % (m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum'])))
PY
  done
done
