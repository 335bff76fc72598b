cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for L in dfe final; do
  if [ $L = final ]; then unset WAVEDM_LIB; else export WAVEDM_LIB=$R/tools/abl_lib_$L.so; fi
  rm -rf /tmp/ps_$L
  timeout 600 rocprofv3 --kernel-trace --stats -M --output-format csv -d /tmp/ps_$L -- python bench.py --no-extras --no-cpu-baseline > /tmp/b_$L.json 2>/dev/null
  python3 - $L <<'P'
import csv, glob, json, sys
L = sys.argv[1]
rows = []
for f in glob.glob(f"/tmp/ps_{L}/*/*kernel_stats.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
d = json.loads(open(f"/tmp/b_{L}.json").read().strip().splitlines()[-1])
print(f"{L}: {d['value']} img/s traced, kernel time {tot/1e6:.1f} ms; top: " + "; ".join(f"{r['Name'][:28]} {float(r['AverageNs'])/1e3:.2f} us x{r['Calls']}" for r in rows[:4]))
P
done
