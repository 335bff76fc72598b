"""Kernel-to-kernel gaps from a rocprofv3 kernel trace (csv) of tools/abl_unet_run: per forward call the span, the sum of kernel durations and the idle time
between consecutive kernels, and the gaps grouped by the kernel that FOLLOWS them (what it costs to start that kernel after its predecessor drained)."""
import csv, sys, re, collections
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from summarize_rocprof import short as short_name  # noqa: E402

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# forwards: split at the first kernel of the UNet (the timestep embedding); keep the last four
starts = [i for i, r in enumerate(rows) if "timestep_embedding" in r[2]]
starts.append(len(rows))
calls = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][-4:]
gap_after = collections.defaultdict(list)
for a, b in calls:
    ks = rows[a:b]
    span = ks[-1][1] - ks[0][0]
    busy = sum(e - s for s, e, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    print(f"forward: {len(ks)} kernels, span {span/1e3:.1f} us, kernel time {busy/1e3:.1f} us, idle {sum(gaps)/1e3:.1f} us ({100*sum(gaps)/span:.1f} %), mean gap {sum(gaps)/len(gaps)/1e3:.2f} us")
    for i, g in enumerate(gaps):
        gap_after[short_name(ks[i + 1][2])].append(g)
print("\ngap BEFORE a kernel, by kernel (us: mean, max, count over the forwards above)")
for k, v in sorted(gap_after.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k[:70]:70s} {sum(v)/len(v)/1e3:6.2f} {max(v)/1e3:7.2f} {len(v):5d}   total {sum(v)/1e3/len(calls):7.1f} us per forward")
