#!/usr/bin/env python3
"""Turn the raw output of scripts/prof_r01.sh (gpurun_out/raw/*.csv[.gz]) into the committed summaries under profiles/:
   <tag>_kernel_stats.md, <tag>_pmc_<counter>.json, <tag>_traffic.json (HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB,
   the gfx950 correction of MI355X_MICROARCH.md's HBM section), <tag>_bench.json.      usage: make_profiles.py [tag]"""
import glob, gzip, json, os, shutil, subprocess, sys, tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
raw = os.path.join(REPO, "gpurun_out", "raw")
prof = os.path.join(REPO, "profiles")
summ = os.path.join(REPO, "scripts", "summarize_rocprof.py")


def unpack(name, as_name):
    d = tempfile.mkdtemp()
    src = os.path.join(raw, name)
    with (gzip.open(src, "rb") if src.endswith(".gz") else open(src, "rb")) as f, open(os.path.join(d, as_name), "wb") as o:
        shutil.copyfileobj(f, o)
    return d


d = unpack("kernel_trace.csv.gz", "x_kernel_trace.csv")
subprocess.run([sys.executable, summ, "stats", d, os.path.join(prof, f"{tag}_kernel_stats.md")], check=True, stdout=subprocess.DEVNULL)
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "sq"):
    f = f"pmc_{c}.csv.gz"
    if not os.path.isfile(os.path.join(raw, f)):
        continue
    d = unpack(f, "x_counter_collection.csv")
    out = os.path.join(prof, f"{tag}_pmc_{c}.json")
    subprocess.run([sys.executable, summ, "pmc", d, out], check=True, stdout=subprocess.DEVNULL)
    res[c] = json.load(open(out))
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    ks = {}
    for k, v in res["FETCH_SIZE"].items():
        w = res["WRITE_SIZE"].get(k)
        if not w:
            continue
        fk, wk = v["FETCH_SIZE"]["mean_per_dispatch"], w["WRITE_SIZE"]["mean_per_dispatch"]
        ks[k] = {"dispatches": v["dispatches"], "fetch_kib_raw": fk, "write_kib": wk, "hbm_bytes_per_launch": (2 * fk + wk) * 1024}
    json.dump({"_note": "HBM bytes per launch from rocprofv3 PMC passes (scripts/prof_r01.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs of "
                        "bench.py --ddim-steps 5). FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced "
                        "read (MI355X_MICROARCH.md, HBM section), so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.", "kernels": ks},
              open(os.path.join(prof, f"{tag}_traffic.json"), "w"), indent=1, sort_keys=True)
bj = os.path.join(REPO, "gpurun_out", "bench_prof.json")
if os.path.isfile(bj):
    shutil.copy(bj, os.path.join(prof, f"{tag}_bench_under_rocprof.json"))
print("profiles written:", sorted(os.listdir(prof)))
