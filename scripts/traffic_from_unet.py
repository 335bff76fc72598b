#!/usr/bin/env python3
"""profiles/<tag>_traffic.json and profiles/<tag>_pmc_unet_sq.json from the whole-path PMC passes of scripts/pmc_unet.sh (rocprofv3 --pmc over
tools/abl_unet_run: one raindrop_wavelet UNet call at batch 64 through the C ABI, no Python in the process).

Per kernel: dispatches per UNet call, HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB (on gfx950 FETCH_SIZE reports half of a wide coalesced
read -- MI355X_MICROARCH.md, HBM section), the library's algorithmic bytes per launch (input + weights + output (+ residual) once: wdm_prof_report), their
ratio, and the share of the call's kernel time (from the GRBM pass's timestamps).      usage: traffic_from_unet.py [tag]"""
import json
import os
import sys

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import short  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
RAW = os.path.join(REPO, "gpurun_out", "raw")
NCALLS = 3          # abl_unet_run <B> 2: one warm-up + two timed calls


def load(name):
    d = pd.read_csv(os.path.join(RAW, f"unet_pmc_{name}.csv.gz"))
    d["k"] = d.Kernel_Name.map(lambda n: short(n.replace(".kd", "")))
    return d


alg = {}
for line in open(os.path.join(REPO, "gpurun_out", "unet_prof.jsonl")):
    if not line.startswith("{"):
        continue
    e = json.loads(line)
    k = e["kernel"].split("|")[0]
    a = alg.setdefault(k, {"launches": 0, "bytes": 0.0, "flops": 0.0, "ms": 0.0})
    for f in ("launches", "bytes", "flops", "ms"):
        a[f] += e[f]
fetch, write, grbm = load("FETCH_SIZE"), load("WRITE_SIZE"), load("GRBM_GUI_ACTIVE")
grbm["us"] = (grbm.End_Timestamp - grbm.Start_Timestamp) / 1e3
tot_us = grbm.us.sum()
out = {"_note": "HBM-side bytes per launch of every kernel of one raindrop_wavelet UNet call (batch 64, bf16, 64x64) from separate rocprofv3 --pmc passes "
                "(FETCH_SIZE, WRITE_SIZE, GRBM_GUI_ACTIVE; never combined with tracing) over tools/abl_unet_run, the Python-free driver of the C ABI: real "
                "launch sequence, cold weights, residual / shortcut / statistics operands.  bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE "
                "reports half of a wide coalesced read: MI355X_MICROARCH.md, HBM section); algorithmic = input + weights + output (+ residual) once "
                "(wdm_prof_report).  Infinity-Cache hits are counted by FETCH_SIZE, so ratios above 1 are re-reads served on-die or from HBM alike.  "
                "clock_ghz = GRBM_GUI_ACTIVE / 8 / dispatch duration in that pass.", "kernels": {}}
f_m, w_m = fetch.groupby("k").Counter_Value.mean(), write.groupby("k").Counter_Value.mean()
n_k = fetch.groupby("k").size()
g_us, g_cnt = grbm.groupby("k").us.sum(), grbm.groupby("k").Counter_Value.sum()
for k in sorted(n_k.index, key=lambda k: -g_us.get(k, 0.0)):
    hbm = (2 * f_m[k] + w_m.get(k, 0.0)) * 1024
    e = {"launches_per_unet_call": round(n_k[k] / NCALLS, 2), "fetch_kib_raw": round(float(f_m[k]), 1), "write_kib": round(float(w_m.get(k, 0.0)), 1),
         "hbm_bytes_per_launch": round(float(hbm)), "share_of_kernel_time": round(float(g_us.get(k, 0.0) / tot_us), 4),
         "avg_us_under_pmc": round(float(g_us.get(k, 0.0) / max(1, (grbm.k == k).sum())), 2),
         "clock_ghz": round(float(g_cnt.get(k, 0.0) / 8 / max(1e-9, g_us.get(k, 0.0) * 1e3)), 3)}
    if k in alg and alg[k]["launches"]:
        e["algorithmic_bytes_per_launch"] = round(alg[k]["bytes"] / alg[k]["launches"])
        e["ratio"] = round(hbm / e["algorithmic_bytes_per_launch"], 3)
        e["algorithmic_gflop_per_launch"] = round(alg[k]["flops"] / alg[k]["launches"] / 1e9, 3)
    out["kernels"][k] = e
json.dump(out, open(os.path.join(REPO, "profiles", f"{tag}_traffic.json"), "w"), indent=1)

sq = load("sq")
p = sq.pivot_table(index=["Dispatch_Id", "k"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
agg = p.groupby("k").sum(numeric_only=True)
res = {"_note": "SQ counters per kernel over the same driver (one pass, 8 SQ slots), summed over the dispatches of three UNet calls.  mfma_util = "
                "SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES) (matrix-pipe busy share of the CU-busy time, four SIMDs per CU); the SQ_WAIT_* / "
                "SQ_ACTIVE_* ratios are shares of SQ_WAVE_CYCLES.", "kernels": {}}
for k, r in agg.sort_values("SQ_BUSY_CU_CYCLES", ascending=False).iterrows():
    if r["SQ_WAVE_CYCLES"] <= 0:
        continue
    res["kernels"][k] = {"mfma_util": round(r["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * max(1.0, r["SQ_BUSY_CU_CYCLES"])), 3),
                         "wait_inst_any": round(r["SQ_WAIT_INST_ANY"] / r["SQ_WAVE_CYCLES"], 3), "wait_any": round(r["SQ_WAIT_ANY"] / r["SQ_WAVE_CYCLES"], 3),
                         "active_inst_any": round(r["SQ_ACTIVE_INST_ANY"] / r["SQ_WAVE_CYCLES"], 3),
                         "lds_bank_conflict": round(r["SQ_LDS_BANK_CONFLICT"] / max(1.0, r["SQ_LDS_IDX_ACTIVE"]), 4)}
json.dump(res, open(os.path.join(REPO, "profiles", f"{tag}_pmc_unet_sq.json"), "w"), indent=1)
for k, e in list(out["kernels"].items())[:16]:
    print(f"{k:<44s} n {e['launches_per_unet_call']:6.1f}  {e['hbm_bytes_per_launch'] / 1e6:8.1f} MB  alg {e.get('algorithmic_bytes_per_launch', 0) / 1e6:8.1f} MB  x{e.get('ratio', 0):5.2f}  "
          f"{100 * e['share_of_kernel_time']:5.1f} %  mfma {res['kernels'].get(k, {}).get('mfma_util', 0):.2f}  {e['clock_ghz']:.2f} GHz")
