#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the per-shape PMC passes over the stand-alone launchers (scripts/prof_r02b.sh):
gpurun_out/pmc_dma_shapes.csv (dominant LDS-DMA 3x3 kernel, weighted by the launches of each shape in one UNet call) and
gpurun_out/pmc_dma8_shapes.csv (8x8 kernel).  bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB -- on gfx950 FETCH_SIZE reports half of a wide
coalesced read (MI355X_MICROARCH.md, HBM section).      usage: traffic_from_shapes.py [tag]"""
import csv, json, os, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
B = 64


def rows(path):
    return list(csv.DictReader(open(path))) if os.path.isfile(path) else []


out = {"_note": "HBM bytes per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs, never combined with tracing) over the "
                "stand-alone launchers of the two LDS-DMA conv kernels (tools/dma_ablate.hip, tools/dma8_ablate.hip) per layer shape at batch 64; the "
                "dominant kernel's figure is weighted by the launches of each shape in one UNet call.  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 "
                "FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section), so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.  The PMC "
                "passes over the whole bench.py process crash inside this rocprofv3 build in round 2 (SIGSEGV in the tool at the first elementwise launch; "
                "one pass hung with 'AQL packet is malformed').", "kernels": {}}
sh = {}
for r in rows(os.path.join(REPO, "gpurun_out", "pmc_dma_shapes.csv")):
    k = (int(r["H"]), int(r["Cin"]), int(r["Cout"]), int(r["pro"]), int(r["count_per_unet_call"]))
    sh.setdefault(k, {})[r["counter"]] = float(r["mean_kib"])
per, tot, n = [], 0.0, 0
for (H, cin, cout, pro, cnt), v in sh.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    alg = B * H * H * (cin + cout) * 2 + 9 * cin * cout * 2
    per.append({"shape": f"{H}x{H} {cin}->{cout}" + (" gn" if pro else ""), "launches_per_unet_call": cnt, "fetch_kib_raw": v["FETCH_SIZE"], "write_kib": v["WRITE_SIZE"],
                "hbm_bytes_per_launch": round(hbm), "algorithmic_bytes": alg, "ratio": round(hbm / alg, 3)})
    tot += hbm * cnt
    n += cnt
if n:
    out["kernels"]["convdma_3x3s1_t16x16x1_bn128w8_bf16"] = {"hbm_bytes_per_launch": tot / n, "launches_weighted": n, "per_shape": per}
sh8 = {}
for r in rows(os.path.join(REPO, "gpurun_out", "pmc_dma8_shapes.csv")):
    sh8.setdefault((int(r["Cin"]), int(r["Cout"])), {})[r["counter"]] = float(r["mean_kib"])
per8 = []
for (cin, cout), v in sh8.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    alg = B * 64 * (cin + cout) * 2 + 9 * cin * cout * 2
    per8.append({"shape": f"8x8 {cin}->{cout}", "fetch_kib_raw": v["FETCH_SIZE"], "write_kib": v["WRITE_SIZE"], "hbm_bytes_per_launch": round(hbm), "algorithmic_bytes": alg,
                 "ratio": round(hbm / alg, 3)})
if per8:
    out["kernels"]["conv_dma8_kernel"] = {"hbm_bytes_per_launch": next(p["hbm_bytes_per_launch"] for p in per8), "per_shape": per8,
                                          "_note": "slab-major weights, 48-wide N tile, N tiles grouped four ways over the XCDs (the library's default, GN=4 in the "
                                                   "launcher): every XCD streams a quarter of the 10-21 MB weight tensor; ungrouped: 93.7 MB per 768->768 launch"}
json.dump(out, open(os.path.join(REPO, "profiles", f"{tag}_traffic.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    print(k, round(v["hbm_bytes_per_launch"] / 1e6, 1), "MB/launch")
