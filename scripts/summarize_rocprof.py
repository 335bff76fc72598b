#!/usr/bin/env python3
"""Reduce rocprofv3 output directories to small per-kernel summaries (committed under profiles/).

    python scripts/summarize_rocprof.py stats  <dir> <out.md>     # --kernel-trace --stats run
    python scripts/summarize_rocprof.py pmc    <dir> <out.json>   # --pmc run (one or more counters)

Kernel names are shortened to the template arguments that identify a conv configuration."""
import glob
import json
import os
import re
import sys

import pandas as pd


def short(name: str) -> str:
    """Mangled kernel name (rocprofv3 -M) -> readable id; the 16-bit kernels are templates on the element type since round 5: DF16b = bf16, DF16_ = IEEE half."""
    r = _short(name)
    if r.endswith("_bf16") and "DF16_" in name and "DF16b" not in name:
        r = r[:-5] + "_f16"
    return r


def _short(name: str) -> str:
    """conv_kernel<T, MODE, TH, TW, NI, WAVES_M, WAVES_N, WM, WN>."""
    m = re.search(r"conv_kernelI(DF16b|DF16_|f|NS_7f32x3_tE)((?:Li\d+E)+)", name)
    if m:
        a = [int(v) for v in re.findall(r"Li(\d+)E", m.group(2))]
        if len(a) in (8, 9):
            mode = {0: "3x3s1", 1: "3x3s2", 2: "3x3ups", 3: "1x1"}[a[0]]
            ty = {"DF16b": "bf16", "DF16_": "f16", "f": "f32"}.get(m.group(1), "f32x3")
            return f"conv_{mode}_t{a[1]}x{a[2]}x{a[3]}_bn{16 * a[7] * a[5]}_{ty}"
    m = re.search(r"conv_dma_kernelI((?:Li\d+E)+)", name)
    if m:                                            # rounds 1-3: a template over the wave layout
        a = [int(v) for v in re.findall(r"Li(\d+)E", m.group(1))]
        return f"convdma_3x3s1_t16x16x1_bn128w{a[0] * a[1]}_bf16"
    m = re.search(r"conv_dma_kernel(?:ILb(\d)E|<(true|false)>|E)", name)
    if m:                                            # round 4: one configuration, <true> = bf16-tile ("packed") epilogue, <false> = fp32 epilogue
        return f"convdma_3x3s1_t16x16x1_bn128w8{'p' if m.group(1) == '1' or m.group(2) == 'true' else ''}_bf16"
    m = re.search(r"conv_dma256_kernelI((?:Li\d+E)+)(?:Lb(\d)E)?(?:Lb(\d)E)?", name)
    if m:                                            # <..., TH, PACKED, SC>: "p" = 16-bit-tile epilogue, "s" = the 512 x 128 tile's binary with the shortcut phase
        a = [int(v) for v in re.findall(r"Li(\d+)E", m.group(1))]
        return f"convdma_3x3s1_t{a[4]}x16x1_bn{16 * a[3] * a[1]}w8{'p' if m.group(2) == '1' else ''}{'s' if a[4] == 32 and m.group(3) == '1' else ''}_bf16"
    if "conv_dmap_kernel" in name:                   # <true>: packed epilogue, <false>: fp32 epilogue (residual convs) -- one name, as the library's profiler reports them
        return "convdmap_3x3s1_t16x16x1_bn128w8_bf16"
    m = re.search(r"conv_up4_kernelILi(\d+)ELi(\d+)E(?:Li(\d+)E)?", name)
    if m:
        return f"convup4_2x2x4_t{m.group(1)}x{m.group(1)}x{m.group(2)}_bn{32 * int(m.group(3) or 4)}w8_bf16"
    m = re.search(r"conv_dmax3t_kernel(?:ILb(\d)E|<(true|false)>)", name)
    if m:                                            # round 4: the f32x3 big tiles, <true> = 512 x 128, <false> = 256 x 256
        tall = m.group(1) == "1" or m.group(2) == "true"
        return "convdmax3_3x3s1_t32x16x1_bn128w8_f32x3" if tall else "convdmax3_3x3s1_t16x16x1_bn256w8_f32x3"
    if "conv_dmax3_kernel" in name:
        return "convdmax3_3x3s1_t16x16x1_bn128w8_f32x3"
    m = re.search(r"conv_dma8x3_kernelILi(\d+)E", name)
    if m:
        return f"convdma8x3_3x3s1_t8x8x2_bn{m.group(1)}w4_f32x3"
    m = re.search(r"conv_up4x3_kernelILi(\d+)ELi(\d+)E", name)
    if m:
        return f"convup4x3_2x2x4_t{m.group(1)}x{m.group(1)}x{m.group(2)}_bn128w8_f32x3"
    m = re.search(r"conv_s2x3_kernelILi(\d+)E", name)
    if m:
        return f"convs2x3_3x3s2_t16x16x1_bn{32 * int(m.group(1))}w8_f32x3"
    if "conv_gemmx3_kernel" in name:
        return "gemmx3_1x1_t16x16x1_bn128_f32x3"
    m = re.search(r"conv_s2_kernelILi(\d+)ELi(\d+)ELi(\d+)E", name)
    if m:
        return f"convs2_3x3s2_t{m.group(1)}x{m.group(1)}x{m.group(2)}_bn{32 * int(m.group(3))}w8_bf16"
    m = re.search(r"conv_dma8_kernelILi(\d+)E", name)
    if m:
        return f"convdma8_3x3s1_t8x8x2_bn{m.group(1)}w4_bf16"
    if "attn_fused_kernel" in name:
        m = re.search(r"attn_fused_kernelILb\dE(?:DF16b|DF16_)Lb(\d)E(?:Lb(\d)E)?", name)      # <PROJ, T, VTOK, QPROJ>: "t" = token-major V (the folded AttnBlock), "q" = query projection inside
        if m and m.group(1) == "1":
            return "attn_fused_n256tq_bf16" if m.group(2) == "1" else "attn_fused_n256t_bf16"
        return "attn_fused_n256_bf16"
    m = re.search(r"conv_wgrad_kernelILb(\d)E", name)
    if m:                                            # training: the direct weight gradient, <true> = 8 x 8 maps
        return "conv_wgrad_8x8_bf16" if m.group(1) == "1" else "conv_wgrad_bf16"
    m = re.search(r"conv_gemm_kernelI((?:Li\d+E)+)", name)
    if m:
        a = [int(v) for v in re.findall(r"Li(\d+)E", m.group(1))]
        return f"gemm_1x1_t{a[0]}x{a[1]}x{a[2]}_bn{16 * a[6] * a[4]}_bf16"
    m = re.search(r"conv_ws_kernelI(DF16b|f)((?:Li\d+E)+)", name)
    if m:
        a = [int(v) for v in re.findall(r"Li(\d+)E", m.group(2))]
        return f"convws_3x3s1_t{a[0]}x{a[1]}x{a[2]}_bn{16 * a[6] * a[4]}_{'bf16' if m.group(1) == 'DF16b' else 'f32'}"
    m = re.match(r"_ZN3wdm\d+([A-Za-z0-9_]+?)(?:I|E)", name)
    if m:
        return m.group(1)
    m = re.match(r"(?:void )?(?:wdm::)?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name[:60]


def find(d, pat):
    fs = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not fs:
        raise SystemExit(f"no {pat} under {d}")
    return fs


def stats(d, out):
    rows = []
    for f in find(d, "*kernel_trace.csv"):
        rows.append(pd.read_csv(f))
    df = pd.concat(rows)
    df = df[df["Kernel_Name"] != "Kernel_Name"].copy()          # (several per-process files concatenated into one: their header lines)
    for c in ("End_Timestamp", "Start_Timestamp"):
        df[c] = pd.to_numeric(df[c])
    # bench.py's board-roof calibration (tools/mfma_power_ubench.hip: k<...> / kl<...>, a CHILD process behind the timed passes; rocprofv3 follows it since ROCm 7.2)
    # is not part of the measured path: listed on its own line, kept out of the table and of the percentages
    cal = df["Kernel_Name"].str.match(r"^_Z\d+kl?I")
    cal_ms, cal_n = float(((df.loc[cal, "End_Timestamp"] - df.loc[cal, "Start_Timestamp"]) / 1e6).sum()), int(cal.sum())
    df = df[~cal].copy()
    df["k"] = df["Kernel_Name"].map(short)
    df["dur_us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
    g = df.groupby("k")["dur_us"].agg(["count", "sum", "mean", "min", "max"]).sort_values("sum", ascending=False)
    tot = g["sum"].sum()
    with open(out, "w") as fh:
        fh.write(f"rocprofv3 --kernel-trace --stats summary ({len(df)} dispatches, {tot / 1e6:.3f} s of kernel time)\n")
        if cal_n:
            fh.write(f"(not in the table: {cal_n} dispatches, {cal_ms:.0f} ms of tools/mfma_power_ubench.hip -- bench.py's board-roof calibration, a child process behind the timed passes)\n")
        fh.write("\n")
        fh.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for k, r in g.iterrows():
            fh.write(f"| {k} | {int(r['count'])} | {r['sum'] / 1e3:.2f} | {r['mean']:.2f} | {r['min']:.2f} | {r['max']:.2f} | {100 * r['sum'] / tot:.2f} |\n")
        extra = [c for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size") if c in df.columns]
        if extra:
            fh.write("\nresources per kernel (first dispatch):\n\n| kernel | " + " | ".join(extra) + " | grid | wg |\n|---|" + "---|" * (len(extra) + 2) + "\n")
            for k in g.index:
                r = df[df["k"] == k].iloc[0]
                fh.write(f"| {k} | " + " | ".join(str(r[c]) for c in extra) + f" | {r.get('Grid_Size_X', r.get('Grid_Size', ''))} | {r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))} |\n")
    print(open(out).read())


def pmc(d, out):
    rows = []
    for f in find(d, "*counter_collection.csv"):
        rows.append(pd.read_csv(f))
    df = pd.concat(rows)
    df["k"] = df["Kernel_Name"].map(short)
    piv = df.pivot_table(index=["k", "Dispatch_Id"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    res = {}
    for k, g in piv.groupby("k"):
        ent = {"dispatches": int(len(g))}
        for c in g.columns:
            if c in ("k", "Dispatch_Id"):
                continue
            ent[c] = {"sum": float(g[c].sum()), "mean_per_dispatch": float(g[c].mean())}
        res[k] = ent
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True)[:6000])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
