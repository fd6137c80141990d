"""Condense rocprofv3 output (kernel-trace stats + separate --pmc passes) into one JSON/markdown
summary per round.  Usage: python tools/summarize_profiles.py gpurun_out/r01 profiles/r01"""
import collections
import csv
import json
import os
import sys

import numpy as np


def short(n):
    epi = "EpiGradStore" if "EpiGradStore" in n else "EpiGradAdam"
    if "wgrad_pair" in n: return "wgrad_pair_kernel<%s>" % epi
    if "bwd_pair64" in n: return "bwd_pair64_kernel<EpiMask,%s>" % epi
    if "bwd_pair" in n: return "bwd_pair_kernel<%s,%s>" % ("EpiSamplerSeed" if "SamplerSeed" in n else "EpiMask", epi)
    if "splitk_ws64" in n:                       # 64x32 tiles, or 64x64 (third template argument 64)
        kind = "P_ROW,EpiBiasAct" if "EpiBiasAct" in n else "P_COL,EpiMask"
        return "gemm_splitk_ws64_kernel<%s%s>" % (kind, ",64" if (", 64>" in n or "Li64E" in n) else "")
    if "EpiActionSeed" in n: return "gemm_splitk_%s_kernel<P_COL,EpiActionSeed>" % ("reg16" if "reg16" in n else "ws")
    if "wgrad_reg" in n: return "gemm_wgrad_reg_kernel<%s>" % epi
    if "reg16" in n and "EpiMse" in n: return "gemm_splitk_reg16_kernel<EpiMse>"
    if "reg16" in n: return "gemm_splitk_reg16_kernel<EpiBiasAct>"
    if "splitk_ws_pro" in n: return "gemm_splitk_ws_pro_kernel<EpiBiasAct,ProSampler>"
    if "splitk_ws" in n and "EpiMse" in n: return "gemm_splitk_ws_kernel<P_ROW,EpiMse>"
    if "splitk_ws" in n and "EpiBiasAct" in n: return "gemm_splitk_ws_kernel<P_ROW,EpiBiasAct>"
    if "splitk_ws" in n and "EpiMask" in n: return "gemm_splitk_ws_kernel<P_COL,EpiMask>"
    if "EpiMse" in n: return "gemm_splitk_reg_kernel<P_ROW,EpiMse>"
    if "EpiBiasAct" in n: return "gemm_splitk_reg_kernel<P_ROW,EpiBiasAct>"
    if "EpiMask" in n: return "gemm_splitk_reg_kernel<P_COL,EpiMask>"
    return n.split("(")[0][:60]


# the profiler categories bench.py's `roofline` / `kernels` blocks use (bench.CAT_MATCH): several template instances each
CATEGORIES = {
    "forward layer (gemm_splitk_ws*<P_ROW>)": ("gemm_splitk_ws_kernel<P_ROW", "gemm_splitk_ws64_kernel<P_ROW", "gemm_splitk_ws_pro_kernel",
                                                "gemm_splitk_reg_kernel<P_ROW"),
    "narrow forward layer (gemm_splitk_reg16<P_ROW>)": ("gemm_splitk_reg16_kernel<EpiBiasAct", "gemm_splitk_reg16_kernel<EpiMse"),
    "input gradient alone (P_COL)": ("gemm_splitk_ws_kernel<P_COL", "gemm_splitk_ws64_kernel<P_COL", "gemm_splitk_reg16_kernel<P_COL",
                                     "gemm_splitk_reg_kernel<P_COL"),
    "trailing weight gradient (wgrad_pair / gemm_wgrad_reg)": ("wgrad_pair_kernel", "gemm_wgrad_reg_kernel"),
    "bwd_pair_kernel (all instances)": ("bwd_pair_kernel", "bwd_pair64_kernel"),
}


def derive(d, stat):
    """HBM traffic / MFMA utilisation of one row from its `stat` (mean | median) counters."""
    f, w, b = ("%s_%s_per_launch" % (c, stat) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"))
    sfx = "" if stat == "median" else "_mean"
    # MI355X_MICROARCH.md HBM section: FETCH_SIZE (KB) under-reports wide coalesced reads by 2x
    if f in d:
        d["hbm_fetch_MB_corrected" + sfx] = d[f] * 2 * 1024 / 1e6
    if w in d:
        d["hbm_write_MB" + sfx] = d[w] * 1024 / 1e6
    if f in d and w in d:
        d["hbm_traffic_MB" + sfx] = d["hbm_fetch_MB_corrected" + sfx] + d["hbm_write_MB" + sfx]
    if b in d and "avg_us" in d:
        per_simd = d[b] / 1024.0
        d["mfma_busy_cycles_per_simd" + sfx] = per_simd
        d["mfma_util_at_2.4GHz" + sfx] = per_simd / (d["avg_us"] * 1e-6 * 2.4e9)


def main(src, dst):
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    out = collections.defaultdict(dict)
    import glob
    stats = list(csv.DictReader(open(glob.glob(src + "_trace/**/*kernel_stats.csv", recursive=True)[0])))
    for r in stats:
        k = short(r["Name"])
        if k.startswith(("void at::", "__amd", "at::")):
            continue
        out[k].update(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                      min_us=float(r["MinNs"]) / 1e3, max_us=float(r["MaxNs"]) / 1e3,
                      pct_of_gpu_time=float(r["Percentage"]))
    raw = collections.defaultdict(lambda: collections.defaultdict(list))          # kernel -> counter -> per-launch values
    for tag, f in (("fetch", "_fetch/f"), ("write", "_write/w"), ("mfma", "_mfma/m"), ("lds", "_lds/l")):
        found = glob.glob(src + f.split("/")[0] + "/**/*counter_collection.csv", recursive=True)
        if not found:
            continue
        for r in csv.DictReader(open(found[0])):
            raw[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in raw.items():
        if k in out:
            for c, v in cs.items():
                out[k][c + "_median_per_launch"] = float(np.median(v))
                out[k][c + "_mean_per_launch"] = float(np.mean(v))
    # category rows: what bench.py's `roofline` / `kernels` blocks report (the MEAN over every launch of every template
    # instance of the category; the per-kernel rows above them are per template instance, MEDIAN launch in the table)
    cats = {}
    for name, pats in CATEGORIES.items():
        members = [k for k in out if any(k.startswith(p) for p in pats)]
        calls = sum(out[k].get("calls", 0) for k in members)
        if not calls:
            continue
        d = {"members": members, "calls": calls,
             "avg_us": sum(out[k]["avg_us"] * out[k]["calls"] for k in members) / calls,
             "pct_of_gpu_time": sum(out[k].get("pct_of_gpu_time", 0) for k in members)}
        for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
            vals = [x for k in members for x in raw.get(k, {}).get(c, [])]
            if vals:
                d[c + "_mean_per_launch"] = float(np.mean(vals))
                d[c + "_median_per_launch"] = float(np.median(vals))
        cats["[category] " + name] = d
    out.update(cats)
    for k, d in out.items():
        derive(d, "median")
        derive(d, "mean")
        if "SQ_LDS_BANK_CONFLICT_median_per_launch" in d and d.get("SQ_LDS_IDX_ACTIVE_median_per_launch"):
            d["lds_conflict_frac"] = d["SQ_LDS_BANK_CONFLICT_median_per_launch"] / d["SQ_LDS_IDX_ACTIVE_median_per_launch"]
    json.dump(out, open(dst + "_summary.json", "w"), indent=1, sort_keys=True)

    def cell(d, key, fmt, scale=1.0):
        a, b = d.get(key), d.get(key + "_mean")
        if a is None and b is None:
            return "-"
        return " / ".join(fmt % (x * scale) if x is not None else "-" for x in (a, b))
    with open(dst + "_summary.md", "w") as f:
        f.write("Per launch: `median / mean` over the launches of the row (a template instance, or a `[category]` = every instance "
                "bench.py files under it: its `roofline.traffic` and `mfma_busy` are the category MEAN).  HBM = FETCH_SIZE x 2 + "
                "WRITE_SIZE (MI355X_MICROARCH.md); MFMA util = busy cycles per SIMD / (mean launch duration x 2.4 GHz).\n\n")
        f.write("| kernel | calls | avg us | % GPU time | HBM MB/launch median / mean | MFMA util median / mean | LDS conflict |\n|---|---|---|---|---|---|---|\n")
        for k, d in sorted(out.items(), key=lambda kv: (kv[0].startswith("[category]"), -kv[1].get("pct_of_gpu_time", 0))):
            f.write("| %s | %d | %.2f | %.1f | %s | %s | %s |\n" % (
                k, d.get("calls", 0), d.get("avg_us", 0), d.get("pct_of_gpu_time", 0),
                cell(d, "hbm_traffic_MB", "%.1f"), cell(d, "mfma_util_at_2.4GHz", "%.0f%%", 100.0),
                "%.1f%%" % (100 * d["lds_conflict_frac"]) if "lds_conflict_frac" in d else "-"))
    print(open(dst + "_summary.md").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
