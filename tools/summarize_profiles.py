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
    for tag, f in (("fetch", "_fetch/f"), ("write", "_write/w"), ("mfma", "_mfma/m"), ("lds", "_lds/l")):
        found = glob.glob(src + f.split("/")[0] + "/**/*counter_collection.csv", recursive=True)
        if not found:
            continue
        path = found[0]
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if k in out:
                for c, v in cs.items():
                    out[k][c + "_median_per_launch"] = float(np.median(v))
    for k, d in out.items():
        # MI355X_MICROARCH.md HBM section: FETCH_SIZE (KB) under-reports wide coalesced reads by 2x
        if "FETCH_SIZE_median_per_launch" in d:
            d["hbm_fetch_MB_corrected"] = d["FETCH_SIZE_median_per_launch"] * 2 * 1024 / 1e6
        if "WRITE_SIZE_median_per_launch" in d:
            d["hbm_write_MB"] = d["WRITE_SIZE_median_per_launch"] * 1024 / 1e6
        if "hbm_fetch_MB_corrected" in d and "hbm_write_MB" in d:
            d["hbm_traffic_MB"] = d["hbm_fetch_MB_corrected"] + d["hbm_write_MB"]
        if "SQ_VALU_MFMA_BUSY_CYCLES_median_per_launch" in d and "avg_us" in d:
            per_simd = d["SQ_VALU_MFMA_BUSY_CYCLES_median_per_launch"] / 1024.0
            d["mfma_busy_cycles_per_simd"] = per_simd
            d["mfma_util_at_2.4GHz"] = per_simd / (d["avg_us"] * 1e-6 * 2.4e9)
        if "SQ_LDS_BANK_CONFLICT_median_per_launch" in d and d.get("SQ_LDS_IDX_ACTIVE_median_per_launch"):
            d["lds_conflict_frac"] = d["SQ_LDS_BANK_CONFLICT_median_per_launch"] / d["SQ_LDS_IDX_ACTIVE_median_per_launch"]
    json.dump(out, open(dst + "_summary.json", "w"), indent=1, sort_keys=True)
    with open(dst + "_summary.md", "w") as f:
        f.write("| kernel | calls | avg us | %% GPU time | HBM MB/launch (fetch x2 + write) | MFMA util | LDS conflict |\n|---|---|---|---|---|---|---|\n")
        for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("pct_of_gpu_time", 0)):
            f.write("| %s | %d | %.2f | %.1f | %s | %s | %s |\n" % (
                k, d.get("calls", 0), d.get("avg_us", 0), d.get("pct_of_gpu_time", 0),
                "%.1f" % d["hbm_traffic_MB"] if "hbm_traffic_MB" in d else "-",
                "%.0f%%" % (100 * d["mfma_util_at_2.4GHz"]) if "mfma_util_at_2.4GHz" in d else "-",
                "%.1f%%" % (100 * d["lds_conflict_frac"]) if "lds_conflict_frac" in d else "-"))
    print(open(dst + "_summary.md").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
