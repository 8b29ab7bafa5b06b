#!/usr/bin/env python3
"""tools/summarize_profile.py <gpurun_out/prof_TAG> <out_prefix>

Condenses the rocprofv3 output of tools/profile_bench.sh into the two files kept under profiles/:
  <out_prefix>_kernel_stats.csv   the --kernel-trace --stats table as rocprofv3 wrote it
  <out_prefix>_pmc_summary.json   mean counter value per launch for every gm:: kernel, plus derived figures for
                                  the pair kernel (HBM traffic with the gfx950 FETCH_SIZE x2 correction, MFMA busy)
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    shutil.copyfile(stats, prefix + "_kernel_stats.csv")
    acc = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        f = newest(os.path.join(d, "*", "*_counter_collection.csv"))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if name.startswith("gm::k_pair_score_sample"):      # the product instantiation <false> and the debug one
                    name = "gm::k_pair_score_sample"
                if "gm::" not in name:
                    continue
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in sorted(cs.items())}
           for k, cs in acc.items()}
    pair = out.get("gm::k_pair_score_sample", {})
    if {"FETCH_SIZE", "WRITE_SIZE"} <= set(pair):
        out["_derived"] = {
            "pair_kernel_hbm_traffic_bytes_per_launch":
                (2 * pair["FETCH_SIZE"]["mean_per_launch"] + pair["WRITE_SIZE"]["mean_per_launch"]) * 1024,
            "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE x2 correction, MI355X_MICROARCH.md HBM section)",
        }
        if {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"} <= set(pair):
            out["_derived"]["pair_kernel_mfma_busy_fraction"] = (
                pair["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024
                / (pair["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8))
            out["_derived"]["note"] = "SQ_VALU_MFMA_BUSY_CYCLES summed over 1024 SIMDs / (GRBM_GUI_ACTIVE/8 XCDs)"
    # the aggregation kernel: counter traffic against its compulsory bytes is the figure to watch (wasted re-reads)
    sp = out.get("gm::k_spmm_seg<false, 8, 1>", {})
    if {"FETCH_SIZE", "WRITE_SIZE"} <= set(sp):
        out.setdefault("_derived", {})["spmm_seg_hbm_traffic_bytes_per_launch_mean"] = (
            2 * sp["FETCH_SIZE"]["mean_per_launch"] + sp["WRITE_SIZE"]["mean_per_launch"]) * 1024
    with open(prefix + "_pmc_summary.json", "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out.get("_derived"), indent=1))


if __name__ == "__main__":
    main()
