#!/usr/bin/env python3
"""Aggregate rocprofv3 outputs (one directory per --pmc pass, plus one --kernel-trace --stats pass) into the small
summaries kept under profiles/: mean counter value per launch for our kernels, and the kernel_stats rows.

usage: pmc_aggregate.py <rocprof_out_dir> <dest_dir> <tag>
"""
import csv
import glob
import json
import os
import sys

KERNELS = {"k_scan": "smi::k_scan", "k_bc_match_ed1": "smi::k_bc_match_ed1", "k_bc_codes_ed1t": "smi::k_bc_codes_ed1t", "k_bc_pick_ed1t": "smi::k_bc_pick_ed1t", "k_bc_match_ed2": "smi::k_bc_match_ed2",
           "k_pack_ends": "smi::k_pack_ends", "k_umi_dist": "smi::k_umi_dist<", "k_umi_dist_tiles": "smi::k_umi_dist_tiles<", "k_hist_windows": "smi::k_hist_windows", "k_chimera": "smi::k_chimera", "k_pack_reads": "smi::k_pack_reads",
           "k_chim_filter": "smi::k_chim_tso_filter", "k_chimera_B": "smi::k_chimera<27, 22, 1>", "k_chimera_C": "smi::k_chimera<27, 22, 2>",
           "k_write": "smi::k_write(", "k_write_name": "smi::k_write_name", "k_fq_lines": "smi::k_fq_lines",
           "k_deflate_blocks": "k_deflate_blocks", "k_deflate_gather": "k_deflate_gather", "k_umi_parse": "smi::k_umi_parse", "k_umi_cluster": "smi::k_umi_cluster",
           "k_ends_from_planes": "smi::k_ends_from_planes", "k_inflate": "smi::k_inflate(",
           "k_chima_flat": "smi::k_chima_flat", "k_chima_finish": "smi::k_chima_finish", "k_chim_owner": "smi::k_chim_owner", "k_chimb_select2": "smi::k_chimb_select2",
           "k_chimb_align": "smi::k_chimb_align", "k_chimb_fold": "smi::k_chimb_fold", "k_chimc_walk": "smi::k_chimc_walk", "k_chimc_gate": "smi::k_chimc_gate",
           "k_chimc_align": "smi::k_chimc_align", "k_chimc_rules": "smi::k_chimc_rules", "k_umi_cluster_own": "smi::k_umi_own"}


def main():
    src, dst, tag = sys.argv[1:4]
    os.makedirs(dst, exist_ok=True)
    agg = {}
    for path in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                kname = row["Kernel_Name"]
                for short, pat in KERNELS.items():
                    if pat in kname:
                        # one row per (dispatch, counter[, dimension instance]): sum instances of one dispatch
                        key = (short, row["Counter_Name"])
                        d = agg.setdefault(key, {})
                        d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    out = {}
    for (short, cname), d in sorted(agg.items()):
        vals = list(d.values())
        out.setdefault(short, {})[cname] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
    with open(os.path.join(dst, f"{tag}_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    for path in glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True):
        with open(path, newline="") as f:
            rows = list(csv.reader(f))
        keep = [rows[0]] + [r for r in rows[1:] if "smi::" in r[0]]
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            csv.writer(f).writerows(keep)
    print(json.dumps({k: {c: v["mean_per_launch"] for c, v in d.items() if c in ("FETCH_SIZE", "WRITE_SIZE")}
                      for k, d in out.items()}))


if __name__ == "__main__":
    main()
