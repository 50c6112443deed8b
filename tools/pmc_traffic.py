#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a round's aggregated counters (tools/pmc_aggregate.py output), with the commit it was measured at.

usage: pmc_traffic.py profiles/r02/pass2_pmc.json <reads per launch> [more.json <units per launch> ...]
(further files add their kernels: profiles/r03/cfg2_pmc.json -> k_bc_match_ed2, k_hist_windows, k_scan22; profiles/r03/umi_pmc.json -> k_umi_dist, units = pairs)
HBM bytes per launch per kernel, as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE / WRITE_SIZE come in KiB from separate --pmc
passes; on gfx950 FETCH_SIZE counts 32-B-granule traffic of wide coalesced loads at half its value (the guide's correction: x2 for the
streaming kernel K-SCAN), while K-BC1's 4/8-byte gathers are outside the calibrated widths (no correction, FETCH_SIZE = TCC_MISS x 64 B there)."""
import json
import subprocess
import sys

commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
dst = "profiles/pmc_traffic.json"
try:
    out = json.load(open(dst))     # kernels measured in earlier calls stay (each entry names its source and commit)
except Exception:
    out = {}


def entry(k, units, fetch_scale, src):
    e = {"hbm_bytes_per_launch": int((fetch_scale * k["FETCH_SIZE"]["mean_per_launch"] + k["WRITE_SIZE"]["mean_per_launch"]) * 1024),
         "reads_per_launch": units, "fetch_kib": k["FETCH_SIZE"]["mean_per_launch"], "write_kib": k["WRITE_SIZE"]["mean_per_launch"],
         "fetch_correction": fetch_scale, "source": src, "commit": commit}
    for c, name in (("SQ_INSTS_VALU", "valu_insts_per_launch"), ("SQ_INSTS_SALU", "salu_insts_per_launch"), ("TCC_MISS_sum", "l2_misses_per_launch")):
        if c in k:
            e[name] = int(k[c]["mean_per_launch"])
    return e


args = sys.argv[1:]
for src, units in zip(args[0::2], args[1::2]):
    units = int(units)
    d = json.load(open(src))
    if "k_bc_codes_ed1t" in d:  # K-BC1 on its table path is two kernels: their counters are added up under the old key
        a, b = d["k_bc_codes_ed1t"], d["k_bc_pick_ed1t"]
        d["k_bc_match_ed1"] = {c: {"mean_per_launch": a[c]["mean_per_launch"] + b[c]["mean_per_launch"]} for c in a if c in b}
    # fetch correction: x2 for the kernels that stream wide coalesced loads (K-SCAN); none for the gather kernels (FETCH_SIZE = TCC_MISS x 64 B
    # there: K-BC1, K-BC2, K-HIST) nor for K-UMI (8-byte window loads, traffic is its matrix stores)
    for key, fetch_scale in (("k_scan", 2.0), ("k_bc_match_ed1", 1.0), ("k_bc_match_ed2", 1.0), ("k_hist_windows", 1.0), ("k_umi_dist", 1.0)):
        if key not in d or "FETCH_SIZE" not in d[key]:
            continue
        name = key
        if key == "k_scan" and "cfg2" in src:
            name = "k_scan_cfg2_mean_of_22_and_10"   # both instantiations match the pattern in the configs[2] run
        out[name] = entry(d[key], units, fetch_scale, src)
out.pop("source", None)   # every kernel entry names its own source file and commit
out["commit"] = commit
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
