#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a round's aggregated counters (tools/pmc_aggregate.py output), with the commit it was measured at.

usage: pmc_traffic.py profiles/r02/pass2_pmc.json <reads per launch>
HBM bytes per launch per kernel, as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE / WRITE_SIZE come in KiB from separate --pmc
passes; on gfx950 FETCH_SIZE counts 32-B-granule traffic of wide coalesced loads at half its value (the guide's correction: x2 for the
streaming kernel K-SCAN), while K-BC1's 4/8-byte gathers are outside the calibrated widths (no correction, FETCH_SIZE = TCC_MISS x 64 B there)."""
import json
import subprocess
import sys

src, reads = sys.argv[1], int(sys.argv[2])
d = json.load(open(src))
commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
out = {"commit": commit, "source": src}
if "k_bc_codes_ed1t" in d:  # K-BC1 on its table path is two kernels: their counters are added up under the old key
    a, b = d["k_bc_codes_ed1t"], d["k_bc_pick_ed1t"]
    d["k_bc_match_ed1"] = {c: {"mean_per_launch": a[c]["mean_per_launch"] + b[c]["mean_per_launch"]} for c in a if c in b}
for key, fetch_scale in (("k_scan", 2.0), ("k_bc_match_ed1", 1.0)):
    k = d[key]
    out[key] = {
        "hbm_bytes_per_launch": int((fetch_scale * k["FETCH_SIZE"]["mean_per_launch"] + k["WRITE_SIZE"]["mean_per_launch"]) * 1024),
        "reads_per_launch": reads,
        "valu_insts_per_launch": int(k["SQ_INSTS_VALU"]["mean_per_launch"]),
        "salu_insts_per_launch": int(k["SQ_INSTS_SALU"]["mean_per_launch"]),
        "fetch_kib": k["FETCH_SIZE"]["mean_per_launch"], "write_kib": k["WRITE_SIZE"]["mean_per_launch"], "fetch_correction": fetch_scale,
    }
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
