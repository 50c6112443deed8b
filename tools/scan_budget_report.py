#!/usr/bin/env python3
"""gpurun_out/scan_budget.json (tools/gpu_scan_budget.sh with ABLATES="0 1 2 4 8 16 64 66 128 9 15") -> profiles/r06/k_scan_budget.json (round 4: profiles/r04/):
K-SCAN's VALU instructions and time by region.  usage: scan_budget_report.py [raw.json] [out.json]"""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/scan_budget.json"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r06/k_scan_budget.json"
raw = json.load(open(src))
json.dump(raw, open(dst.replace(".json", "_raw.json"), "w"), indent=1)
W = 312500                                  # waves per launch: 10 M reads = 20 M ends / 64
v = lambda a: raw[str(a)]["SQ_INSTS_VALU"]  # noqa: E731
ms = lambda a: raw[str(a)]["kernel_ms"]     # noqa: E731
tot = v(0)
regions = {
    "TSO alignments (16 x 16 band: fill 184 cells x 7 operations + bit-parallel walk)": v(0) - v(1),
    "adapter alignments (10 x 10 band: fill + bit-parallel walk with the end-of-read statistics + fold)": v(0) - v(2),
    "polyT finder (bit-parallel, round 4)": v(64) - v(4),
    "adapter 4-mer gates + candidate queue": v(2) - v(66),
    "TSO 4-mer gates + isolated-candidate pre-filter + queue": v(1) - v(9),
    "staging of the planes, strand decision, TSO rules, records": v(15),
}
out = {
    "what": "K-SCAN k_scan<10,false,true> (bit-parallel polyT finder), 10 M reads per launch (312,500 waves of 64 read ends): SQ_INSTS_VALU and duration with "
            "parts of the kernel switched off (measurement build, SMI_SCAN_ABLATE; tools/gpu_scan_budget.sh, tools/scan_budget_report.py)",
    "raw": dst.replace(".json", "_raw.json") + " (ablate value -> counters; 1 no TSO alignments, 2 no adapter alignments, 4 no polyT finder (no adapter scan follows), "
           "8 no TSO gates (no TSO scan follows), 16 no TSO pre-filter, 64 no adapter gates, 128 no folds; sums of those)",
    "total_valu_wave_instructions": tot, "per_wave": tot / W, "kernel_ms": ms(0),
    "regions": {k: {"valu_wave_instructions": x, "share": round(x / tot, 3), "per_wave": round(x / W)} for k, x in regions.items()},
    "unattributed_interaction": tot - sum(regions.values()),
    "folds_inside_the_regions_above": {"valu_wave_instructions": v(0) - v(128), "per_wave": round((v(0) - v(128)) / W)},
    "ms_with_region_off": {"no TSO alignments": ms(1), "no adapter alignments": ms(2), "no finder / adapter scan": ms(4), "no TSO scan": ms(8), "no adapter gates": ms(64),
                           "nothing but staging + decision + records": ms(15), "TSO pre-filter off (more alignments)": ms(16)},
    "before_the_bit_parallel_walk (round 4 / 5, profiles/r04/k_scan_budget.json)": {
        "kernel_ms": 3.13, "per_wave": 5779, "tso_alignments_per_wave": 2252, "adapter_alignments_per_wave": 987, "salu_wave_instructions": 664085423,
        "note": "the walk went step by step through ~40 operations of bookkeeping, once per left move for the whole wave; as two step masks and popcounts "
                "(smi_nw.h nw_walk_bits) it is ~18 operations per row and ~70 - 110 after the rows; the scalar unit's share fell from 664 M to ~260 M instructions "
                "(the loops' branches)"},
    "reading": [
        "issue rate: %.0f G wave-instructions/s over the whole kernel = %.0f %% of the 601 G/s ceiling of 4-cycle forms" % (tot / ms(0) / 1e6, 100 * tot / ms(0) / 1e6 / 601.4),
        "what an alignment needs: the fill is 7 operations per cell (bit extract, multiply-add, two subtractions, max3, and-or, tag shift) over the exact band "
        "(smi_nw.h `Band`: 184 cells of 256 for the 16-base TSO, 44 of 100 for the 10-base adapter) = 1,290 + 310 per wave for one round of each; the candidates are what "
        "the reference's gate lets through minus the isolated ones under the Levenshtein bound (one round of each per wave: NOTES R5.13, R5.16); the walk is one "
        "find-highest-bit per row.  No formulation of the cell with fewer than seven operations was found (DESIGN section 4)",
        "with every scan off the kernel takes %.2f ms for %.0f M instructions: the time of moving 2.78 GB (ends in, records and windows out) -- the HBM floor of this "
        "launch; the scans hide it" % (ms(15), v(15) / 1e6)]}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: (r["per_wave"], r["share"]) for k, r in out["regions"].items()}))
