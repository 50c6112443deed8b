#!/bin/bash
# counters of the splitter microbench (K-PACKR, K-CHIM-A/B/C)
set -u
PROFILE_PROG=$PWD/tools/microbench.py bash tools/profile_gpu.sh r02chim chimera 2>&1 | tail -3
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/summary_r02chim/r02chim_pmc.json"))
for k, v in d.items():
    print(k, {a: (round(b["mean_per_launch"]) if isinstance(b, dict) else b) for a, b in v.items()})
PY
cat gpurun_out/summary_r02chim/r02chim_kernel_stats.csv | cut -c1-160
