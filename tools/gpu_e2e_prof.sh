#!/bin/bash
# kernel trace of the bench with its end-to-end leg (pass 2, text in HBM -> text in HBM); summary -> gpurun_out/e2e_kernel_stats.csv
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_e2e" -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --two-pass-reads 0 > "$ROOT/gpurun_out/prof_e2e.log" 2>&1
cd "$ROOT"
f=$(find gpurun_out/prof_e2e -name "*kernel_stats.csv" | head -1)
(head -1 "$f"; grep "smi::" "$f") > gpurun_out/e2e_kernel_stats.csv
find gpurun_out/prof_e2e -name "*.csv" -size +1M -delete
python3 - <<'PY'
import csv
for row in csv.DictReader(open("gpurun_out/e2e_kernel_stats.csv")):
    nm = row["Name"].split("(")[0][-48:]
    print(f'{nm:50s} calls {row["Calls"]:>3s} avg {float(row["AverageNs"])/1e6:7.3f} min {float(row["MinNs"])/1e6:7.3f} ms')
PY
tail -c 600 gpurun_out/prof_e2e.log | grep -o '"end_to_end": {[^}]*}' | cut -c1-300
