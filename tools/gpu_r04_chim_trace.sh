#!/bin/bash
# kernel trace of the chimera splitter's microbench leg -> gpurun_out/chim_trace_stats.csv
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_chim" -- python3 $ROOT/tools/microbench.py chimera > "$ROOT/gpurun_out/prof_chim.log" 2>&1
cd "$ROOT"
f=$(find gpurun_out/prof_chim -name "*kernel_stats.csv" | head -1)
(head -1 "$f"; grep "smi::" "$f") > gpurun_out/chim_trace_stats.csv
find gpurun_out/prof_chim -name "*.csv" -size +1M -delete
python3 - <<'PY'
import csv
for row in csv.DictReader(open("gpurun_out/chim_trace_stats.csv")):
    nm = row["Name"].split("(")[0][-40:]
    print(f'{nm:42s} calls {row["Calls"]:>3s} avg {float(row["AverageNs"])/1e6:7.3f} min {float(row["MinNs"])/1e6:7.3f} ms')
PY
grep -o '"chimera": {[^}]*}' gpurun_out/prof_chim.log | cut -c1-160
