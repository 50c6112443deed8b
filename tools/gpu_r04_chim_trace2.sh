#!/bin/bash
# kernel trace of tools/chim_crosscheck.py (args: reads [5p]) -> stdout
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_chimx" -- python3 $ROOT/tools/chim_crosscheck.py "$@" > "$ROOT/gpurun_out/prof_chimx.log" 2>&1
cd "$ROOT"
f=$(find gpurun_out/prof_chimx -name "*kernel_stats.csv" | head -1)
(head -1 "$f"; grep "smi::" "$f") > gpurun_out/chimx_trace_stats.csv
find gpurun_out/prof_chimx -name "*.csv" -size +1M -delete
python3 - <<'PY'
import csv
for row in csv.DictReader(open("gpurun_out/chimx_trace_stats.csv")):
    nm = row["Name"].split("(")[0][-40:]
    print(f'{nm:42s} calls {row["Calls"]:>3s} avg {float(row["AverageNs"])/1e6:7.3f} min {float(row["MinNs"])/1e6:7.3f} max {float(row["MaxNs"])/1e6:7.3f} ms')
PY
tail -2 gpurun_out/prof_chimx.log | cut -c1-400
