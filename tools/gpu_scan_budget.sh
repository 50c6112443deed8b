#!/bin/bash
# K-SCAN's instruction budget by region -- the step with parts of the kernel switched off (measurement build, wrong results by
# construction): SQ_INSTS_VALU and the kernel's duration per variant -> gpurun_out/scan_budget.json
set -u
TAG=${TAG:-r06}   # the round the outputs are named after (profiles/$TAG/ once copied there)
ulimit -c 0
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/scanb
export SMI_LIBRARY=$ROOT/sicelore-2.1_amd/csrc/libsicelore_mi_measure.so
OFF="--steps 3 --warmup 1 --no-cpu-baseline --two-pass-reads 0 --e2e-reads 0 --umi-molecules 0 --h2h-reads 0 --f2f-reads 0 --assignumis-file-records 0"
cd /tmp
for a in ${ABLATES:-0 1 2 4 8 16 9 15}; do
  export SMI_SCAN_ABLATE=$a
  rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/scanb/t$a" -- python3 $ROOT/bench.py $OFF > "$ROOT/gpurun_out/scanb/t$a.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$ROOT/gpurun_out/scanb/p$a" -- python3 $ROOT/bench.py $OFF > "$ROOT/gpurun_out/scanb/p$a.log" 2>&1
  echo "ablate $a done"
done
cd "$ROOT"
python3 - <<'PY'
import csv, glob, json
out = {}
import os
for a in [int(x) for x in os.environ.get("ABLATES", "0 1 2 4 8 16 9 15").split()]:
    e = {}
    for f in glob.glob(f"gpurun_out/scanb/t{a}/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "smi::k_scan<" in row["Name"]:
                e["kernel_ms"] = float(row["AverageNs"]) / 1e6
                e["calls"] = int(row["Calls"])
    agg = {}
    for f in glob.glob(f"gpurun_out/scanb/p{a}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "smi::k_scan<" in row["Kernel_Name"]:
                d = agg.setdefault(row["Counter_Name"], {})
                d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    for c, d in agg.items():
        e[c] = sum(d.values()) / len(d)
    out[str(a)] = e
json.dump(out, open("gpurun_out/scan_budget.json", "w"), indent=1)
print(json.dumps(out))
PY
find gpurun_out/scanb -name "*.csv" -size +1M -delete
