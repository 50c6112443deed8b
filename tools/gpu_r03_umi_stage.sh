#!/bin/bash
# round 3, call g: the device UMI stage after its two rewrites (K-UPARSE rolling window, K-UCLUST in LDS), file-to-file test
set -u
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_umi_stage_gpu.py tests/test_umi_gpu.py tests/test_run_files_gpu.py -x -q -m gpu 2>&1 | tail -15 || exit 1
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_r03_umi_stage" -- python3 $ROOT/tools/microbench.py assignumis > "$ROOT/gpurun_out/mb_assignumis.json" 2> "$ROOT/gpurun_out/prof_r03_umi_stage.log"
cd "$ROOT"
f=$(find gpurun_out/prof_r03_umi_stage -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && (head -1 "$f"; grep -E "smi::|hipcub|rocprim" "$f") > gpurun_out/prof_r03_umi_stage_kernel_stats.csv
find gpurun_out/prof_r03_umi_stage -name "*.csv" -size +1M -delete
python3 - <<'P'
import csv, json
rows = list(csv.reader(open("gpurun_out/prof_r03_umi_stage_kernel_stats.csv")))
for r in rows[1:10]:
    print("  %-60s calls %5s avg %10.1f us  total %8.1f ms" % (r[0][:60], r[1], float(r[3]) / 1e3, float(r[2]) / 1e6))
d = json.load(open("gpurun_out/mb_assignumis.json"))
print({k: round(v["records_per_s"] / 1e6, 1) for k, v in d.items() if "records_per_s" in v}, d["assignumis_device_stage_lanes"]["runs"])
P
