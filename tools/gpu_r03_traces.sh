#!/bin/bash
# round 3, call f: kernel trace of the device UMI stage (tools/microbench.py assignumis) and of the packed chunk worker
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_r03_umi_stage" -- python3 $ROOT/tools/microbench.py assignumis > "$ROOT/gpurun_out/prof_r03_umi_stage.log" 2>&1
SMI_MB_READS=200000 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_r03_packed" -- python3 $ROOT/tools/microbench.py packed > "$ROOT/gpurun_out/prof_r03_packed.log" 2>&1
cd "$ROOT"
for d in prof_r03_umi_stage prof_r03_packed; do
  f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && (head -1 "$f"; grep -E "smi::|hipcub|rocprim" "$f") > gpurun_out/${d}_kernel_stats.csv
  find gpurun_out/$d -name "*.csv" -size +1M -delete
  cut -c1-230 gpurun_out/${d}_kernel_stats.csv | head -30
done
