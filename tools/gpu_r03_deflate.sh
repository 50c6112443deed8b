#!/bin/bash
# round 3: kernel trace of K-DEFLATE (tools/microbench.py deflate: 0.5 M reads = 1.2 GB of FASTQ text per call)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_r03_deflate" -- python3 $ROOT/tools/microbench.py deflate > "$ROOT/gpurun_out/prof_r03_deflate.log" 2>&1
cd "$ROOT"
f=$(find gpurun_out/prof_r03_deflate -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && (head -1 "$f"; grep -E "smi::|hipcub|rocprim" "$f") > gpurun_out/deflate_kernel_stats.csv
find gpurun_out/prof_r03_deflate -name "*.csv" -size +1M -delete
cut -c1-200 gpurun_out/deflate_kernel_stats.csv | head -12
tail -1 gpurun_out/prof_r03_deflate.log | cut -c1-400
