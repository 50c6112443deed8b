#!/bin/bash
# round 3, the last build: kernel trace of the default bench command (the assignumis file leg included)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_r03l_bench" -- python3 $ROOT/bench.py --steps 20 --warmup 5 --f2f-reads 0 > "$ROOT/gpurun_out/prof_r03l_bench.log" 2>&1
cd "$ROOT"
f=$(find gpurun_out/prof_r03l_bench -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && (head -1 "$f"; grep -E "smi::|hipcub|rocprim" "$f") > gpurun_out/r03l_bench_kernel_stats.csv
find gpurun_out/prof_r03l_bench -name "*.csv" -size +1M -delete
cut -c1-170 gpurun_out/r03l_bench_kernel_stats.csv | grep -E "Name|k_scan<10|k_bc_match_ed1<1>|k_deflate_blocks|k_umi_parse" | head -8
tail -c 600 gpurun_out/prof_r03l_bench.log | head -c 400
