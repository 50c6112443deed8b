#!/bin/bash
# round 4: kernel trace of the bench's end-to-end leg alone -> gpurun_out/e2e_kernel_stats.csv, gpurun_out/e2e_timeline.json
set -u
ulimit -c 0
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
OFF="--umi-molecules 0 --h2h-reads 0 --f2f-reads 0 --assignumis-file-records 0"
rm -rf gpurun_out/prof_e2e
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_e2e" -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --two-pass-reads 0 $OFF > "$ROOT/gpurun_out/prof_e2e.log" 2>&1
cd "$ROOT"
f=$(find gpurun_out/prof_e2e -name "*kernel_stats.csv" | head -1)
(head -1 "$f"; grep "smi::" "$f") > gpurun_out/e2e_kernel_stats.csv
python3 tools/e2e_timeline.py gpurun_out/prof_e2e gpurun_out/e2e_timeline.json
find gpurun_out/prof_e2e -name "*.csv" -size +1M -delete
tail -c 1500 gpurun_out/prof_e2e.log | grep -o '"end_to_end": {.*' | cut -c1-300
