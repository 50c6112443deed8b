#!/bin/bash
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_chimera_gpu.py -m gpu -x -q 2>&1 | tail -2
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_chim_r" -- python3 $ROOT/tools/microbench.py chimera > "$ROOT/gpurun_out/prof_chim_r.log" 2>&1
cd "$ROOT"
f=$(find gpurun_out/prof_chim_r -name "*kernel_stats.csv" | head -1)
grep "smi::" "$f" | awk -F'","|",' '{print substr($1,1,60), $2, $4}' | cut -c1-120
find gpurun_out/prof_chim_r -name "*.csv" -size +1M -delete
