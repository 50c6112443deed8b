#!/bin/bash
# round 3, call i: the profiles the bench line rests on -- kernel trace of the default bench command, configs[2] at 100 M reads with the
# counter traffic of this round, --config 4
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_r03_bench" -- python3 $ROOT/bench.py --steps 5 --warmup 1 --f2f-reads 0 > "$ROOT/gpurun_out/prof_r03_bench.log" 2>&1
cd "$ROOT"
f=$(find gpurun_out/prof_r03_bench -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && (head -1 "$f"; grep -E "smi::" "$f") > gpurun_out/prof_r03_bench_kernel_stats.csv
find gpurun_out/prof_r03_bench -name "*.csv" -size +1M -delete
cut -c1-150 gpurun_out/prof_r03_bench_kernel_stats.csv | head -12
timeout -k 10 900 python bench.py --config 2 --reads 100000000 --steps 2 --warmup 1 > gpurun_out/bench_cfg2_100m.json 2> gpurun_out/bench_cfg2.err; tail -2 gpurun_out/bench_cfg2.err; cut -c1-600 gpurun_out/bench_cfg2_100m.json
timeout -k 10 600 python bench.py --config 4 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err; cut -c1-400 gpurun_out/bench_cfg4.json
