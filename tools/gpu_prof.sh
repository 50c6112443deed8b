#!/bin/bash
# round-2 profiles: the bench step (kernel trace + PMC passes), then the chimera / fastq microbench kernel traces
set -u
bash tools/profile_gpu.sh r02 2>&1 | tail -4
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_r02_chim" -- python3 $ROOT/tools/microbench.py chimera > "$ROOT/gpurun_out/prof_r02_chim.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_r02_e2e" -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --two-pass-reads 0 > "$ROOT/gpurun_out/prof_r02_e2e.log" 2>&1
cd "$ROOT"
for d in prof_r02_chim prof_r02_e2e; do
  f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && (head -1 "$f"; grep "smi::" "$f") > gpurun_out/${d}_kernel_stats.csv
  find gpurun_out/$d -name "*.csv" -size +1M -delete
done
tail -3 gpurun_out/prof_r02_chim.log | cut -c1-300
cat gpurun_out/prof_r02_chim_kernel_stats.csv | cut -c1-200
