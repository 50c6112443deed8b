#!/bin/bash
# round 4: the traces and counters the bench line cites, on the build it describes
#   profiles/r04/step_*      the timed step alone (K-SCAN + K-BC1, 10 M reads, every side leg off): kernel trace + counters
#   profiles/r04/chimera_*   the splitter's microbench leg (0.9 M reads): kernel trace + counters
#   profiles/r04/e2e_*       the bench's end-to-end leg alone: kernel trace
set -u
ulimit -c 0
mkdir -p gpurun_out
OFF="--umi-molecules 0 --h2h-reads 0 --f2f-reads 0 --assignumis-file-records 0"
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
  timeout -k 10 900 bash tools/profile_gpu.sh r04step --steps 5 --warmup 2 $OFF 2>&1 | tail -3
PROFILE_PROG=$PWD/tools/microbench.py PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES;SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" \
  timeout -k 10 900 bash tools/profile_gpu.sh r04chimera chimera 2>&1 | tail -3
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_e2e" -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --two-pass-reads 0 $OFF > "$ROOT/gpurun_out/prof_e2e.log" 2>&1
cd "$ROOT"
f=$(find gpurun_out/prof_e2e -name "*kernel_stats.csv" | head -1)
(head -1 "$f"; grep "smi::" "$f") > gpurun_out/e2e_kernel_stats.csv
python3 tools/e2e_timeline.py gpurun_out/prof_e2e gpurun_out/e2e_timeline.json
find gpurun_out/prof_e2e -name "*.csv" -size +1M -delete
python3 - <<'PY'
import csv
for row in csv.DictReader(open("gpurun_out/e2e_kernel_stats.csv")):
    nm = row["Name"].split("(")[0][-44:]
    print(f'{nm:46s} calls {row["Calls"]:>3s} avg {float(row["AverageNs"])/1e6:7.3f} min {float(row["MinNs"])/1e6:7.3f} ms')
PY
tail -c 1500 gpurun_out/prof_e2e.log | grep -o '"end_to_end": {.*' | cut -c1-1200
# the device clusterer of large UMI groups: one 8,000-read group, device against host
SMI_AU_TIMING=1 timeout -k 10 300 python tools/own_cluster_bench.py 8000 > gpurun_out/own_cluster_8000.json 2> gpurun_out/own_cluster_8000.err; echo "own rc=$?"; cat gpurun_out/own_cluster_8000.json
# the splitter's microbench line itself (no profiler attached)
timeout -k 10 300 python tools/microbench.py chimera > gpurun_out/microbench_chimera.json 2> gpurun_out/microbench_chimera.err; echo "mb rc=$?"; cut -c1-600 gpurun_out/microbench_chimera.json
