#!/bin/bash
# the traces and counters the bench line cites, on the build it describes
#   profiles/$TAG/step_*      the timed step alone (K-SCAN + K-BC1, 10 M reads, every side leg off): kernel trace + counters
#   profiles/$TAG/chimera_*   the splitter's microbench leg (0.9 M reads): kernel trace + counters
#   profiles/$TAG/umi_*       K-UMI's microbench leg: kernel trace + counters
#   profiles/$TAG/e2e_*       the bench's end-to-end leg alone: kernel trace, timeline, traffic counters
set -u
TAG=${TAG:-r06}   # the round the outputs are named after (profiles/$TAG/ once copied there)
ulimit -c 0
mkdir -p gpurun_out
OFF="--umi-molecules 0 --h2h-reads 0 --f2f-reads 0 --assignumis-file-records 0"
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
  timeout -k 10 900 bash tools/profile_gpu.sh ${TAG}step --steps 5 --warmup 2 $OFF 2>&1 | tail -3
PROFILE_PROG=$PWD/tools/microbench.py PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES;SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" \
  timeout -k 10 900 bash tools/profile_gpu.sh ${TAG}chimera chimera 2>&1 | tail -3
# K-UMI: the microbench leg, kernel trace + write counters -> gpurun_out/summary_${TAG}umi/
PROFILE_PROG=$PWD/tools/microbench.py PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES;SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  timeout -k 10 900 bash tools/profile_gpu.sh ${TAG}umi umi 2>&1 | tail -n 3 | cut -c1-400
timeout -k 10 300 python tools/microbench.py umi > gpurun_out/microbench_umi.json 2> gpurun_out/microbench_umi.err; echo "umi mb rc=$?"
# the end-to-end leg alone: kernel trace + timeline, then FETCH_SIZE / WRITE_SIZE per kernel
bash tools/gpu_part.sh e2e
export TMPDIR=/tmp
ROOT=$(pwd)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d "$ROOT/gpurun_out/prof_e2e_pmc/$c" -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --two-pass-reads 0 --e2e-lanes 1 $OFF > "$ROOT/gpurun_out/prof_e2e_pmc_$c.log" 2>&1) || echo "pass $c failed"
done
python3 - <<'PY'
import csv, glob, json, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/prof_e2e_pmc/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c or "smi::" not in r["Kernel_Name"]:
                continue
            nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smi::", "")
            acc[nm].append(float(r["Counter_Value"]))
    for nm, v in acc.items():
        v = v[len(v) // 2:]     # the later dispatches: the timed repetitions
        res[nm][c + "_KB_per_launch"] = sum(v) / len(v)
        res[nm]["launches"] = len(v)
json.dump(res, open("gpurun_out/e2e_pmc.json", "w"), indent=1)
PY
find gpurun_out/prof_e2e_pmc -name "*.csv" -size +1M -delete
# the device clusterer of large UMI groups: one 8,000-read group, device against host
SMI_AU_TIMING=1 timeout -k 10 300 python tools/own_cluster_bench.py 8000 > gpurun_out/own_cluster_8000.json 2> gpurun_out/own_cluster_8000.err; echo "own rc=$?"; cat gpurun_out/own_cluster_8000.json
# the splitter's microbench line itself (no profiler attached)
timeout -k 10 300 python tools/microbench.py chimera > gpurun_out/microbench_chimera.json 2> gpurun_out/microbench_chimera.err; echo "mb rc=$?"; cut -c1-600 gpurun_out/microbench_chimera.json
