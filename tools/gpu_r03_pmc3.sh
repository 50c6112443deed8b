#!/bin/bash
# round 3: instruction and wait counters of K-INFLATE (tools/microbench.py inflate, 100 k reads in 64 files) -> gpurun_out/summary_r03inflate
set -u
mkdir -p gpurun_out
SMI_MB_READS=100000 PROFILE_PROG=$PWD/tools/microbench.py PMC_GROUPS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES;SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA;SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
  timeout -k 10 800 bash tools/profile_gpu.sh r03inflate inflate 2>&1 | tail -3
python3 - <<'PY'
import json,glob
for f in glob.glob('gpurun_out/summary_r03inflate/*pmc*.json'):
    d=json.load(open(f))
    for k,v in d.items():
        if isinstance(v,dict) and 'inflate' in k: print(k,{c:round(x['mean_per_launch']) for c,x in v.items() if isinstance(x,dict)})
PY
