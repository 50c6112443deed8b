#!/bin/bash
# round 3: HBM counters and instruction counts of K-DEFLATE (tools/microbench.py deflate) -> gpurun_out/summary_r03deflate
set -u
mkdir -p gpurun_out
PROFILE_PROG=$PWD/tools/microbench.py PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  timeout -k 10 800 bash tools/profile_gpu.sh r03deflate deflate 2>&1 | tail -4
ls gpurun_out/summary_r03deflate; cat gpurun_out/summary_r03deflate/*pmc*.json | head -c 3000
