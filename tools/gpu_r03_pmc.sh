#!/bin/bash
# round 3, call a: what the host side of the box offers (cores, memory bandwidth), the RCCL entry after its rewrite, and the HBM counters
# of the configs[2] kernels (K-BC2, K-HIST) and of K-UMI -> gpurun_out/summary_r03cfg2, summary_r03umi
set -u
mkdir -p gpurun_out
python tools/host_probe.py > gpurun_out/host_probe.json 2>&1; cat gpurun_out/host_probe.json
timeout -k 10 600 python -m pytest tests/test_capi_gpu.py -x -q -m gpu 2>&1 | tail -3
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
  timeout -k 10 900 bash tools/profile_gpu.sh r03cfg2 --config 2 --reads 10000000 --steps 2 --warmup 1 2>&1 | tail -5
PROFILE_PROG=$PWD/tools/microbench.py PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
  timeout -k 10 600 bash tools/profile_gpu.sh r03umi umi 2>&1 | tail -5
