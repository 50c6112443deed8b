#!/bin/bash
# round 4, call 0: the GPU suite on the tree as round 3 left it, then kernel trace + counters of the chimera splitter (microbench leg)
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests0.log 2>&1; tail -3 gpurun_out/r04_gpu_tests0.log
PROFILE_PROG=$PWD/tools/microbench.py PMC_GROUPS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES;SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA;SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU;FETCH_SIZE;WRITE_SIZE" \
  timeout -k 10 900 bash tools/profile_gpu.sh r04chim0 chimera 2>&1 | tail -5
