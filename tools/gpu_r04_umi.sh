#!/bin/bash
# round 4: K-UMI with tiles for the large groups -- its tests, the microbench leg, kernel trace and the write counters
set -u
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_umi_gpu.py tests/test_umi_stage_gpu.py -x -q -m gpu > gpurun_out/r04_umi_tests.log 2>&1; rc=$?; echo "rc=$rc"; tail -6 gpurun_out/r04_umi_tests.log | cut -c1-400
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/microbench.py umi > gpurun_out/microbench_umi.json 2> gpurun_out/microbench_umi.err; echo "mb rc=$?"; cut -c1-600 gpurun_out/microbench_umi.json
PROFILE_PROG=$PWD/tools/microbench.py PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
  timeout -k 10 900 bash tools/profile_gpu.sh r04umi umi 2>&1 | tail -4 | cut -c1-600
