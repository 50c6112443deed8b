#!/bin/bash
set -u
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_bam.py tests/test_bc_gpu.py tests/test_bench_gpu.py tests/test_capi_gpu.py tests/test_chimera_gpu.py -x -q -m gpu > gpurun_out/r04_dbg.log 2>&1
echo "rc=$?"; head -c 1500 gpurun_out/r04_dbg.log; echo ...; tail -c 600 gpurun_out/r04_dbg.log
