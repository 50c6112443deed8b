#!/bin/bash
set -u
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ref_exec_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "not pass2w_3p_ed2" > gpurun_out/r04_t1.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r04_t1.log | cut -c1-700
