#!/bin/bash
set -u
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_cli_gpu.py tests/test_chimera_gpu.py -x -q -m gpu > gpurun_out/r04_cli.log 2>&1; echo "rc=$?"; tail -30 gpurun_out/r04_cli.log | cut -c1-600
timeout -k 10 200 python tools/microbench.py chimera 2>/dev/null | grep -o '"chimera": {[^}]*}' | cut -c1-200
