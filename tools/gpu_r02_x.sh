#!/bin/bash
# K-BC2 with level 2 from the neighbourhood table: matcher suites + reference-executed vectors, micro bc, configs[2] at 20 M reads
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_bc_gpu.py tests/test_ref_exec_gpu.py tests/test_capi_gpu.py -m gpu -x -q > gpurun_out/gputests_x.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputests_x.log
timeout -k 10 600 python tools/microbench.py bc 2>/dev/null | tail -1 | cut -c1-700
timeout -k 10 900 python bench.py --config 2 --reads 20000000 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms'])"
