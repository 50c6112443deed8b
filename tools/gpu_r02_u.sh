#!/bin/bash
# K-BC2 with the item filter: parity suites with and without it, then configs[2] at 20 M reads
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_bc_gpu.py tests/test_ref_exec_gpu.py tests/test_capi_gpu.py -m gpu -x -q > gpurun_out/gputests_u.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputests_u.log
SMI_BC2_NO_FILTER=1 timeout -k 10 900 python -m pytest tests/test_bc_gpu.py -m gpu -x -q -k "ed2" > gpurun_out/gputests_u2.log 2>&1; echo "pytest(no filter) rc=$?"; tail -2 gpurun_out/gputests_u2.log
timeout -k 10 900 python bench.py --config 2 --reads 20000000 > gpurun_out/bench_cfg2_u.json 2> gpurun_out/bench_cfg2_u.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/bench_cfg2_u.json
