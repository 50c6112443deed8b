#!/bin/bash
# round 4: the chimera splitter's tests, its generations against each other on 0.9 M reads (3') and 0.3 M (5'), the microbench leg
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_chimera_gpu.py tests/test_ref_exec_gpu.py -x -q -m gpu 2>&1 | tail -4 || exit 1
timeout -k 10 300 python tools/chim_crosscheck.py 1000000 > gpurun_out/chim_cross_3p.json 2> gpurun_out/chim_cross_3p.err; echo "rc=$?"; cat gpurun_out/chim_cross_3p.json
timeout -k 10 300 python tools/chim_crosscheck.py 300000 5p > gpurun_out/chim_cross_5p.json 2> gpurun_out/chim_cross_5p.err; echo "rc=$?"; cat gpurun_out/chim_cross_5p.json
bash tools/gpu_r04_chim_trace.sh
