#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_chimera_gpu.py tests/test_ref_exec_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 300 python tools/chim_crosscheck.py 300000 5p 2> gpurun_out/chim_cross5.err | cut -c1-700
timeout -k 10 300 python tools/chim_crosscheck.py 1000000 2> gpurun_out/chim_cross3.err | cut -c1-700
bash tools/gpu_r04_chim_trace.sh
