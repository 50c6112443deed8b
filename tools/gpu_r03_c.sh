#!/bin/bash
# round 3, call c: the rest of the packed-boundary parity tests, then the host stages one by one (malloc'ed vs page-locked buffers, threads)
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_packed_gpu.py tests/test_ref_exec_gpu.py tests/test_write_gpu.py -q -m gpu 2>&1 | tail -15
timeout -k 10 600 python tools/host_stage_bench.py > gpurun_out/host_stage.json 2> gpurun_out/host_stage.err
cat gpurun_out/host_stage.err | tail -12; cat gpurun_out/host_stage.json
timeout -k 10 900 python tools/microbench.py packed > gpurun_out/mb_packed.json 2> gpurun_out/mb_packed.err
tail -3 gpurun_out/mb_packed.err; cat gpurun_out/mb_packed.json
