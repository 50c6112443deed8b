#!/bin/bash
# round 3, call e: the device UMI stage (parity with the host path, the oracle and the reference-executed groups), the new K-BC2 and
# large-group tests, then the assignumis microbench
set -u
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_umi_stage_gpu.py tests/test_umi_gpu.py tests/test_pipeline_gpu.py tests/test_config4_gpu.py -x -q -m gpu 2>&1 | tail -15 || exit 1
timeout -k 10 900 python -m pytest tests/test_bc_gpu.py -x -q -m gpu -k "long_lists" 2>&1 | tail -5
SMI_AU_TIMING=1 timeout -k 10 600 python tools/microbench.py assignumis > gpurun_out/mb_assignumis.json 2> gpurun_out/mb_assignumis.err
tail -12 gpurun_out/mb_assignumis.err; cat gpurun_out/mb_assignumis.json
