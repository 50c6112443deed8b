#!/bin/bash
set -u
timeout 1800 python -m pytest tests/test_write_gpu.py tests/test_pipeline_gpu.py tests/test_fastq_gpu.py tests/test_config4_gpu.py tests/test_ref_exec_gpu.py -m gpu -x -q 2>&1 | tail -3
bash tools/gpu_lanes.sh
