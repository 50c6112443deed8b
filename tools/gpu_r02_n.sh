#!/bin/bash
# writer parity + the end-to-end kernel trace
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_write_gpu.py tests/test_fastq_gpu.py tests/test_pipeline_gpu.py tests/test_ref_exec_gpu.py -m gpu -x -q > gpurun_out/gputests_n.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/gputests_n.log
bash tools/gpu_e2e_prof.sh
