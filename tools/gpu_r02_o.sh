#!/bin/bash
# SWAR packers + K-FQ with 64 bytes per thread: the whole -m gpu suite, then the end-to-end kernel trace
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_o.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/gputests_o.log
bash tools/gpu_e2e_prof.sh
