#!/bin/bash
set -u
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_assignumis_shards_gpu.py tests/test_pipeline_gpu.py tests/test_gene_counts.py -x -q -m gpu > gpurun_out/r04_t2.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r04_t2.log | cut -c1-900
