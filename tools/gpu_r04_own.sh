#!/bin/bash
# round 4: the device clusterer of large UMI groups -- the UMI tests, then one 8,000-read group device against host (stage laps on stderr)
set -u
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_umi_gpu.py tests/test_umi_stage_gpu.py -x -q -m gpu > gpurun_out/r04_own_tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r04_own_tests.log | cut -c1-400
SMI_AU_TIMING=1 timeout -k 10 300 python tools/own_cluster_bench.py 8000 > gpurun_out/own_cluster_8000.json 2> gpurun_out/own_cluster_8000.err; echo "rc=$?"; cat gpurun_out/own_cluster_8000.json
grep -i "big groups" gpurun_out/own_cluster_8000.err | head -14
python3 tools/own_cluster_profile.py gpurun_out/own_cluster_8000 gpurun_out/own_cluster_8000_profile.json
