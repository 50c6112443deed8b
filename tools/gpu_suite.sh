#!/bin/bash
# the whole GPU suite, then the default bench line (no core files: a crashing torch process writes tens of GB)
set -u
TAG=${TAG:-r06}   # the round the outputs are named after (profiles/$TAG/ once copied there)
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu "$@" > gpurun_out/${TAG}_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/${TAG}_gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/${TAG}_bench.json
