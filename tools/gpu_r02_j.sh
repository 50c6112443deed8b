#!/bin/bash
# re-entry baseline of round 2: the whole -m gpu suite, then the default bench line
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_j.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputests_j.log
timeout -k 10 600 python bench.py > gpurun_out/bench_j.json 2> gpurun_out/bench_j.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/bench_j.json
