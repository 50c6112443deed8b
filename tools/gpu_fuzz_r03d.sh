#!/bin/bash
# round 3, last build: every leg (incl. the host gzip decoder and the native BAM writer) on a sixth seed range
set -u
mkdir -p gpurun_out/fuzz
timeout -k 10 800 python tools/fuzz_parity.py 11 5000000 > gpurun_out/fuzz/r03_last_all.log 2>&1; echo "all legs rc=$?"; tail -1 gpurun_out/fuzz/r03_last_all.log
