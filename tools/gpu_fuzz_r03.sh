#!/bin/bash
# round 3: seed sweep over the paths added this round (packed boundary, device UMI stage, K-DEFLATE) and, on another seed range, over all legs
set -u
mkdir -p gpurun_out/fuzz
SMI_FUZZ_LEGS=r3 timeout -k 10 700 python tools/fuzz_parity.py 10 2000000 > gpurun_out/fuzz/r03_new_paths.log 2>&1; echo "r03 new paths rc=$?"; tail -2 gpurun_out/fuzz/r03_new_paths.log
timeout -k 10 400 python tools/fuzz_parity.py 5 2500000 > gpurun_out/fuzz/r03_all.log 2>&1; echo "r03 all legs rc=$?"; tail -2 gpurun_out/fuzz/r03_all.log
