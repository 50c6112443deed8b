#!/bin/bash
# round 3, final build: seed sweep over the round's paths (packed boundary, device UMI stage with the device-sorted region grouping, K-DEFLATE, K-INFLATE)
set -u
mkdir -p gpurun_out/fuzz
SMI_FUZZ_LEGS=r3 timeout -k 10 700 python tools/fuzz_parity.py 10 3000000 > gpurun_out/fuzz/r03_final.log 2>&1; echo "r03 final rc=$?"; tail -2 gpurun_out/fuzz/r03_final.log
