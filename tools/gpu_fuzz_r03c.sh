#!/bin/bash
# round 3, final build: every leg (round 2's bc / records, round 3's packed / umi_stage / deflate / inflate) on a fourth seed range, then the kernels' older / generic
# generations (everything switched off) on a fifth
set -u
mkdir -p gpurun_out/fuzz
timeout -k 10 700 python tools/fuzz_parity.py 10 4000000 > gpurun_out/fuzz/r03_final_all.log 2>&1; echo "all legs rc=$?"; tail -1 gpurun_out/fuzz/r03_final_all.log
SMI_SCAN_GENERIC=1 SMI_CHIM_NO_PREFILTER=1 SMI_BC1_NO_FILTER=1 SMI_BC2_NO_FILTER=1 SMI_BC2_NO_OFFSET_FILTER=1 SMI_HOST_SIMD=0 timeout -k 10 400 python tools/fuzz_parity.py 5 4500000 > gpurun_out/fuzz/r03_final_plain.log 2>&1; echo "plain rc=$?"; tail -1 gpurun_out/fuzz/r03_final_plain.log
