#!/bin/bash
# a second seed range for the sweep of tools/gpu_fuzz.sh (the default range always starts at seed 1000): shipped build, then the generic
# kernels with every filter and table off
set -u
mkdir -p gpurun_out/fuzz
timeout 1000 python tools/fuzz_parity.py 9 500000 > gpurun_out/fuzz/range2.log 2>&1; echo "range2 rc=$?"; tail -1 gpurun_out/fuzz/range2.log
SMI_SCAN_GENERIC=1 SMI_CHIM_NO_PREFILTER=1 SMI_BC1_NO_FILTER=1 SMI_BC2_NO_FILTER=1 SMI_BC2_NO_OFFSET_FILTER=1 timeout 500 python tools/fuzz_parity.py 4 700000 > gpurun_out/fuzz/range3_plain.log 2>&1; echo "range3 (no filters) rc=$?"; tail -1 gpurun_out/fuzz/range3_plain.log
