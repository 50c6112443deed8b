#!/bin/bash
# final build of round 2: two more seed ranges (shipped build; everything switched off)
set -u
mkdir -p gpurun_out/fuzz
timeout 800 python tools/fuzz_parity.py 11 1200000 > gpurun_out/fuzz/range5.log 2>&1; echo "range5 rc=$?"; tail -1 gpurun_out/fuzz/range5.log
SMI_SCAN_GENERIC=1 SMI_CHIM_NO_PREFILTER=1 SMI_BC1_NO_FILTER=1 SMI_BC2_NO_FILTER=1 SMI_BC2_NO_OFFSET_FILTER=1 timeout 400 python tools/fuzz_parity.py 5 1500000 > gpurun_out/fuzz/range6_plain.log 2>&1; echo "range6 (no filters) rc=$?"; tail -1 gpurun_out/fuzz/range6_plain.log
