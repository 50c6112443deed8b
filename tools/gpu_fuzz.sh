#!/bin/bash
# seed sweep of whole records through the native chunk worker and of the matcher against the oracle: the shipped build, then the
# generic K-SCAN kernels and K-CHIM without its exact prefilter
set -u
mkdir -p gpurun_out/fuzz
timeout 900 python tools/fuzz_parity.py 6 > gpurun_out/fuzz/default.log 2>&1; echo "default rc=$?"; tail -2 gpurun_out/fuzz/default.log; grep -c "^ok" gpurun_out/fuzz/default.log
SMI_SCAN_GENERIC=1 SMI_CHIM_NO_PREFILTER=1 timeout 600 python tools/fuzz_parity.py 3 > gpurun_out/fuzz/generic.log 2>&1; echo "generic rc=$?"; tail -2 gpurun_out/fuzz/generic.log; grep -c "^ok" gpurun_out/fuzz/generic.log
