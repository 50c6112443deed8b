#!/bin/bash
# where K-SCAN's time goes: the bench step with parts of the kernel switched off (results are wrong by construction)
set -u
# the shipped library refuses SMI_SCAN_ABLATE: build the measurement variant on the box first (and never ship it)
touch sicelore-2.1_amd/csrc/smi_scan.hip sicelore-2.1_amd/csrc/smi_chimera.hip && make -s -j16 -C sicelore-2.1_amd/csrc MEASURE=1
for a in 0 16 1 2 4 8 3 15; do
  SMI_SCAN_ABLATE=$a timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --two-pass-reads 0 --e2e-reads 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate $a', round(d['roofline']['kernels_ms']['k_scan<10>'],3), 'ms')"
done
