#!/bin/bash
# K-SCAN at 4 / 3 / 2 waves per SIMD (register cap 128 / 168 / 256): the bench step's kernel time per build (rebuilt on the box)
set -u
mkdir -p gpurun_out
for w in 3 2 4; do
  touch sicelore-2.1_amd/csrc/smi_scan.hip && make -s -j16 -C sicelore-2.1_amd/csrc EXTRA=-DSMI_SCAN_WAVES=$w || exit 1
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --two-pass-reads 0 --e2e-reads 0 --umi-molecules 0 --h2h-reads 0 --f2f-reads 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves $w', d['roofline']['kernels_ms'], d['ms_per_step'])" | tee -a gpurun_out/scan_waves.txt
done
