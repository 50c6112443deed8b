#!/bin/bash
# K-SCAN held to three blocks per CU by unused LDS (measurement build, SMI_SCAN_LDS_PAD) with K-BC1 of the previous step beside it
# (bench.py --overlap): does a free wave slot per SIMD let the two kernels share a CU?
set -u
mkdir -p gpurun_out
touch sicelore-2.1_amd/csrc/smi_scan.hip && make -s -j16 -C sicelore-2.1_amd/csrc MEASURE=1 || exit 1
for pad in 11000 12500; do
  for o in "" "--overlap"; do
    SMI_SCAN_LDS_PAD=$pad timeout -k 10 200 python bench.py $o --steps 60 --warmup 3 --no-cpu-baseline --two-pass-reads 0 --e2e-reads 0 --umi-molecules 0 --h2h-reads 0 --f2f-reads 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pad $pad $o', round(d['ms_per_step'],3), d['roofline']['kernels_ms'], d['config']['bc_assigned_total'])" | tee -a gpurun_out/scan_pad.txt || exit 1
  done
done
