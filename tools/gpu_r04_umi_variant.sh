#!/bin/bash
# round 4: K-UMI variants side by side (SMI_LIBRARY): microbench time + write counters.  usage: gpu_r04_umi_variant.sh <lib suffix> ...
set -u
ulimit -c 0
mkdir -p gpurun_out
for v in "$@"; do
  export SMI_LIBRARY=$PWD/sicelore-2.1_amd/csrc/libsicelore_mi_$v.so
  timeout -k 10 300 python tools/microbench.py umi > gpurun_out/microbench_umi_$v.json 2> gpurun_out/microbench_umi_$v.err; echo "$v mb rc=$?"; cut -c1-300 gpurun_out/microbench_umi_$v.json
  PROFILE_PROG=$PWD/tools/microbench.py PMC_GROUPS="WRITE_SIZE" timeout -k 10 600 bash tools/profile_gpu.sh r04umi_$v umi 2>&1 | grep k_umi | cut -c1-400
done
