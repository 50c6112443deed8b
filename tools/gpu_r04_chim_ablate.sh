#!/bin/bash
# round 4: what K-CHIM-A queues, and the splitter's time with parts switched off (measurement build, wrong results by construction)
set -u
mkdir -p gpurun_out
export SMI_LIBRARY=$PWD/sicelore-2.1_amd/csrc/libsicelore_mi_measure.so
for a in 0 16 1 2 32 64 3; do
  echo "== SMI_CHIM_ABLATE=$a"
  SMI_CHIM_STATS=1 SMI_CHIM_ABLATE=$a timeout -k 10 300 python tools/microbench.py chimera 2> gpurun_out/chim_ablate_$a.err | grep -o '"chimera": {[^}]*}' | cut -c1-200
  grep "chim stats" gpurun_out/chim_ablate_$a.err | tail -1
done
