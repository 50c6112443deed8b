#!/bin/bash
set -u
for l in 1 2 3 4 6; do SMI_MB_LANES=$l timeout 900 python tools/microbench.py fastq 2>/dev/null | python -c "
import sys,json; d=json.load(sys.stdin); a=d['pass2_chunk_host_to_host_lanes']; b=d['pass2_chunk_host_to_host']
print('lanes', a['lanes'], round(a['reads_per_s']/1e6,2), 'M reads/s', round(a['GB_per_s_both_directions'],1), 'GB/s; single call', round(b['reads_per_s']/1e6,2), 'M reads/s', round(b['ms'],1), 'ms')"; done
