#!/bin/bash
# microbenchmarks after the pass-2 byte-kernel work: splitter, fastq ingest / writer / host-to-host chunk, worker lanes
set -u
mkdir -p gpurun_out
timeout -k 10 600 python tools/microbench.py chimera 2>/dev/null | tail -1 > gpurun_out/micro_chim_t.json; cut -c1-700 gpurun_out/micro_chim_t.json
timeout -k 10 600 python tools/microbench.py fastq 2>/dev/null | tail -1 > gpurun_out/micro_fastq_t.json; cut -c1-1500 gpurun_out/micro_fastq_t.json
bash tools/gpu_lanes.sh
