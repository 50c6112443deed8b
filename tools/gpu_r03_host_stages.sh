#!/bin/bash
# round 3, call d: per-block cost of the host primitives on the box's CPU, the host stages, the packed microbench
set -u
mkdir -p gpurun_out
# (the SIMD probe ran in the first version of this call: profiles/r03/host_simd_probe.txt)
timeout -k 10 600 python tools/host_stage_bench.py > gpurun_out/host_stage.json 2> gpurun_out/host_stage.err
grep -E "^(1|8|16|32) " gpurun_out/host_stage.err
timeout -k 10 900 python tools/microbench.py packed > gpurun_out/mb_packed.json 2> gpurun_out/mb_packed.err
tail -3 gpurun_out/mb_packed.err
