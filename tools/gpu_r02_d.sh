#!/bin/bash
# round 2, after the gene tagger and the K-CHIM split: full GPU suite, then the default bench line
set -u
mkdir -p gpurun_out/r02d
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 900 python bench.py > gpurun_out/r02d/bench_n1.json 2> gpurun_out/r02d/bench_n1.err
tail -c 3000 gpurun_out/r02d/bench_n1.json
tail -3 gpurun_out/r02d/bench_n1.err
