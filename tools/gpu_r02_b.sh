#!/bin/bash
# bench (configs[1]) with the two-pass leg, then configs[2] at reduced and full size
set -u
mkdir -p gpurun_out/r02b
timeout 900 python bench.py > gpurun_out/r02b/bench_n1.json 2> gpurun_out/r02b/bench_n1.err; echo "rc=$?"; tail -c 2500 gpurun_out/r02b/bench_n1.json; tail -5 gpurun_out/r02b/bench_n1.err
timeout 900 python bench.py --config 2 --reads 20000000 --steps 2 --warmup 1 > gpurun_out/r02b/bench_cfg2_20m.json 2> gpurun_out/r02b/bench_cfg2_20m.err; echo "rc=$?"; tail -c 1500 gpurun_out/r02b/bench_cfg2_20m.json; tail -5 gpurun_out/r02b/bench_cfg2_20m.err
timeout 1500 python bench.py --config 2 --reads 100000000 --steps 2 --warmup 1 > gpurun_out/r02b/bench_cfg2_100m.json 2> gpurun_out/r02b/bench_cfg2_100m.err; echo "rc=$?"; tail -c 1500 gpurun_out/r02b/bench_cfg2_100m.json; tail -5 gpurun_out/r02b/bench_cfg2_100m.err
