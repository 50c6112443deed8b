#!/bin/bash
# round-2 first GPU call: the VALU issue ceiling, then the state of the tree (gpu tests + bench)
set -u
mkdir -p gpurun_out/r02a
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/valu_peak.hip -o /tmp/valu_peak && timeout 300 /tmp/valu_peak > gpurun_out/r02a/valu_peak.json 2> gpurun_out/r02a/valu_peak.err
tail -c 600 gpurun_out/r02a/valu_peak.json
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; tail -3 gpurun_out/r02a/pytest.log
timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; tail -c 1500 gpurun_out/r02a/bench.json
