#!/bin/bash
# the integer VALU issue ceiling (tools/valu_peak.hip) -> gpurun_out/r02/valu_peak.json
set -u
mkdir -p gpurun_out/r02
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/valu_peak.hip -o /tmp/valu_peak && timeout 600 /tmp/valu_peak > gpurun_out/r02/valu_peak.json 2> gpurun_out/r02/valu_peak.err
tail -c 400 gpurun_out/r02/valu_peak.json
