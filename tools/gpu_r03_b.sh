#!/bin/bash
# round 3, call b: the packed boundary -- parity of the packed chunk workers with the text workers / the reference's records, then the
# host-to-host rate by lanes x host threads (tools/microbench.py packed)
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_packed_gpu.py tests/test_ref_exec_gpu.py -x -q -m gpu 2>&1 | tail -15 || exit 1
timeout -k 10 900 python tools/microbench.py packed > gpurun_out/mb_packed.json 2> gpurun_out/mb_packed.err
tail -3 gpurun_out/mb_packed.err; cat gpurun_out/mb_packed.json
