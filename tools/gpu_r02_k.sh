#!/bin/bash
# banded Needleman-Wunsch: the parity suites that reach K-SCAN / K-CHIM, then the bench step
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_scan_gpu.py tests/test_ref_exec_gpu.py tests/test_chimera_gpu.py tests/test_write_gpu.py tests/test_pipeline_gpu.py tests/test_config4_gpu.py -m gpu -x -q > gpurun_out/gputests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputests_k.log
timeout -k 10 600 python bench.py --no-cpu-baseline --two-pass-reads 0 > gpurun_out/bench_k.json 2> gpurun_out/bench_k.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_k.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"], d.get("value_full_pass2"))
PY
