#!/bin/bash
set -u
timeout 1800 python -m pytest tests/test_bc_gpu.py tests/test_ref_exec_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --two-pass-reads 0 --e2e-reads 0 > gpurun_out/bench_i.json 2> gpurun_out/bench_i.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_i.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"])
PY
