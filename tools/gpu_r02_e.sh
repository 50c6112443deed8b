#!/bin/bash
# K-SCAN with compile-time adapters and packed fold state: parity (both builds), then the bench line
set -u
mkdir -p gpurun_out/r02e
timeout 1200 python -m pytest tests/test_scan_gpu.py tests/test_ref_exec_gpu.py tests/test_pipeline_gpu.py tests/test_write_gpu.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py > gpurun_out/r02e/bench_n1.json 2> gpurun_out/r02e/bench_n1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02e/bench_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"], d["value_full_pass2"], d["cpu_baseline"]["matches_gpu"])
PY
SMI_SCAN_GENERIC=1 timeout 900 python bench.py --no-cpu-baseline --two-pass-reads 0 > gpurun_out/r02e/bench_generic.json 2> gpurun_out/r02e/bench_generic.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02e/bench_generic.json").read().strip().splitlines()[-1])
print("generic", d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"])
PY
