#!/bin/bash
set -u
mkdir -p gpurun_out/r02g
timeout 1800 python -m pytest tests/test_write_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --two-pass-reads 0 --steps 3 --warmup 1 > gpurun_out/r02g/bench.json 2> gpurun_out/r02g/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02g/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["value_full_pass2"], d["end_to_end"]["ms"])
PY
