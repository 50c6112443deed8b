#!/bin/bash
set -u
timeout -k 10 800 python -m pytest tests/test_pipeline_gpu.py tests/test_config4_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 800 python tools/microbench.py assignumis 2> gpurun_out/micro_au.err | tail -1 > gpurun_out/micro_au.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/micro_au.json"))
for k, v in d.items():
    print(k, round(v["ms"], 1), "ms", round(v["records_per_s"] / 1e6, 2), "M rec/s", v["clustered"])
PY
