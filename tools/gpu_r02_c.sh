#!/bin/bash
set -u
mkdir -p gpurun_out/r02c
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/r02c/bench_n1.json 2> gpurun_out/r02c/bench_n1.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02c/bench_n1.json'))
print(d['value'], d['ms_per_step'], d.get('value_full_pass2'), {k:v for k,v in d['end_to_end'].items() if k!='stages'})
PY
