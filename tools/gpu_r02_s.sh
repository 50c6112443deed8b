#!/bin/bash
# state after the pass-2 byte kernels and K-CHIM-C at 4 waves: whole -m gpu suite, default bench, round profile (kernel trace + PMC + chimera + e2e traces)
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_s.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputests_s.log
timeout -k 10 600 python bench.py > gpurun_out/bench_s.json 2> gpurun_out/bench_s.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_s.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["kernels_ms"], d.get("value_full_pass2"), d["end_to_end"]["ms"], d["cpu_baseline"]["matches_gpu"])
PY
bash tools/gpu_prof.sh > gpurun_out/prof_s.log 2>&1; tail -3 gpurun_out/prof_s.log | cut -c1-300
