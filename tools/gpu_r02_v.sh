#!/bin/bash
# K-BC1 behind the offset filter: the whole -m gpu suite, the matcher suites again without the filter, the bench step
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_v.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputests_v.log
SMI_BC1_NO_FILTER=1 timeout -k 10 900 python -m pytest tests/test_bc_gpu.py tests/test_ref_exec_gpu.py -m gpu -x -q > gpurun_out/gputests_v2.log 2>&1; echo "pytest(no filter) rc=$?"; tail -2 gpurun_out/gputests_v2.log
timeout -k 10 600 python bench.py --two-pass-reads 0 --e2e-reads 0 > gpurun_out/bench_v.json 2> gpurun_out/bench_v.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_v.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"], d["cpu_baseline"]["matches_gpu"], d["config"]["bc_assigned_frac"])
PY
