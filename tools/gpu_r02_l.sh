#!/bin/bash
# banded alignments in K-CHIM: parity suites, splitter microbench, bench with the end-to-end leg
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ref_exec_gpu.py tests/test_chimera_gpu.py tests/test_write_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > gpurun_out/gputests_l.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputests_l.log
timeout -k 10 600 python tools/microbench.py chimera > gpurun_out/micro_chim_l.log 2>&1; echo "micro rc=$?"; tail -5 gpurun_out/micro_chim_l.log | cut -c1-600
timeout -k 10 600 python bench.py --no-cpu-baseline --two-pass-reads 0 > gpurun_out/bench_l.json 2> gpurun_out/bench_l.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_l.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"], d.get("value_full_pass2"), d["end_to_end"]["ms"])
PY
