#!/bin/bash
# writer rewrite (plan rows, 16-byte copies, window loads): parity suites, fastq microbench, bench with the end-to-end leg
set -u
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 tools/perm_check.hip -o /tmp/perm_check && /tmp/perm_check
timeout -k 10 900 python -m pytest tests/test_write_gpu.py tests/test_fastq_gpu.py tests/test_pipeline_gpu.py tests/test_ref_exec_gpu.py tests/test_config4_gpu.py -m gpu -x -q > gpurun_out/gputests_m.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/gputests_m.log
timeout -k 10 600 python tools/microbench.py fastq > gpurun_out/micro_fastq_m.log 2>&1; echo "micro rc=$?"; tail -3 gpurun_out/micro_fastq_m.log | cut -c1-1500
timeout -k 10 600 python bench.py --no-cpu-baseline --two-pass-reads 0 > gpurun_out/bench_m.json 2> gpurun_out/bench_m.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_m.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"], d.get("value_full_pass2"), d["end_to_end"]["ms"])
PY
