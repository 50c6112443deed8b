#!/bin/bash
# in-place ingest (no gathers), pass 2 without quality packing: parity of the affected suites, then the bench line
set -u
mkdir -p gpurun_out/r02f
timeout 1800 python -m pytest tests/test_write_gpu.py tests/test_fastq_gpu.py tests/test_pipeline_gpu.py tests/test_ref_exec_gpu.py tests/test_chimera_gpu.py tests/test_config4_gpu.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py > gpurun_out/r02f/bench_n1.json 2> gpurun_out/r02f/bench_n1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02f/bench_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"], d["value_full_pass2"], d["end_to_end"]["ms"], d["cpu_baseline"]["matches_gpu"])
print(d["two_pass"]["pass2_ms"], d["two_pass"]["pass1_ms"])
PY
tail -3 gpurun_out/r02f/bench_n1.err
timeout 600 python tools/microbench.py lanes 2>&1 | tail -3 | cut -c1-600
