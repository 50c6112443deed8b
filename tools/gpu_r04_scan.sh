#!/bin/bash
# round 4: K-SCAN with the bit-parallel polyT finder -- the scan's parity tests (shipped and generic kernels against the oracle, the executed
# fixtures), a seed sweep, then the step alone
set -u
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_scan_gpu.py tests/test_ref_exec_gpu.py tests/test_pipeline_gpu.py tests/test_fastq_gpu.py tests/test_config4_gpu.py -x -q -m gpu > gpurun_out/r04_scan_tests.log 2>&1; rc=$?; echo "rc=$rc"; tail -6 gpurun_out/r04_scan_tests.log | cut -c1-400
[ $rc -eq 0 ] || exit $rc
SMI_FUZZ_LEGS=r2 timeout -k 10 600 python tools/fuzz_parity.py 1.5 4000 > gpurun_out/r04_scan_fuzz.log 2>&1; rc=$?; echo "fuzz rc=$rc"; tail -3 gpurun_out/r04_scan_fuzz.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --two-pass-reads 0 --umi-molecules 0 --h2h-reads 0 --f2f-reads 0 --assignumis-file-records 0 > gpurun_out/r04_scan_bench.json 2> gpurun_out/r04_scan_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_scan_bench.json").read().strip().splitlines()[-1])
print(json.dumps({"value": d["value"], "ms_per_step": d["ms_per_step"], "k_scan_ms": d["roofline"]["kernel_ms"], "e2e_ms": d["end_to_end"]["ms"], "cpu_matches": d["cpu_baseline"]["matches_gpu"]}))
PY
