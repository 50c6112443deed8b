#!/bin/bash
# round 4: the one-sweep line index -- its tests, the workers that sit on it, then the bench's end-to-end leg
set -u
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_fastq_gpu.py tests/test_pipeline_gpu.py tests/test_write_gpu.py tests/test_packed_gpu.py -x -q -m gpu > gpurun_out/r04_fq_tests.log 2>&1; rc=$?; echo "rc=$rc"; tail -6 gpurun_out/r04_fq_tests.log | cut -c1-400
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --two-pass-reads 0 --umi-molecules 0 --h2h-reads 0 --f2f-reads 0 --assignumis-file-records 0 > gpurun_out/r04_fq_bench.json 2> gpurun_out/r04_fq_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_fq_bench.json").read().strip().splitlines()[-1])
print(json.dumps({k: d["end_to_end"][k] for k in ("reads", "ms", "reads_per_s")}))
PY
