#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_chimera_gpu.py tests/test_write_gpu.py tests/test_pipeline_gpu.py tests/test_ref_exec_gpu.py tests/test_fastq_gpu.py -m gpu -x -q > gpurun_out/gputests_p.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/gputests_p.log
bash tools/gpu_e2e_prof.sh
python - <<'PY'
import json,re
t=open("gpurun_out/prof_e2e.log").read()
m=re.search(r'"end_to_end": (\{.*?\})', t)
print(m.group(1)[:400] if m else "no e2e")
PY
