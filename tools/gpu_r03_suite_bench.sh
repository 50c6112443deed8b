#!/bin/bash
# round 3, call h: the whole -m gpu suite, then the bench line as the driver runs it (+ the single-process two-pass leg)
set -u
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 || exit 1
( time timeout -k 10 900 python bench.py --single-process-gpus 1 ) > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err
tail -5 gpurun_out/bench_r03.err
python3 - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/bench_r03.json") if l.startswith("{")][-1])
for k in ("metric", "value", "ms_per_step", "value_full_pass2", "value_bc_umi", "value_host_to_host"):
    print(k, d.get(k))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "kernel_ms", "achieved", "frac", "basis", "traffic")})
print("other", {k: {kk: v[kk] for kk in ("kernel_ms", "achieved", "frac", "basis", "traffic")} for k, v in d["roofline"]["other"].items()})
print("umi_stage", {k: d["umi_stage"][k] for k in ("records_per_chunk", "records_per_s", "lanes_at_best", "runs")})
print("host_to_host", {k: d["host_to_host"][k] for k in ("reads_per_s", "ms_per_chunk", "lanes", "host_threads_per_lane")})
print("file_to_file", d.get("file_to_file"))
print("two_pass_single_process", d.get("two_pass_single_process"))
print("cpu_baseline", d.get("cpu_baseline"))
P
