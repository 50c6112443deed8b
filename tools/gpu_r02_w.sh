#!/bin/bash
# K-BC1 filter: time of loading the 3.6 M set, bench step
set -u
mkdir -p gpurun_out
python - <<'PY'
import time, importlib, torch
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module(g.PKG_NAME + ".synth")
dev = torch.device("cuda:0")
ctx = pkg.Context(0)
wl = synth.make_whitelist(3_600_000, seed=1, device=dev)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.set_barcode_set_device(wl.to(torch.int32), mode=1)
    torch.cuda.synchronize(); print("set_barcode_set_device 3.6M:", round((time.perf_counter() - t0) * 1e3, 1), "ms")
used = synth.pick_used(wl, 5000, seed=2)
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.set_barcode_set_device(used.to(dev).to(torch.int32), mode=0)
    torch.cuda.synchronize(); print("set_barcode_set_device 5k:", round((time.perf_counter() - t0) * 1e3, 1), "ms")
PY
timeout -k 10 600 python bench.py --two-pass-reads 0 --e2e-reads 0 --no-cpu-baseline > gpurun_out/bench_w.json 2> gpurun_out/bench_w.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_w.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"])
PY
