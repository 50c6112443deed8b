"""The phases of one pass-2 chunk through the one-call worker (SMI_WK_TIMING: host clock per phase, each phase ends in one of the call's four
waits), 0.45 M reads with 10 % chimeras as in bench.py's end_to_end leg.  Usage on the GPU box: python tools/e2e_phases.py [n_reads]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

graft.build()
import torch  # noqa: E402

pkg = importlib.import_module("sicelore_amd")
synth = importlib.import_module("sicelore_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 450_000
dev = torch.device("cuda", 0)
wl = synth.make_whitelist(3_600_000, seed=1)
used = wl[:5000]
ctx = pkg.Context(0)
ctx.set_barcode_set(wl.numpy().astype("uint64"), mode=1)
rd = synth.gen_reads(n, used, seed=77, device=dev)
text, _buf, offs0 = synth.fastq_text_device(rd, chimera_frac=0.10)
for _ in range(3):
    ctx.scanfastq_pass2_chunk(text, max_ed=1, device_output=True, copy=False)
os.environ["SMI_WK_TIMING"] = "1"
for _ in range(5):
    ctx.scanfastq_pass2_chunk(text, max_ed=1, device_output=True, copy=False)
import time  # noqa: E402

del os.environ["SMI_WK_TIMING"]
for kw in (dict(device_output=True, copy=False), dict(device_output=True, copy=True)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.scanfastq_pass2_chunk(text, max_ed=1, **kw)
    torch.cuda.synchronize()
    print(kw, "ms per call from Python:", (time.perf_counter() - t0) * 100)
import threading  # noqa: E402


def rep(k):
    for _ in range(k):
        ctx.scanfastq_pass2_chunk(text, max_ed=1, device_output=True, copy=False)


for reps in (20, 20):
    th = threading.Thread(target=rep, args=(reps,))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th.start()
    th.join()
    torch.cuda.synchronize()
    print("in a fresh thread, ms per call:", (time.perf_counter() - t0) * 1e3 / reps)
os.environ["SMI_WK_TIMING"] = "1"
th = threading.Thread(target=rep, args=(3,))
th.start()
th.join()
