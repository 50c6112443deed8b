#!/usr/bin/env python3
"""Host <-> device link rate of the box, pinned memory, each direction alone and both at once: the bound of every host-buffer entry point
(smi_scanfastq_pass2_chunk moves ~2.4 KB in and ~2.5 KB out per read).  Prints one JSON line."""
import json
import time

import torch

dev = torch.device("cuda", 0)
n = 1 << 30
h_in = torch.empty(n, dtype=torch.uint8, pin_memory=True)
h_out = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d_a = torch.empty(n, dtype=torch.uint8, device=dev)
d_b = torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def h2d():
    with torch.cuda.stream(s1):
        d_a.copy_(h_in, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        h_out.copy_(d_b, non_blocking=True)


def both():
    h2d()
    d2h()


res = {"bytes": n, "h2d_GB_s": n / timed(h2d) / 1e9, "d2h_GB_s": n / timed(d2h) / 1e9}
t = timed(both)
res["both_GB_s_sum"] = 2 * n / t / 1e9
print(json.dumps(res))
