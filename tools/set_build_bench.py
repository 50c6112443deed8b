#!/usr/bin/env python3
"""The one-time cost behind K-BC1: smi_set_barcode_set of the whole 3.6 M list (pyramid + nb + nb5 + nt), timed per call (smi_set_stats) -- run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (profiles/r06/set_build_kernel_stats.csv).  usage: set_build_bench.py [n_keys] [reps]"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_600_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    pkg = graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    dev = torch.device("cuda", 0)
    wl = synth.make_whitelist(n, seed=1, device=dev)
    ctx = pkg.Context(0)
    out = []
    for r in range(reps):
        ctx.set_barcode_set_device(wl.to(torch.int32), mode=1)
        out.append(ctx.set_stats(digests=(r == reps - 1)))
    print(json.dumps({"keys": n, "build_ms": [round(o["build_ms"], 2) for o in out], "hbm_bytes": out[-1]["hbm_bytes"], "last": out[-1],
                      "switches": {k: v for k, v in os.environ.items() if k.startswith("SMI_")}}))


if __name__ == "__main__":
    main()
