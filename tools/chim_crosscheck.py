#!/usr/bin/env python3
"""K-CHIM generations against each other on one large synthetic batch (0.9 M reads, 10 % joined): the records of the default
pipeline must equal those of the first-generation kernels (SMI_CHIM_V1: every gated position aligned by one wave per read) and of
the filter-less path (SMI_CHIM_NO_PREFILTER, on a slice).  Prints one JSON object; exit code 1 on any difference."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    five = len(sys.argv) > 2 and sys.argv[2] == "5p"
    pkg = graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    dev = torch.device("cuda:0")
    ctx = pkg.Context(0)
    wl = synth.make_whitelist(200_000, seed=1, device=dev)
    used = synth.pick_used(wl, 5000, seed=2)
    rd = (synth.gen_reads_5p if five else synth.gen_reads)(n, used, seed=7, device=dev)
    buf, offs = synth.materialize_device(rd)
    keep = torch.ones(n + 1, dtype=torch.bool, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    keep[1:n][torch.rand(n - 1, device=dev, generator=g) < 0.1] = False
    offs = offs[keep].contiguous()
    n = offs.numel() - 1
    total = int(offs[-1])
    planes = torch.zeros(ctx.read_planes_words(total, n), dtype=torch.int32, device=dev)
    ctx.pack_reads_device(buf, offs, n, total, planes)
    cfg = ctx.chimera_config(five)
    res = {}

    def run(env, n_sub=None):
        for k in ("SMI_CHIM_V1", "SMI_CHIM_A1", "SMI_CHIM_NO_PREFILTER", "SMI_CHIM_GENERIC"):
            os.environ.pop(k, None)
        for k in env:
            os.environ[k] = "1"
        m = n if n_sub is None else n_sub
        out = torch.zeros((m, 4), dtype=torch.int32, device=dev)
        tot = int(offs[m])
        for _ in range(4):   # (the first launches after an idle spell run at a lower clock)
            ctx.chimera_device(planes, offs[:m + 1].contiguous(), m, tot, cfg, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.chimera_device(planes, offs[:m + 1].contiguous(), m, tot, cfg, out)
        torch.cuda.synchronize()
        return out.cpu().numpy(), (time.perf_counter() - t0) / 3 * 1e3

    a, ta = run([])
    b, tb = run(["SMI_CHIM_V1"])
    res["reads"] = n
    res["default_ms"], res["v1_ms"] = ta, tb
    res["default_equals_v1"] = bool((a == b).all())
    c, tc = run(["SMI_CHIM_NO_PREFILTER", "SMI_CHIM_V1"])
    res["no_prefilter_v1_ms"] = tc
    res["default_equals_no_prefilter"] = bool((a == c).all())
    c2, tc2 = run(["SMI_CHIM_NO_PREFILTER"])
    res["no_prefilter_ms"] = tc2
    res["default_equals_no_prefilter_v2"] = bool((a == c2).all())
    a1, ta1 = run(["SMI_CHIM_A1"])
    res["a1_ms"] = ta1
    res["default_equals_a1"] = bool((a == a1).all())
    d, td = run(["SMI_CHIM_GENERIC"])
    res["generic_ms"] = td
    res["default_equals_generic"] = bool((a == d).all())
    cr = a.view(pkg.CHIMERA_RESULT_DTYPE).reshape(-1)
    res["split_frac"] = float((cr["n_split"] > 0).mean())
    res["multi"] = int((cr["flags"] & 1).sum())
    if not res["default_equals_v1"]:
        bad = np.nonzero((a != b).any(axis=1))[0]
        res["first_diff"] = [int(x) for x in bad[:10]]
        res["diff_rows"] = {int(i): [a[i].tolist(), b[i].tolist()] for i in bad[:5]}
    print(json.dumps(res))
    ok = res["default_equals_v1"] and res["default_equals_no_prefilter"] and res["default_equals_no_prefilter_v2"] and res["default_equals_generic"] and res["default_equals_a1"]
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
