#!/usr/bin/env python3
"""K-SCAN's shipped kernels (bit-parallel polyT finder, banded alignments, TSO pre-filter) against its generic kernels (the finder as a loop, full
matrices, every candidate aligned: SMI_SCAN_GENERIC) on one large synthetic batch, records and barcode windows byte for byte -- 3' pass 2 (10-mer),
3' pass 1 (22-mer, with qualities), 5' with and without the polyA search.  usage: scan_crosscheck.py [reads]; prints one JSON object, exit code 1
on any difference."""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    pkg = graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    dev = torch.device("cuda:0")
    ctx = pkg.Context(0)
    wl = synth.make_whitelist(200_000, seed=1, device=dev)
    used = synth.pick_used(wl, 5000, seed=2)
    res, ok = {"reads": n}, True
    for name, five, pass_no, dont_polya in (("3p_pass2", False, 2, False), ("3p_pass1", False, 1, False), ("5p_polya", True, 2, False), ("5p_nopolya", True, 2, True)):
        rd = (synth.gen_reads_5p if five else synth.gen_reads)(n, used, seed=17 + pass_no, device=dev, n_rate=0.002)
        buf, offs = synth.materialize_device(rd)
        quals = torch.randint(33 + 2, 33 + 40, (int(offs[-1]),), dtype=torch.uint8, device=dev)
        ends = torch.zeros((28, 2 * n), dtype=torch.int32, device=dev)
        lens = torch.zeros(n, dtype=torch.int32, device=dev)
        qtail = torch.zeros((n, 224), dtype=torch.uint8, device=dev)
        qsum = torch.zeros(n, dtype=torch.int32, device=dev)
        ctx.pack_ends_device(buf, quals, offs, n, ends, lens, qtail, qsum, five_prime=five)
        cfg = ctx.scan_config_5p(pass_no, dont_polya) if five else ctx.scan_config(pass_no)
        outs = []
        for generic in (False, True):
            if generic:
                os.environ["SMI_SCAN_GENERIC"] = "1"
            else:
                os.environ.pop("SMI_SCAN_GENERIC", None)
            out = torch.zeros((n, 8), dtype=torch.int32, device=dev)
            win = torch.zeros((n, 2), dtype=torch.int64, device=dev)
            ctx.scan_device(ends, lens, n, cfg, out, win, qtail if pass_no == 1 else None, qsum if pass_no == 1 else None)
            torch.cuda.synchronize()
            outs.append((out, win))
        os.environ.pop("SMI_SCAN_GENERIC", None)
        same = bool(torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]))
        found = int(((outs[0][0][:, 6] >> 16) & 0xFF).eq(1).sum())
        res[name] = {"shipped_equals_generic": same, "adapter_found": found}
        ok = ok and same
        del rd, buf, quals, ends, qtail
    print(json.dumps(res))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
