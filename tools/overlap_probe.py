"""Do a VALU-bound kernel and a memory-bound one share the GPU when they come from two streams?  (run on the GPU box)

The end-to-end chunk of scanfastq is a chain of kernels of two kinds: integer-issue bound (the splitter's filter, K-SCAN) and memory bound (K-FQ's
sweep, K-PACKR, K-WRITE).  A job runs several chunks at once on worker lanes (a stream each), so a filter of one chunk can meet a writer of another.
This probe measures what that meeting is worth: the splitter (K-CHIM, 1 M reads) and K-SCAN (10 M read ends' worth of planes) alone, a device copy of the
bytes a chunk's memory-bound kernels move alone, then both at once from two streams, in both launch orders -- the time of the pair against the sum and
against the longer of the two.  One JSON line."""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    pkg = graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = pkg.Context(0)
    wl = synth.make_whitelist(200_000, seed=1, device=dev)
    used = synth.pick_used(wl, 3000, seed=2)
    ctx.set_barcode_set_device(wl.to(torch.int32), mode=1)
    # ---- the splitter's input
    n = 1_000_000
    rd = synth.gen_reads(n, used, seed=7, device=dev)
    buf, offs = synth.materialize_device(rd)
    keep = torch.ones(n + 1, dtype=torch.bool, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    keep[1:n][torch.rand(n - 1, device=dev, generator=g) < 0.1] = False
    offs = offs[keep].contiguous()
    n = offs.numel() - 1
    total = int(offs[-1])
    planes = torch.zeros(ctx.read_planes_words(total, n), dtype=torch.int32, device=dev)
    cres = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    ccfg = ctx.chimera_config()
    ctx.pack_reads_device(buf, offs, n, total, planes)
    # ---- K-SCAN's input
    ns = 2_000_000
    rs = synth.gen_reads(ns, used, seed=9, device=dev)
    ends = synth.pack_ends(rs["head"], rs["tail"])
    lens = (2 * synth.END_BASES + rs["mid_len"]).to(torch.int32)
    so = torch.zeros((ns, 8), dtype=torch.int32, device=dev)
    win = torch.zeros((ns, 2), dtype=torch.int64, device=dev)
    scfg = ctx.scan_config(2)
    # ---- the memory-bound side: a copy of 1.25 GB (read + write = 2.5 GB, what K-WRITE of a 0.45 M-read chunk moves), several times
    src = torch.empty(1_250_000_000, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def valu_chim(reps):
        for _ in range(reps):
            ctx.chimera_device(planes, offs, n, total, ccfg, cres, stream=sa)

    def valu_scan(reps):
        for _ in range(reps):
            ctx.scan_device(ends, lens, ns, scfg, so, win, stream=sa)

    def mem(reps):
        with torch.cuda.stream(sb):
            for _ in range(reps):
                dst.copy_(src, non_blocking=True)

    def timed(*fns):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in fns:
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    res = {}
    for name, valu, vr in (("splitter", valu_chim, 4), ("k_scan", valu_scan, 12)):
        valu(1)
        mem(1)
        a = min(timed(lambda: valu(vr)) for _ in range(3))
        # as many copies as fill about the same time
        one = min(timed(lambda: mem(4)) for _ in range(3)) / 4
        mr = max(1, int(round(a / one)))
        b = min(timed(lambda: mem(mr)) for _ in range(3))
        c1 = min(timed(lambda: valu(vr), lambda: mem(mr)) for _ in range(3))
        c2 = min(timed(lambda: mem(mr), lambda: valu(vr)) for _ in range(3))
        res[name] = {"valu_ms": a, "copy_ms": b, "copies": mr, "copy_TBps": 2 * src.numel() * mr / b / 1e9, "both_valu_first_ms": c1, "both_copy_first_ms": c2,
                     "sum_ms": a + b, "longer_ms": max(a, b), "overlap_gain": (a + b) / min(c1, c2)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
