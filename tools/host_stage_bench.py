#!/usr/bin/env python3
"""Where the host side of the packed boundary spends its time on a GPU box: every host stage (index, planes, records) timed alone, with its
buffers in ordinary (malloc'ed, pre-touched) memory and in page-locked memory (smi_host_alloc = hipHostMalloc), on 1 .. 32 threads."""
import ctypes
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


def best(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    pkg = graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    libmod = importlib.import_module(graft.PKG_NAME + ".lib")
    L = libmod.load_library()
    dev = torch.device("cuda:0")
    n = int(os.environ.get("SMI_MB_READS", "500000"))
    wl = synth.make_whitelist(3_600_000, seed=1, device=dev)
    used = synth.pick_used(wl, 5000, seed=2)
    ctx = pkg.Context(0)
    ctx.set_barcode_set_device(used.to(torch.int32), mode=0)
    rd = synth.gen_reads(n, used, seed=9, device=dev)
    text_d = synth.fastq_text_device(rd)[0]
    total = int(text_d.numel())
    text_np = text_d.cpu().numpy().copy()
    pin_text = libmod.PinnedBuffer(total)
    pin_text.array[:] = text_np
    res = {"reads": n, "text_bytes": total, "threads": {}}
    recs, offs, err = libmod.fastq_index_host(text_np, n_threads=16)
    nrec = recs.size
    pw = libmod.read_planes_words(int(offs[-1]), nrec)
    planes_np = np.zeros(pw, dtype=np.uint32)
    pin_planes = libmod.PinnedBuffer(pw * 4)
    # decisions of this chunk from one device call (pointers stay valid: no further call on ctx)
    cfg = libmod.Pass2Config()
    L.smi_pass2_default_config(ctypes.byref(cfg))
    L.smi_pack_reads_host(text_np.ctypes.data, recs.ctypes.data, offs.ctypes.data, nrec, planes_np.ctypes.data, 16)
    dec = libmod.Pass2Decisions()
    rc = L.smi_scanfastq_pass2_packed(ctx._h, planes_np.ctypes.data, offs.ctypes.data, nrec, ctypes.byref(cfg), ctypes.byref(dec))
    assert rc == 0, L.smi_last_error()
    cap = 2 * total + 400 * dec.n_records_out
    out_np = [np.zeros(cap, dtype=np.uint8), np.zeros(cap, dtype=np.uint8)]
    out_pin = [libmod.PinnedBuffer(cap), libmod.PinnedBuffer(cap)]
    recs_buf = np.zeros(nrec + 8, dtype=libmod.FASTQ_RECORD_DTYPE)
    offs_buf = np.zeros(nrec + 9, dtype=np.uint64)
    wcfg = (ctypes.c_int32 * 2)(0, 0)
    totals = (ctypes.c_uint64 * 3)()
    werr = ctypes.c_uint32(0)
    nn, ee = ctypes.c_size_t(0), ctypes.c_uint32(0)
    pk = libmod.PackedReads()
    pstart_buf = np.zeros(nrec + 8, dtype=np.uint32)
    fw = int(L.smi_packed_planes_words(total, 32))
    fused_np = np.zeros(fw, dtype=np.uint32)
    for nt in [int(x) for x in os.environ.get("SMI_HSB_THREADS", "1,4,8,12,16,24,32").split(",")]:
        r = {}
        r["index_pack_one_pass_ms"] = best(lambda: L.smi_fastq_index_pack_host(text_np.ctypes.data, total, recs_buf.ctypes.data, offs_buf.ctypes.data,
                                                                                pstart_buf.ctypes.data, nrec + 4, fused_np.ctypes.data, fw, ctypes.byref(pk),
                                                                                ctypes.byref(nn), ctypes.byref(ee), nt)) * 1e3
        assert pk.n_seg == nt or nt == 1
        for label, tx in (("malloc", text_np), ("pinned", pin_text.array)):
            r[f"index_text_{label}_ms"] = best(lambda: L.smi_fastq_index_host(tx.ctypes.data, total, recs_buf.ctypes.data, offs_buf.ctypes.data, nrec + 4,
                                                                              ctypes.byref(nn), ctypes.byref(ee), nt)) * 1e3
        for label, pl in (("malloc", planes_np.ctypes.data), ("pinned", pin_planes.array.ctypes.data)):
            r[f"pack_planes_{label}_ms"] = best(lambda: L.smi_pack_reads_host(text_np.ctypes.data, recs.ctypes.data, offs.ctypes.data, nrec, pl, nt)) * 1e3
        for label, outs in (("malloc", [o.ctypes.data for o in out_np]), ("pinned", [o.array.ctypes.data for o in out_pin])):
            def wr():
                rc = L.smi_fastq_write_host(text_np.ctypes.data, recs.ctypes.data, offs.ctypes.data, ctypes.byref(dec), 1, ctypes.byref(wcfg), outs[0], cap,
                                            outs[1], cap, totals, ctypes.byref(werr), nt)
                assert rc == 0
            r[f"write_out_{label}_ms"] = best(wr) * 1e3
        res["threads"][str(nt)] = r
        print(nt, json.dumps(r), file=sys.stderr, flush=True)
    res["out_bytes"] = int(totals[0] + totals[1])
    # plain copies, 16 numpy threads: malloc -> malloc, malloc -> pinned
    import threading

    def copy_bw(dst, src, k=16):
        m = src.size // k
        def work(i):
            np.copyto(dst[i * m:(i + 1) * m], src[i * m:(i + 1) * m])
        def run():
            th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        return src.size / best(run) / 1e9
    res["copy16_malloc_to_malloc_GBps"] = copy_bw(out_np[0][:total], text_np)
    res["copy16_malloc_to_pinned_GBps"] = copy_bw(out_pin[0].array[:total], text_np)
    res["copy16_pinned_to_malloc_GBps"] = copy_bw(out_np[0][:total], pin_text.array)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
