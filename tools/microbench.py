#!/usr/bin/env python3
"""Secondary kernels' throughput on one MI355X (not the headline bench): K-BC2 (ed<=2, used-list mode),
K-SCAN pass 1 (complete adapter + quality filter) + histogram, K-UMI.  Prints one JSON object."""
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


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None  # e.g. "chimera": just that leg
    pkg = graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    dev = torch.device("cuda:0")
    ctx = pkg.Context(0)
    res = {}
    wl = synth.make_whitelist(3_600_000, seed=1, device=dev)
    used = synth.pick_used(wl, 5000, seed=2)
    legs = {"bc2": leg_bc2, "bc": leg_bc, "pass1": leg_pass1, "umi": leg_umi, "chimera": leg_chimera, "fastq": leg_fastq, "assignumis": leg_assignumis,
            "packed": leg_packed, "deflate": leg_deflate, "inflate": leg_inflate}
    for name, fn in legs.items():
        if only == name or (only is None and name != "bc2"):
            fn(pkg, synth, ctx, dev, wl, used, res)
    print(json.dumps(res))


def leg_bc(pkg, synth, ctx, dev, wl, used, res):
    # ---- K-BC2 / K-BC1 in used-list mode -------------------------------------------------------------------
    n = 2_000_000
    reg = synth.gen_bc_region(n, used, seed=3, device=dev)
    win = synth.pack_windows(reg["codes"], reg["ae"])
    out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    ctx.set_barcode_set_device(used.to(torch.int32), mode=0)
    ctx.set_barcode_set_device(used.to(torch.int32), mode=0)          # (the second load: into structures that are allocated)
    res["used_list_5k_set"] = ctx.set_stats()
    for ed in (1, 2):
        dt = timed(lambda: ctx.bc_match_device(win, out, n, max_ed=ed))
        found = ((out[:, 2] & 0xFF) == 1)
        acc = float(((out[:, 0].to(torch.int64) & 0xFFFFFFFF)[found] == reg["truth"][found]).float().mean())
        res[f"bc_match_ed{ed}_used_list_5k"] = {"reads": n, "ms": dt * 1e3, "reads_per_s": n / dt,
                                                "assigned_frac": float(found.float().mean()), "accuracy": acc}
    ctx.set_barcode_set_device(wl.to(torch.int32), mode=1)
    dt = timed(lambda: ctx.bc_match_device(win, out, n, max_ed=2))
    res["bc_match_ed2_whitelist_3p6M"] = {"reads": n, "ms": dt * 1e3, "reads_per_s": n / dt}


def leg_bc2(pkg, synth, ctx, dev, wl, used, res):
    """K-BC2 alone against the 5 k used list (configs[2]'s dominant kernel): for counters (PROFILE_PROG=tools/microbench.py tools/profile_gpu.sh <tag> bc2)"""
    n = 2_000_000
    reg = synth.gen_bc_region(n, used, seed=3, device=dev)
    win = synth.pack_windows(reg["codes"], reg["ae"])
    out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    ctx.set_barcode_set_device(used.to(torch.int32), mode=0)
    dt = timed(lambda: ctx.bc_match_device(win, out, n, max_ed=2))
    res["bc_match_ed2_used_list_5k"] = {"reads": n, "ms": dt * 1e3, "reads_per_s": n / dt, "set": ctx.set_stats()}


def leg_pass1(pkg, synth, ctx, dev, wl, used, res):
    # ---- pass 1: scan (22-mer) + quality filter + histogram ---------------------------------------------------
    ctx.set_barcode_set_device(wl.to(torch.int32), mode=1)
    n = 2_000_000
    rd = synth.gen_reads(n, used, seed=5, device=dev, q_mean=14.0)
    ends = synth.pack_ends(rd["head"], rd["tail"])
    lens = (2 * synth.END_BASES + rd["mid_len"]).to(torch.int32)
    qtail = rd["qtail"].contiguous()
    qsum = (rd["qhead"].to(torch.int32).sum(1) + rd["qtail"].to(torch.int32).sum(1) - 33 * 2 * synth.END_BASES +
            (rd["qmid"].to(torch.int32) - 33) * rd["mid_len"].to(torch.int32)).to(torch.int32)
    cfg1 = ctx.scan_config(1)
    so = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    win = torch.zeros((n, 2), dtype=torch.int64, device=dev)
    hist = torch.zeros(wl.numel(), dtype=torch.int32, device=dev)

    def pass1():
        ctx.scan_device(ends, lens, n, cfg1, so, win, qtail, qsum)
        ctx.hist_windows_device(win, so, n, hist)

    dt = timed(pass1)
    res["pass1_scan22_filter_hist"] = {"reads": n, "ms": dt * 1e3, "reads_per_s": n / dt,
                                       "pass1_ok_frac": float((((so[:, 7]) & 0xFF) == 1).float().mean())}


def leg_umi(pkg, synth, ctx, dev, wl, used, res):
    # ---- K-UMI ------------------------------------------------------------------------------------------------
    rng = np.random.default_rng(1)
    sizes = np.minimum(rng.zipf(1.6, 400_000), 400).astype(np.int64) + 1
    # SMI_UMI_DENSE=1: the dense n x n layout of smi_umi_dist_device; default: the padded rows the chunk worker gives its own matrices (round 6)
    padded = not os.environ.get("SMI_UMI_DENSE")
    go, po, mo = ctx.umi_offsets(sizes, padded=padded)
    n_reads = int(go[-1])
    w = torch.randint(0, 4, (n_reads, 14), device=dev)
    codes = torch.tensor([1, 2, 4, 8], device=dev)[w]
    packed = (codes << (4 * torch.arange(14, device=dev))).sum(1)
    d_go = torch.from_numpy(go.view(np.int32)).to(dev)
    d_po = torch.from_numpy(po.view(np.int64)).to(dev)
    d_mo = torch.from_numpy(mo.view(np.int64)).to(dev)
    d_out = torch.zeros(int(mo[-1]), dtype=torch.uint8, device=dev)
    dt = timed(lambda: ctx.umi_dist_device(packed, d_go, d_po, d_mo, len(sizes), int(po[-1]), d_out, padded=padded))
    res["umi_dist"] = {"groups": int(len(sizes)), "reads": n_reads, "pairs": int(po[-1]), "ms": dt * 1e3, "layout": "padded rows" if padded else "dense",
                       "matrix_bytes": int((sizes.astype(np.int64) ** 2).sum()), "buffer_bytes": int(mo[-1]),
                       "pairs_per_s": int(po[-1]) / dt, "levenshtein_per_s": 9 * int(po[-1]) / dt}


def leg_assignumis(pkg, synth, ctx, dev, wl, used, res):
    """the second worker end to end: smi_assignumis_chunk (names parsed as getScanDatFromReadName does, clustering positions, region
    grouping, K-UMI, UMI clustering) on one BamReader chunk.  Names come out of a real pass 2 over synthetic molecules; every molecule is
    then read `copies` times (same barcode and UMI window, an error now and then), aligned to one of `genes` loci."""
    import ctypes
    scanfastq = importlib.import_module(graft.PKG_NAME + ".scanfastq")
    lib = importlib.import_module(graft.PKG_NAME + ".lib")
    rng = np.random.default_rng(5)
    n_mol, copies, genes = 20_000, 6, 2_000
    ctx.set_barcode_set(used.cpu().numpy().astype(np.uint64), mode=0)
    mol = synth.gen_reads(n_mol, used, seed=77, err=0.0, q_mean=20.0)
    seqs, quals = zip(*(synth.materialize(mol, i) for i in range(n_mol)))
    text = "".join(f"@m{i} ch=1\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(zip(seqs, quals))).encode()
    rs = scanfastq.ReadScanner(ctx, max_ed=1, split_chimeras=False)
    recs = [r for r in rs.pass2_chunk(text) if "_FAILED" not in r["name"] and "bc=" in r["name"]]
    gene = rng.integers(0, genes, len(recs))
    rows = []
    for m, r in enumerate(recs):
        q = r["name"].split(" ")[0]
        head, x_rest = q.split("_X=")
        x, rest = x_rest.split("_", 1)
        for c in range(copies):
            xs = list(x)
            if rng.random() < 0.3:  # a sequencing error inside the window
                xs[int(rng.integers(0, len(xs)))] = "ACGT"[int(rng.integers(0, 4))]
            nm = f"{head.replace('m', 'r%d_' % c, 1)}_X={''.join(xs)}_{rest}"
            rows.append((int(gene[m]) * 5_000 + int(rng.integers(-100, 100)), nm, 16 if gene[m] & 1 else 0, r["length"]))
    rows.sort(key=lambda t: t[0])
    n = len(rows)
    enc = [t[1].encode() for t in rows]
    noff = np.zeros(n + 1, dtype=np.uint32)
    noff[1:] = np.cumsum([len(e) for e in enc])
    nbuf = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8)
    coff = np.arange(n + 1, dtype=np.uint32)
    cbuf = np.array([(t[3] << 4) | 0 for t in rows] + [0], dtype=np.uint32)  # one M operation per record
    fl = np.array([t[2] for t in rows], dtype=np.uint16)
    p0 = np.array([max(t[0], 0) + 1_000_000 for t in rows], dtype=np.int32)
    out = np.zeros(n, dtype=lib.UMI_TAG_DTYPE)
    nd = ctypes.c_int32(0)
    import threading

    def make_call(c, o, ndv, threads):
        cfg = lib.AssignUmisConfig()
        c._check(c._lib.smi_assignumis_default_config(ctypes.byref(cfg)))
        cfg.n_threads = threads

        def call():
            c._check(c._lib.smi_assignumis_chunk(c._h, nbuf.ctypes.data, noff.ctypes.data, fl.ctypes.data, p0.ctypes.data, cbuf.ctypes.data,
                                                 coff.ctypes.data, n, ctypes.byref(cfg), o.ctypes.data, ctypes.byref(ndv)))
        return call

    base = {"records": n, "molecules": len(recs), "copies": copies, "loci": genes}
    for label, env in (("host_path", "1"), ("device_stage", None)):
        if env:
            os.environ["SMI_AU_HOST"] = env
        for threads in (1, 16):
            dt = timed(make_call(ctx, out, nd, threads))
            res[f"assignumis_chunk_{label}_{threads}_threads"] = dict(base, ms=dt * 1e3, records_per_s=n / dt, clustered=int((out["flags"] & 4 != 0).sum()),
                                                                     with_region=int((out["region"] >= 0).sum()))
        os.environ.pop("SMI_AU_HOST", None)
    res["assignumis_chunk_device_stage_1_threads"]["note"] = ("one smi_assignumis_chunk call: names / flags / positions / CIGARs up, K-UPARSE, region grouping on "
                                                              "the host (two threads), key sort, K-UMI, K-UCLUST, K-UTAG, tags down")
    # several chunks side by side on worker lanes of the one GPU (UmiFinderWorker runs several OneBatchExecutors the same way)
    runs = []
    for lanes in (1, 2, 4, 8, 16):
        ctxs = [ctx] + [ctx.lane() for _ in range(lanes - 1)]
        outs = [np.zeros(n, dtype=lib.UMI_TAG_DTYPE) for _ in ctxs]
        calls = [make_call(c, o, ctypes.c_int32(0), 2) for c, o in zip(ctxs, outs)]
        for f in calls:
            f()
        per = 6

        def worker(f):
            for _ in range(per):
                f()
        th = [threading.Thread(target=worker, args=(f,)) for f in calls]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        same = all(o.tobytes() == outs[0].tobytes() for o in outs)
        runs.append({"lanes": lanes, "records_per_s": n * lanes * per / dt, "ms_per_chunk": dt / per * 1e3, "same_tags_on_every_lane": same})
        for c in ctxs[1:]:
            c.close()
    res["assignumis_device_stage_lanes"] = {"records_per_chunk": n, "runs": runs, "best": max(runs, key=lambda r: r["records_per_s"])}


def leg_chimera(pkg, synth, ctx, dev, wl, used, res):
    # ---- chimera splitter: K-PACKR + K-CHIM on whole reads; 10 % of the records are two molecules joined -----------
    n = 1_000_000
    rd = synth.gen_reads(n, used, seed=7, device=dev)
    buf, offs = synth.materialize_device(rd)
    keep = torch.ones(n + 1, dtype=torch.bool, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    keep[1:n][torch.rand(n - 1, device=dev, generator=g) < 0.1] = False  # dropping an offset joins two neighbouring reads
    offs = offs[keep].contiguous()
    n = offs.numel() - 1
    total = int(offs[-1])
    planes = torch.zeros(ctx.read_planes_words(total, n), dtype=torch.int32, device=dev)
    cres = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    ccfg = ctx.chimera_config()
    dt_pack = timed(lambda: ctx.pack_reads_device(buf, offs, n, total, planes))
    # K-PACK (read ends for K-SCAN), without and with the quality string (pass 1 sums every quality)
    ends = torch.zeros((28, 2 * n), dtype=torch.int32, device=dev)
    lens = torch.zeros(n, dtype=torch.int32, device=dev)
    dt_ends = timed(lambda: ctx.pack_ends_device(buf, None, offs, n, ends, lens))
    qt = torch.zeros((n, 224), dtype=torch.uint8, device=dev)
    qs = torch.zeros(n, dtype=torch.int32, device=dev)
    dt_ends_q = timed(lambda: ctx.pack_ends_device(buf, buf, offs, n, ends, lens, qt, qs))
    res["pack_ends"] = {"reads": n, "ms": dt_ends * 1e3, "reads_per_s": n / dt_ends, "with_quals_ms": dt_ends_q * 1e3,
                        "with_quals_GBps": 2 * total / dt_ends_q / 1e9}
    del ends, lens, qt, qs
    dt = timed(lambda: ctx.chimera_device(planes, offs, n, total, ccfg, cres))
    cr = cres.cpu().numpy().view(pkg.CHIMERA_RESULT_DTYPE).reshape(-1)
    res["chimera"] = {"reads": n, "bases": total, "pack_ms": dt_pack * 1e3, "ms": dt * 1e3, "reads_per_s": n / dt,
                      "bases_per_s": total / dt, "pack_GBps": total / dt_pack / 1e9,
                      "split_frac": float((cr["n_split"] > 0).mean()), "multi_frac": float((cr["flags"] & 1).mean()),
                      "overflow": int((cr["flags"] & 4).sum()),
                      "n_matches_bit0_frac": float((cr["n_matches"] & 1).mean()), "n_matches_bit1_frac": float(((cr["n_matches"] >> 1) & 1).mean())}


def leg_fastq(pkg, synth, ctx, dev, wl, used, res):
    from sicelore_amd import lib as libmod

    # ---- K-FQ: FASTQ text -> record index -> contiguous reads; text built on the device from synthetic reads -----------
    n = 500_000
    rd = synth.gen_reads(n, used, seed=9, device=dev)
    text, buf, offs = synth.fastq_text_device(rd)
    total = int(text.numel())
    cap = n + 2
    line = torch.zeros(4 * cap + 8, dtype=torch.int64, device=dev)
    ns, ss, qs = (torch.zeros(cap, dtype=torch.int64, device=dev) for _ in range(3))
    nl, sl = (torch.zeros(cap, dtype=torch.int32, device=dev) for _ in range(2))
    o = torch.zeros(cap + 1, dtype=torch.int64, device=dev)
    out = torch.zeros(int(offs[-1]), dtype=torch.uint8, device=dev)
    state = {}

    def run():
        state["n"], state["err"] = ctx.fastq_index_device(text, total, line, ns, nl, ss, sl, qs, o, cap)
        ctx.fastq_gather_device(text, ss, o, state["n"], out)

    dt = timed(run)
    assert state["n"] == n and state["err"] == 0 and bool((out == buf).all())
    res["fastq_ingest"] = {"reads": n, "text_bytes": total, "ms": dt * 1e3, "text_GBps": total / dt / 1e9,
                           "reads_per_s": n / dt}
    # ---- K-WRITE: scan + barcode results of the same chunk -> `passed` / `failed` FASTQ text on the device ---------
    quals = torch.zeros(int(offs[-1]), dtype=torch.uint8, device=dev)
    ctx.fastq_gather_device(text, qs, o, n, quals)
    ends = torch.zeros((28, 2 * n), dtype=torch.int32, device=dev)
    lens32 = torch.zeros(n, dtype=torch.int32, device=dev)
    qt = torch.zeros((n, 224), dtype=torch.uint8, device=dev)
    qsum = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.pack_ends_device(out, quals, o[:n + 1], n, ends, lens32, qt, qsum)
    scan = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    win = torch.zeros((n, 2), dtype=torch.int64, device=dev)
    ctx.scan_device(ends, lens32, n, ctx.scan_config(2), scan, win, qt, qsum)
    bc = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    ctx.set_barcode_set_device(used.to(torch.int32), mode=0)
    ctx.bc_match_device(win, bc, n, max_ed=1)
    capw = 2 * int(offs[-1]) + total + 320 * n
    out_p = torch.empty(capw, dtype=torch.uint8, device=dev)
    out_f = torch.empty(capw, dtype=torch.uint8, device=dev)
    rec_off_w = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    is_p = torch.zeros(n, dtype=torch.uint8, device=dev)

    def write():
        state["tot"] = ctx.fastq_write_device(text, line, out, quals, o[:n + 1], None, None, scan, bc, None, n, 1, out_p, out_f,
                                              rec_off_w, is_p)

    dtw = timed(write)
    wb = state["tot"][0] + state["tot"][1]
    # ---- the native chunk worker: host FASTQ text -> passed / failed text on the host (PCIe in both directions included) ----
    pin = libmod.PinnedBuffer(total)
    pin.array[:] = text.cpu().numpy()
    ctx.scanfastq_pass2_chunk(pin.array, copy=False)  # warm-up: arena and pinned output buffers of this size
    t0 = time.perf_counter()
    pp, ff, inf = ctx.scanfastq_pass2_chunk(pin.array, copy=False)
    dth = time.perf_counter() - t0
    n_out_bytes = int(pp.size + ff.size)
    t0 = time.perf_counter()
    ctx.scanfastq_pass2_chunk(pin.array.copy(), copy=False)  # the same from pageable memory
    dtp = time.perf_counter() - t0
    # three worker threads, each with its own context (stream, arena, pinned output) on the same GPU, as the reference runs
    # nCPU Parser workers: the transfers of one chunk overlap the kernels of another
    import threading

    n_ctx, per_thread = int(os.environ.get("SMI_MB_LANES", "3")), 3
    # worker LANES of the one context (smi_ctx_create_lane): own stream / arena / pinned buffers, the owner's barcode set
    extra = [ctx.lane() for _ in range(n_ctx - 1)]
    ctxs = [ctx] + extra
    pins = [pin] + [libmod.PinnedBuffer(total) for _ in extra]
    for pb in pins[1:]:
        pb.array[:] = pin.array
    for c, pb in zip(ctxs, pins):
        c.scanfastq_pass2_chunk(pb.array, copy=False)  # warm-up
    def worker(c, pb):
        for _ in range(per_thread):
            c.scanfastq_pass2_chunk(pb.array, copy=False)
    th = [threading.Thread(target=worker, args=(c, pb)) for c, pb in zip(ctxs, pins)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dtm = time.perf_counter() - t0
    for c in extra:
        c.close()
    for pb in pins[1:]:
        pb.close()
    pin.close()
    res["pass2_chunk_host_to_host_lanes"] = {"lanes": n_ctx, "reads": n * n_ctx * per_thread, "ms": dtm * 1e3,
                                             "reads_per_s": n * n_ctx * per_thread / dtm,
                                             "GB_per_s_both_directions": (total + n_out_bytes) * n_ctx * per_thread / dtm / 1e9,
                                             "note": f"{n_ctx} host threads, one worker lane each (one context, one barcode set) on one GPU, 3 chunks each"}
    res["pass2_chunk_host_to_host"] = {"reads": n, "text_in_bytes": total, "text_out_bytes": n_out_bytes,
                                       "records_out": inf["n_records_out"], "ms": dth * 1e3, "reads_per_s": n / dth,
                                       "pageable_input_ms": dtp * 1e3,
                                       "note": "one smi_scanfastq_pass2_chunk call: H2D of the text (page-locked buffer from smi_host_alloc), "
                                               "every kernel incl. the chimera splitter, D2H of both streams into the context's pinned buffers"}
    res["fastq_write"] = {"reads": n, "passed": state["tot"][2], "out_bytes": wb, "ms": dtw * 1e3, "out_GBps": wb / dtw / 1e9,
                          "reads_per_s": n / dtw}


def leg_deflate(pkg, synth, ctx, dev, wl, used, res):
    """K-DEFLATE on the FASTQ text of a pass-2 chunk (qualities drawn uniformly from 29 values per base, as run_files.write_synthetic_dir
    does): GB/s of text, size against zlib level 6 and level 1 on a 64 MB sample of the same text"""
    import zlib

    n = int(os.environ.get("SMI_MB_READS", "500000"))
    rd = synth.gen_reads(n, used, seed=9, device=dev, q_mean=20.0)
    text = synth.fastq_text_device(rd)[0]
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    is_q = text == ord("I")
    text[is_q] = torch.randint(35, 64, (int(is_q.sum()),), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
    del rd, is_q
    total = int(text.numel())
    d_out = torch.empty(int(ctx._lib.smi_deflate_bound(total)), dtype=torch.uint8, device=dev)
    state = {}

    def run():
        state["z"] = ctx.gzip_device(text, total, d_out=d_out)

    dt = timed(run, reps=5)
    z = state["z"]
    sample = text[:64 << 20].cpu().numpy().tobytes()
    zs = ctx.gzip_device(text[:64 << 20].contiguous(), len(sample)).cpu().numpy().tobytes()
    assert zlib.decompress(zs, wbits=31) == sample
    t0 = time.perf_counter()
    z6 = len(zlib.compress(sample, 6))
    t6 = time.perf_counter() - t0
    t0 = time.perf_counter()
    z1 = len(zlib.compress(sample, 1))
    t1 = time.perf_counter() - t0
    res["gzip_device"] = {"text_bytes": total, "member_bytes": int(z.numel()), "ratio": total / int(z.numel()), "ms": dt * 1e3, "text_GBps": total / dt / 1e9,
                          "sample_bytes": len(sample), "sample_member_bytes": len(zs), "sample_zlib6_bytes": z6, "sample_zlib1_bytes": z1,
                          "size_vs_zlib6": len(zs) / z6, "zlib6_MBps_one_thread": len(sample) / t6 / 1e6, "zlib1_MBps_one_thread": len(sample) / t1 / 1e6,
                          "note": "one call incl. the 16-byte read-back of size and flags; literals-only dynamic Huffman blocks of 64 KiB"}


def leg_inflate(pkg, synth, ctx, dev, wl, used, res):
    """K-INFLATE: 64 gzip files of FASTQ text (zlib level 6 and level 1, qualities uniform over 29 values per base) resident in HBM -> their
    text in HBM; against zlib on one host thread"""
    import ctypes
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    n = int(os.environ.get("SMI_MB_READS", "500000"))
    n_files = int(os.environ.get("SMI_MB_FILES", "64"))
    rd = synth.gen_reads(n, used, seed=9, device=dev, q_mean=20.0)
    text = synth.fastq_text_device(rd)[0]
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    is_q = text == ord("I")
    text[is_q] = torch.randint(35, 64, (int(is_q.sum()),), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
    host = text.cpu().numpy()
    del rd, is_q, text
    cuts = np.linspace(0, host.size, n_files + 1).astype(np.int64)
    parts = [host[cuts[i]:cuts[i + 1]] for i in range(n_files)]
    for level in (6, 1):
        def gz(a, level=level):
            c = zlib.compressobj(level, zlib.DEFLATED, 31)
            return c.compress(memoryview(a)) + c.flush()

        with ThreadPoolExecutor(16) as pool:
            files = list(pool.map(gz, parts))
        t0 = time.perf_counter()
        zlib.decompress(files[0], wbits=31)
        t_zlib = time.perf_counter() - t0
        in_off, at = [], 0
        for f in files:
            in_off.append(at)
            at = (at + len(f) + 511) & ~511
        hbuf = np.zeros(at + 1024, dtype=np.uint8)
        for f, o in zip(files, in_off):
            hbuf[o:o + len(f)] = np.frombuffer(f, dtype=np.uint8)
        d_in = torch.from_numpy(hbuf).to(dev)
        out_off, at = [], 0
        for p in parts:
            out_off.append(at)
            at = (at + p.size + 255) & ~255
        d_out = torch.zeros(at, dtype=torch.uint8, device=dev)
        S = np.zeros((n_files, 4), dtype=np.uint64)
        S[:, 0], S[:, 1], S[:, 2], S[:, 3] = in_off, [len(f) for f in files], out_off, [p.size for p in parts]
        R = np.zeros(n_files, dtype=np.dtype([("out_len", "<u8"), ("status", "<u4"), ("n_members", "<u4")]))

        def run():
            ctx._check(ctx._lib.smi_gz_inflate_device(ctx._h, d_in.data_ptr(), S.ctypes.data, n_files, d_out.data_ptr(), R.ctypes.data, None))

        dt = timed(run, reps=3)
        assert (R["status"] == 0).all(), R["status"]
        got = d_out.cpu().numpy()
        assert all((got[o:o + p.size] == p).all() for o, p in zip(out_off, parts))
        res[f"gz_inflate_device_level{level}"] = {"files": n_files, "gz_bytes": int(sum(len(f) for f in files)), "text_bytes": int(host.size), "ms": dt * 1e3,
                                                  "text_GBps": host.size / dt / 1e9, "text_MBps_per_file": host.size / dt / 1e6 / n_files,
                                                  "zlib_one_thread_text_MBps": parts[0].size / t_zlib / 1e6,
                                                  "note": "one call: kernel, CRC-32 check of every member, results on the host"}


def leg_packed(pkg, synth, ctx, dev, wl, used, res):
    """host text -> host text through the PACKED boundary (smi_scanfastq_pass2_chunk_packed: index / bit-planes / records on host threads,
    0.7 KB per read over the link instead of 5 KB), one lane with T threads and several lanes sharing the box's host cores"""
    import threading

    from sicelore_amd import lib as libmod

    n = int(os.environ.get("SMI_MB_READS", "500000"))
    rd = synth.gen_reads(n, used, seed=9, device=dev)
    text, buf, offs = synth.fastq_text_device(rd)
    total = int(text.numel())
    del rd, buf
    ctx.set_barcode_set_device(used.to(torch.int32), mode=0)
    pin = libmod.PinnedBuffer(total)
    pin.array[:] = text.cpu().numpy()
    del text
    cores = len(os.sched_getaffinity(0))
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q[0] == "max" else float(q[0]) / float(q[1])
    except Exception:
        quota = None
    out = {"reads_per_chunk": n, "text_bytes": total, "host_cpus_visible": cores, "host_cpu_quota": quota, "runs": []}
    pt, ft, _ = ctx.scanfastq_pass2_chunk(pin.array, copy=True)                       # the text worker: the bytes to match
    pp, fp, info = ctx.scanfastq_pass2_chunk(pin.array, copy=True, packed=True, n_threads=16)
    out["equals_text_worker"] = bool(pt == pp and ft == fp)
    out["text_out_bytes"] = len(pp) + len(fp)
    os.environ["SMI_PK_TIMING"] = "1"
    ctx.scanfastq_pass2_chunk(pin.array, copy=False, packed=True, n_threads=16)      # stages of one call on stderr
    del os.environ["SMI_PK_TIMING"]
    for lanes, threads in ((1, 16), (1, 32), (2, 8), (2, 16), (4, 4), (4, 8), (8, 2), (8, 4), (16, 1), (16, 2)):
        ctxs = [ctx] + [ctx.lane() for _ in range(lanes - 1)]
        pins = [pin] + [libmod.PinnedBuffer(total) for _ in range(lanes - 1)]
        for pb in pins[1:]:
            pb.array[:] = pin.array
        for c, pb in zip(ctxs, pins):
            c.scanfastq_pass2_chunk(pb.array, copy=False, packed=True, n_threads=threads)   # warm-up: arena, pinned buffers
        per = 3

        def worker(c, pb):
            for _ in range(per):
                c.scanfastq_pass2_chunk(pb.array, copy=False, packed=True, n_threads=threads)

        th = [threading.Thread(target=worker, args=(c, pb)) for c, pb in zip(ctxs, pins)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        out["runs"].append({"lanes": lanes, "threads_per_lane": threads, "reads_per_s": n * lanes * per / dt, "ms_per_chunk": dt / per * 1e3})
        for c in ctxs[1:]:
            c.close()
        for pb in pins[1:]:
            pb.close()
    pin.close()
    out["best"] = max(out["runs"], key=lambda r: r["reads_per_s"])
    res["pass2_chunk_packed_host_to_host"] = out


if __name__ == "__main__":
    main()
