"""Seed sweep of the GPU parity checks (run on the GPU box): the same comparisons as tests/ (-m gpu), over many seeds and
set densities, to look for rare disagreements with the oracle.  usage: python tools/fuzz_parity.py [minutes] [first seed]
Prints one line per leg and seed; exits non-zero at the first mismatch."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

COMP = bytes.maketrans(b"ACGTN", b"TGCAN")
FIELDS = ("bc", "ed", "ed_sec", "offset", "ins_minus_del")


def check_bc(pkg, synth, sor, ctx, seed):
    rng = np.random.default_rng(seed)
    n_wl = int(rng.choice([2_000, 50_000, 400_000, 3_600_000]))
    five = bool(seed & 1)
    max_ed = int(rng.choice([0, 1, 1, 1, 2]))
    wl = synth.make_whitelist(n_wl, seed=seed)
    used = synth.pick_used(wl, min(3000, n_wl // 2), seed=seed + 1)
    search = wl if rng.random() < 0.5 else used
    n = 4000 if max_ed == 2 else 60_000
    reg = synth.gen_bc_region(n, used, seed=seed + 2, five_prime=five, n_rate=float(rng.choice([0.0, 0.002, 0.02])))
    win = synth.pack_windows(reg["codes"], reg["ae"], five)
    keys = search.numpy().astype(np.uint64)
    ctx.set_barcode_set(keys, mode=1 if search is wl else 0)
    d_out = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    ctx.bc_match_device(win.cuda(), d_out, n, max_ed=max_ed, five_prime=five)
    got = d_out.cpu().numpy().view(pkg.BC_RESULT_DTYPE).reshape(-1)
    st, exp = sor.assign_batch(sor.BarcodeSet(search.numpy()), reg["codes"].numpy(), reg["ae"].numpy(), max_ed=max_ed, five_prime=five,
                               n_threads=16)
    exp_found = np.where(st < 0, st, exp["found"])
    ok = (got["found"] == exp_found).all()
    sel = exp_found == 1
    for f in FIELDS:
        ok = ok and (got[f][sel].astype(np.int64) == (exp[f][sel].astype(np.int64) & (0xFFFFFFFF if f == "bc" else -1))).all()
    ok = ok and (got["n_matches"][st >= 0] == exp["n_matches"][st >= 0]).all()
    return ok, f"bc seed={seed} wl={n_wl} set={'wl' if search is wl else 'used'} ed={max_ed} 5p={five} found={int(sel.sum())}"


def check_records(pkg, synth, sor, ctx, seed):
    import bammodel  # noqa: F401  (tests dir on the path)
    from test_write_gpu import _fastq, _oracle_records

    rng = np.random.default_rng(seed)
    wl = synth.make_whitelist(20_000, seed=seed)
    used = synth.pick_used(wl, 150, seed=seed + 1)
    reads = synth.gen_reads(160, used, seed=seed + 2, n_rate=float(rng.choice([0.0, 0.003])), err=float(rng.choice([0.03, 0.063, 0.1])))
    chim = synth.make_chimeras(reads, 200, seed=seed + 3)
    seqs, quals = [c[0] for c in chim], [c[1] for c in chim]
    ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    trim = bool(seed & 1)
    got_p, got_f, info = ctx.scanfastq_pass2_chunk(_fastq(seqs, quals), first_read_id=7 + seed, trim_fastq=trim)
    exp_p, exp_f, n_p = _oracle_records(sor, sor.BarcodeSet(used.numpy()), seqs, quals, 1, {}, 7 + seed, trim=trim)
    return got_p == exp_p and got_f == exp_f and info["n_passed"] == n_p, f"records seed={seed} trim={trim} passed={n_p} out={info['n_records_out']}"


def check_packed(pkg, synth, sor, ctx, seed):
    """round 3: the packed boundary (host index / planes / writer in a random SIMD form on 1..5 threads) == the oracle's records; its
    statistics == the text worker's; the text worker with cfg.compress returns members that inflate to the same text"""
    import gzip

    from test_write_gpu import _fastq, _oracle_records

    rng = np.random.default_rng(seed)
    wl = synth.make_whitelist(20_000, seed=seed)
    used = synth.pick_used(wl, 150, seed=seed + 1)
    reads = synth.gen_reads(160, used, seed=seed + 2, n_rate=float(rng.choice([0.0, 0.003])), err=float(rng.choice([0.03, 0.063, 0.1])))
    chim = synth.make_chimeras(reads, 220, seed=seed + 3)
    seqs, quals = [c[0] for c in chim], [c[1] for c in chim]
    ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    trim = bool(seed & 2)
    text = _fastq(seqs, quals)
    os.environ["SMI_HOST_SIMD"] = str(int(rng.integers(0, 3)))
    try:
        got_p, got_f, info = ctx.scanfastq_pass2_chunk(text, first_read_id=7 + seed, trim_fastq=trim, packed=True, n_threads=int(rng.integers(1, 6)),
                                                       want_results=True)
    finally:
        del os.environ["SMI_HOST_SIMD"]
    exp_p, exp_f, n_p = _oracle_records(sor, sor.BarcodeSet(used.numpy()), seqs, quals, 1, {}, 7 + seed, trim=trim)
    ok = got_p == exp_p and got_f == exp_f and info["n_passed"] == n_p
    zp, zf, zinfo = ctx.scanfastq_pass2_chunk(text, first_read_id=7 + seed, trim_fastq=trim, compress=True, want_results=True)
    ok = ok and gzip.decompress(bytes(zp)) == exp_p and gzip.decompress(bytes(zf)) == exp_f and bool((zinfo["stats"] == info["stats"]).all())
    return ok, f"packed seed={seed} trim={trim} passed={n_p} out={info['n_records_out']} member={len(zp)}"


def check_umi_stage(pkg, synth, sor, ctx, seed):
    """round 3: smi_assignumis_chunk with the stage on the device == the host stage of round 2 (which tests/ hold to the oracle), 3' / 5'"""
    from test_umi_stage_gpu import _both_paths, _chunk

    rng = np.random.default_rng(seed)
    five = bool(seed & 1)
    names, flags, pos0, cigars = _chunk(pkg, synth, ctx, int(rng.integers(200, 900)), int(rng.integers(2, 9)), int(rng.integers(5, 120)), seed, five)
    kw = dict(five_prime=five, n_threads=int(rng.integers(1, 5)), keep_data_end=bool(rng.random() < 0.3))
    if rng.random() < 0.3:
        kw["bc_edit_limit"] = int(rng.integers(0, 2))
    tags, n_done = _both_paths(ctx, names, flags, pos0, cigars, **kw)          # asserts equality
    return True, f"umi_stage seed={seed} 5p={five} records={len(names)} done={n_done} clustered={int((tags['flags'] & 1 != 0).sum())}"


def check_umi_pairs(pkg, synth, sor, ctx, seed):
    """K-UMI alone (round 5: two Myers recurrences per register, match masks from an LDS table): matrices of groups on both sides of every
    tile edge against the oracle's, windows with close and distant UMIs, with and without N"""
    rng = np.random.default_rng(seed)
    sizes = [int(x) for x in rng.choice([1, 2, 3, 7, 31, 63, 64, 65, 66, 127, 128, 129, 200, 257, 300], size=int(rng.integers(3, 9)))]
    codes = [1, 2, 4, 8] if seed % 3 else [1, 2, 4, 8, 15, 15]     # the window packers know A, G, C, T and N (anything else) -- no other code exists
    ws = []
    for n in sizes:
        n_umi = max(1, n // int(rng.choice([2, 3, 10])))
        umis = rng.choice([1, 2, 4, 8] if seed % 5 else [1, 2], size=(n_umi, 14)).astype(np.uint8)     # (two-letter UMIs: many near ties)
        for _ in range(n):
            w = umis[rng.integers(n_umi)].copy()
            for _ in range(int(rng.integers(0, 5))):
                p_, op = int(rng.integers(14)), int(rng.integers(3))
                if op == 0:
                    w[p_] = rng.choice(codes)
                elif op == 1:
                    w[p_ + 1:] = w[p_:-1]
                    w[p_] = rng.choice(codes)
                else:
                    w[p_:-1] = w[p_ + 1:]
                    w[-1] = rng.choice(codes)
            ws.append(w)
    ws = np.array(ws, dtype=np.uint8)
    packed = np.zeros(ws.shape[0], dtype=np.uint64)
    for k in range(14):
        packed |= ws[:, k].astype(np.uint64) << np.uint64(4 * k)
    go, po, mo = ctx.umi_offsets(sizes)
    d_out = torch.full((int(mo[-1]),), 255, dtype=torch.uint8, device="cuda")
    ctx.umi_dist_device(torch.from_numpy(packed.view(np.int64)).cuda(), torch.from_numpy(go.view(np.int32)).cuda(), torch.from_numpy(po.view(np.int64)).cuda(),
                        torch.from_numpy(mo.view(np.int64)).cuda(), len(sizes), int(po[-1]), d_out)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    ok = True
    for g, n in enumerate(sizes):
        ok = ok and bool((out[int(mo[g]):int(mo[g + 1])].reshape(n, n) == sor.umi_matrix(ws[go[g]:go[g + 1]])).all())
    return ok, f"umi_pairs seed={seed} groups={sizes} pairs={int(po[-1])}"


def check_deflate(pkg, synth, sor, ctx, seed):
    """round 3: K-DEFLATE round trip on random lengths and alphabets (zlib is the inflater)"""
    import zlib

    rng = np.random.default_rng(seed)
    n = int(rng.choice([0, 1, 63, 64, 65, 4096, 65535, 65536, 65537, int(rng.integers(1, 3_000_000))]))
    k = int(rng.choice([1, 2, 4, 5, 16, 64, 256]))
    alpha = rng.choice(256, size=k, replace=False).astype(np.uint8)
    p = rng.dirichlet(np.full(k, float(rng.choice([0.05, 0.5, 5.0]))))
    data = rng.choice(alpha, size=n, p=p).tobytes()
    d_in = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda() if n else torch.zeros(0, dtype=torch.uint8, device="cuda")
    out = ctx.gzip_device(d_in, n).cpu().numpy().tobytes()
    back = zlib.decompress(out, wbits=31)
    return back == data, f"deflate seed={seed} n={n} alphabet={k} member={len(out)}"


def check_inflate(pkg, synth, sor, ctx, seed):
    """round 3: K-INFLATE on what zlib makes of random data at a random level / strategy / window, several files and members per call"""
    import zlib

    rng = np.random.default_rng(seed)
    files, texts = [], []
    for _ in range(int(rng.integers(1, 9))):
        members, text = [], b""
        for _m in range(int(rng.choice([1, 1, 1, 2, 5]))):
            n = int(rng.choice([0, 1, 100, 70_000, int(rng.integers(1, 400_000))]))
            k = int(rng.choice([1, 4, 5, 30, 256]))
            alpha = rng.choice(256, size=k, replace=False).astype(np.uint8)
            data = rng.choice(alpha, size=n, p=rng.dirichlet(np.full(k, float(rng.choice([0.05, 0.5, 5.0]))))).tobytes()
            if rng.random() < 0.4 and n > 1000:                       # repeats at long and short distances
                data = data[:n // 3] * 2 + data[n // 3:] + bytes([data[0]]) * int(rng.integers(1, 600))
            c = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, 16 + int(rng.integers(9, 16)), int(rng.integers(1, 10)),
                                 int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])))
            members.append(c.compress(data) + c.flush())
            text += data
        files.append(b"".join(members))
        texts.append(text)
    out, offs, lens, status, _ = ctx.gz_inflate_device(files, [len(t) for t in texts])
    host = out.cpu().numpy()
    ok = all(int(status[i]) == 0 and int(lens[i]) == len(t) and host[offs[i]:offs[i] + lens[i]].tobytes() == t for i, t in enumerate(texts))
    return ok, f"inflate seed={seed} files={len(files)} bytes={sum(len(t) for t in texts)}"


def check_host_inflate(pkg, synth, sor, ctx, seed):
    """round 3: the host's own gzip decoder (smi_gz_inflate_into) on the same kind of files, and the BGZF reader over it"""
    import zlib

    lib = importlib.import_module(graft.PKG_NAME + ".lib")
    rng = np.random.default_rng(seed)
    members, text = [], b""
    for _m in range(int(rng.choice([1, 1, 2, 6]))):
        n = int(rng.choice([0, 1, 100, 70_000, int(rng.integers(1, 900_000))]))
        k = int(rng.choice([1, 4, 5, 30, 256]))
        alpha = rng.choice(256, size=k, replace=False).astype(np.uint8)
        data = rng.choice(alpha, size=n, p=rng.dirichlet(np.full(k, float(rng.choice([0.05, 0.5, 5.0]))))).tobytes()
        if rng.random() < 0.5 and n > 1000:
            data = data[:n // 3] * 2 + data[n // 3:] + bytes([data[0]]) * int(rng.integers(1, 600))
        c = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, 16 + int(rng.integers(9, 16)), int(rng.integers(1, 10)),
                             int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])))
        members.append(c.compress(data) + c.flush())
        text += data
    got = lib.gz_inflate(np.frombuffer(b"".join(members), dtype=np.uint8)).tobytes()
    z = lib.bgzf_deflate(text, level=int(rng.integers(0, 10)), n_threads=3)
    back, used = lib.bgzf_inflate(z, n_threads=3)
    return got == text and back.tobytes() == text and used == z.size, f"host_inflate seed={seed} members={len(members)} bytes={len(text)}"


def check_bam_writer(pkg, synth, sor, ctx, seed):
    """round 3: smi_bam_write_batch against the Python mirror that the reference-executed fixtures pin (tests/test_bam.py's generator)"""
    import test_bam

    lib = importlib.import_module(graft.PKG_NAME + ".lib")
    au = importlib.import_module(graft.PKG_NAME + ".assignumis")
    rng = np.random.default_rng(seed)
    five, trunc, lim = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), [None, 0, 1][int(rng.integers(0, 3))]
    bc, exp_bc, umi, exp_umi, g1, g2 = test_bam._write_batch_case(lib, au, seed, five, trunc, lim)
    return bc == exp_bc and umi == exp_umi and g1 == g2, f"bam_writer seed={seed} five_prime={five} -w={trunc} -b={lim} bytes={len(bc)}"


def check_scan_params(pkg, synth, sor, ctx, seed):
    """round 5: K-SCAN with a random polyA window of `scanfastq -p / -f / -w` (smi_ctx_set_polya's range: 5 <= length <= 30, window + length + 10 <= 175)
    against the oracle with the same parameters, on ordinary reads (both passes) and on ends crowded with T / A runs"""
    from test_scan_gpu import AD, _ascii_batch, _compare, _scan_gpu, _t_rich_reads

    rng = np.random.default_rng(seed)
    ml = int(rng.integers(5, 31))
    win = int(rng.integers(20, 175 - ml - 10 + 1))
    frac = float(rng.choice([0.5, 0.6, 0.7, 0.75, 0.8, 0.9, 1.0, round(float(rng.uniform(0.4, 1.0)), 3)]))
    polya = (ml, frac, win)
    par = sor.default_scan_params()
    par["polya_len"], par["polya_frac"], par["window_polya"] = polya
    wl = synth.make_whitelist(20_000, seed=seed)
    used = synth.pick_used(wl, 200, seed=seed + 1)
    n = 2000
    reads = synth.gen_reads(n, used, seed=seed + 2, n_rate=float(rng.choice([0.0, 0.003])), err=float(rng.choice([0.03, 0.063, 0.1])))
    ra, qa, offs = _ascii_batch(synth, reads, n, short_every=40)
    found = 0
    try:
        for pass_no in (2, 1):
            got, _, _, _ = _scan_gpu(pkg, ctx, ra, qa, offs, pass_no, polya=polya)
            st, exp = sor.scan_batch_3p(ra, qa, offs, AD[pass_no], params=par, n_threads=16)
            found += _compare(got, st, exp, pass1=True)
        ra, qa, offs = _t_rich_reads(3000, seed=seed + 3)
        got, _, _, _ = _scan_gpu(pkg, ctx, ra, qa, offs, 2, polya=polya)
        st, exp = sor.scan_batch_3p(ra, qa, offs, AD[2], params=par, n_threads=16)
        _compare(got, st, exp, pass1=True)
        ok = True
    except AssertionError as e:
        return False, f"scan_params seed={seed} polya={polya}: {e}"
    return ok, f"scan_params seed={seed} polya={polya} found={found}"


def check_scan_knobs(pkg, synth, sor, ctx, seed):
    """round 6: K-SCAN under a random set of config.xml's scan knobs (smi_run_knobs: the read scan's TSO -- a random 16-mer or the shipped one with a few bases
    changed --, its window, mismatch limit and rescue rules; the adapter's mismatch limit; minAdapter3pMatches; the quality thresholds) against the oracle
    with the same values, pass 2 and pass 1 and the T-rich reads"""
    from test_knobs_gpu import _oracle_scan_params, _scan_with
    from test_scan_gpu import _ascii_batch, _compare, _t_rich_reads

    lib = importlib.import_module(graft.PKG_NAME + ".lib")
    rng = np.random.default_rng(seed)
    tso = list("AACGCAGAGTACATGG")
    if rng.integers(0, 3) == 0:
        tso = list(rng.choice(list("ACGT"), 16))
    else:
        for p in rng.choice(16, int(rng.integers(0, 4)), replace=False):
            tso[int(p)] = str(rng.choice(list("ACGT")))
    over = dict(tso_scan="".join(tso), tso_scan_window=int(rng.integers(16, 113)), tso_scan_max_mm=int(rng.integers(0, 9)), tso_scan_min_consec=int(rng.integers(4, 14)),
                tso_scan_min_two_best=int(rng.integers(8, 17)), adapter3p_max_mm=int(rng.integers(0, 6)), min_adapter_3p_matches=int(rng.integers(4, 11)),
                min_mean_bc_qv=int(rng.integers(5, 15)), min_mean_read_qv=int(rng.integers(5, 15)), min_read_length=int(rng.choice([180, 200, 400])))
    k = lib.run_knobs(**over)
    par = _oracle_scan_params(sor, k)
    wl = synth.make_whitelist(20_000, seed=seed)
    used = synth.pick_used(wl, 200, seed=seed + 1)
    n = 2000
    reads = synth.gen_reads(n, used, seed=seed + 2, n_rate=float(rng.choice([0.0, 0.003])), err=float(rng.choice([0.03, 0.063, 0.1])))
    ra, qa, offs = _ascii_batch(synth, reads, n, short_every=40)
    found = 0
    try:
        for pass_no in (2, 1):
            got = _scan_with(pkg, ctx, ra, qa, offs, ctx.scan_config(pass_no, knobs=k))
            ad = (k.adapter3p if pass_no == 2 else k.adapter3p_complete).decode()
            st, exp = sor.scan_batch_3p(ra, qa, offs, ad, max_mm=k.adapter3p_max_mm, params=par, n_threads=16)
            found += _compare(got, st, exp, pass1=True)
        ra, qa, offs = _t_rich_reads(3000, seed=seed + 3)
        got = _scan_with(pkg, ctx, ra, qa, offs, ctx.scan_config(2, knobs=k))
        st, exp = sor.scan_batch_3p(ra, qa, offs, k.adapter3p.decode(), max_mm=k.adapter3p_max_mm, params=par, n_threads=16)
        _compare(got, st, exp, pass1=True)
    except AssertionError as e:
        return False, f"scan_knobs seed={seed} {over}: {e}"
    return True, f"scan_knobs seed={seed} tso={over['tso_scan']} w={over['tso_scan_window']} mm={over['tso_scan_max_mm']} found={found}"


def main():
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    pkg = graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    sor = graft.load_oracle()
    sor.build()
    ctx = pkg.Context(0)
    t_end = time.time() + 60 * minutes
    seed, n_ok = (int(sys.argv[2]) if len(sys.argv) > 2 else 1000), 0
    while time.time() < t_end:
        for leg in ([check_bc, check_records] if os.environ.get("SMI_FUZZ_LEGS") == "r2" else [check_packed, check_umi_stage, check_deflate, check_inflate]
                    if os.environ.get("SMI_FUZZ_LEGS") == "r3" else [check_host_inflate, check_bam_writer] if os.environ.get("SMI_FUZZ_LEGS") == "host"
                    else [check_umi_pairs, check_umi_stage] if os.environ.get("SMI_FUZZ_LEGS") == "umi"
                    else [check_scan_params] if os.environ.get("SMI_FUZZ_LEGS") == "scan"
                    else [check_scan_knobs] if os.environ.get("SMI_FUZZ_LEGS") == "knobs"
                    else [check_umi_pairs, check_scan_params, check_bc, check_records, check_packed, check_umi_stage, check_deflate, check_inflate, check_host_inflate, check_bam_writer]):
            ok, msg = leg(pkg, synth, sor, ctx, seed)
            print(("ok   " if ok else "FAIL ") + msg, flush=True)
            if not ok:
                sys.exit(1)
            n_ok += 1
        seed += 7
    print(f"fuzz: {n_ok} legs passed")


if __name__ == "__main__":
    main()
