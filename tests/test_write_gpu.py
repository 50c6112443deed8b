"""K-WRITE: the `passed` / `failed` FASTQ text of pass 2 assembled on the device == the oracle's restatement of
FastqRecordExt.getRecordForWriting + htsjdk's BasicFastqWriter, record by record (chimera fragments, MULTI reads,
too-short reads, rk=, read ids of the passed records only, -u trimming, 5' barcoding, quality-header text, CR LF)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
COMP = bytes.maketrans(b"ACGTN", b"TGCAN")


def _fastq(seqs, quals, eol="\n", qh=lambda i: ""):
    return "".join(f"@read{i} runid=x ch={i % 9}{eol}{s}{eol}+{qh(i)}{eol}{q}{eol}" for i, (s, q) in enumerate(zip(seqs, quals))).encode()


def _oracle_records(sor, bset, seqs, quals, max_ed, rank_of, first_id, five_prime=False, trim=False, split=True, qh=lambda i: "",
                    noname_blank=False, scan_params=None):
    passed, failed = [], []
    rid = first_id
    for i, (s, q) in enumerate(zip(seqs, quals)):
        name = f"read{i}" if noname_blank else f"read{i} runid=x ch={i % 9}"
        splits, multi, raw = [], False, None
        if split:
            rc, splits, multi, _, raw = sor.chimera_split(s, sor.chimera_params(22 if five_prime else 28)) if five_prime else sor.chimera_split(s)
            assert rc == 0
        cuts = [0] + [p for _, p in splits] + [len(s)]
        for k in range(len(cuts) - 1):
            fs, fq = s[cuts[k]:cuts[k + 1]], q[cuts[k]:cuts[k + 1]]
            fname = sor.chimera_fragment_name(name, raw, k) if splits else name
            if five_prime:
                rc, sc = sor.scan_read_5p(fs, fq, "CTTCCGATCT")
            else:
                rc, sc = sor.scan_read_3p(fs, fq, "CTTCCGATCT", params=scan_params)
            assert rc == 0
            a = None
            if sc["adapter_found"] and not multi:
                stranded = fs.encode().translate(COMP)[::-1] if sc["reverse"] else fs.encode()
                rc2, a_ = sor.assign_barcode(bset, stranded, int(sc["adapter_end"]), max_ed=max_ed, five_prime=five_prime)
                if rc2 == 1:
                    a = a_
            rk = rank_of.get(int(a["bc"]) & 0xFFFFFFFF, 0) if a is not None else 0
            rec, ok = sor.fastq_record(fname, qh(i), fs, fq, sc, a, rank=rk, read_id=rid, five_prime=five_prime, trim_fastq=trim,
                                       force_failed=multi)
            assert rec is not None
            if ok:
                passed.append(rec)
                rid += 1  # GET_NEXT_READID() per passed record (FastqWriterThreadPool.java:L302)
            else:
                failed.append(rec)
    return b"".join(passed), b"".join(failed), len(passed)


def _reads(synth, n, seed):
    wl = synth.make_whitelist(20_000, seed=seed)
    used = synth.pick_used(wl, 150, seed=seed + 1)
    reads = synth.gen_reads(n, used, seed=seed + 2, n_rate=0.002)
    return used, reads


@pytest.mark.parametrize("trim", [False, True])
def test_records_equal_oracle_3p(pkg, synth, sor, gpu_ctx, trim):
    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    used, reads = _reads(synth, 260, 911)
    chim = synth.make_chimeras(reads, 340, seed=914)
    seqs = [c[0] for c in chim] + ["ACGT" * 30]  # one too-short read
    quals = [c[1] for c in chim] + ["5" * 120]
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    keys = np.sort(used.numpy().astype(np.uint64))
    ranks = (np.arange(keys.size) * 7 % 1000 + 1).astype(np.int32)
    rank_of = {int(k): int(r) for k, r in zip(keys, ranks)}
    rs = scanfastq.ReadScanner(gpu_ctx, max_ed=1)
    qh = lambda i: f"read{i} again" if i % 5 == 0 else ""  # noqa: E731
    got_p, got_f, info = rs.pass2_write_chunk(_fastq(seqs, quals, qh=qh), rank_keys=keys, rank_values=ranks, first_read_id=35 ** 2,
                                              trim_fastq=trim)
    exp_p, exp_f, n_p = _oracle_records(sor, sor.BarcodeSet(used.numpy()), seqs, quals, 1, rank_of, 35 ** 2, trim=trim, qh=qh)
    assert got_p == exp_p
    assert got_f == exp_f
    assert info["n_passed"] == n_p and n_p > 250 and info["n_records"] > len(seqs)
    assert got_p.count(b"_rk=") > 200 and got_f.count(b"_FAILED ") > 10
    # record offsets: every record starts with '@' in its own stream
    for off, ok in zip(info["rec_off"], info["is_passed"]):
        assert (got_p if ok else got_f)[int(off)] == ord("@")


def test_records_crlf_and_names_without_blank(pkg, synth, sor, gpu_ctx):
    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    used, reads = _reads(synth, 60, 931)
    chim = synth.make_chimeras(reads, 90, seed=934)
    seqs, quals = [c[0] for c in chim], [c[1] for c in chim]
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    rs = scanfastq.ReadScanner(gpu_ctx, max_ed=1)
    text = "".join(f"@read{i}\r\n{s}\r\n+\r\n{q}\r\n" for i, (s, q) in enumerate(zip(seqs, quals))).encode()
    got_p, got_f, info = rs.pass2_write_chunk(text)
    exp_p, exp_f, n_p = _oracle_records(sor, sor.BarcodeSet(used.numpy()), seqs, quals, 1, {}, 1, noname_blank=True)
    assert got_p == exp_p and got_f == exp_f and info["n_passed"] == n_p
    assert b"\r" not in got_p and b"sp1" not in got_p  # a name without a blank takes no fragment tag (String.replaceFirst)


@pytest.mark.parametrize("trim", [False, True])
def test_records_equal_oracle_5p(pkg, synth, sor, gpu_ctx, trim):
    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    wl = synth.make_whitelist(20_000, seed=951)
    used = synth.pick_used(wl, 150, seed=952)
    reads = synth.gen_reads_5p(200, used, seed=953)
    seqs, quals = zip(*(synth.materialize(reads, i) for i in range(200)))
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    rs = scanfastq.ReadScanner(gpu_ctx, max_ed=1, five_prime=True, dont_search_polya=True)
    got_p, got_f, info = rs.pass2_write_chunk(_fastq(seqs, quals), trim_fastq=trim)
    exp_p, exp_f, n_p = _oracle_records(sor, sor.BarcodeSet(used.numpy()), seqs, quals, 1, {}, 1, five_prime=True, trim=trim,
                                        split=False)
    assert got_p == exp_p and got_f == exp_f and info["n_passed"] == n_p and n_p > 120


def test_large_batch_round_trip_properties(pkg, synth, gpu_ctx):
    """200 k reads (no oracle at this size): the two streams re-read by K-FQ hold every record exactly once, with the same
    number of bases; reverse complement preserves the A+T / G+C / N totals; qualities keep their byte histogram"""
    import torch

    n = 200_000
    dev = torch.device("cuda", gpu_ctx.device)
    wl = synth.make_whitelist(50_000, seed=971, device=dev)
    used = synth.pick_used(wl, 300, seed=972)
    gpu_ctx.set_barcode_set_device(used.to(torch.int32), mode=0)
    rd = synth.gen_reads(n, used, seed=973, device=dev)
    text, bases, offs0 = synth.fastq_text_device(rd)
    cap = n + 2
    z64 = lambda k: torch.zeros(k, dtype=torch.int64, device=dev)  # noqa: E731
    z32 = lambda k: torch.zeros(k, dtype=torch.int32, device=dev)  # noqa: E731
    line, ns, ss, qs, offs, nl, sl = z64(4 * cap + 8), z64(cap), z64(cap), z64(cap), z64(cap + 1), z32(cap), z32(cap)
    nr, err = gpu_ctx.fastq_index_device(text, text.numel(), line, ns, nl, ss, sl, qs, offs, cap)
    assert (nr, err) == (n, 0)
    total = int(offs[n])
    reads = torch.zeros(total, dtype=torch.uint8, device=dev)
    quals = torch.zeros(total, dtype=torch.uint8, device=dev)
    gpu_ctx.fastq_gather_device(text, ss, offs, n, reads)
    gpu_ctx.fastq_gather_device(text, qs, offs, n, quals)
    ends = torch.zeros((28, 2 * n), dtype=torch.int32, device=dev)
    lens, qsum = z32(n), z32(n)
    qt = torch.zeros((n, 224), dtype=torch.uint8, device=dev)
    gpu_ctx.pack_ends_device(reads, quals, offs[:n + 1], n, ends, lens, qt, qsum)
    scan = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    win = torch.zeros((n, 2), dtype=torch.int64, device=dev)
    gpu_ctx.scan_device(ends, lens, n, gpu_ctx.scan_config(2), scan, win, qt, qsum)
    bc = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    gpu_ctx.bc_match_device(win, bc, n, max_ed=1)
    capw = 2 * total + text.numel() + 320 * n
    out_p, out_f = torch.empty(capw, dtype=torch.uint8, device=dev), torch.empty(capw, dtype=torch.uint8, device=dev)
    rec_off, is_p = z64(n + 1), torch.zeros(n, dtype=torch.uint8, device=dev)
    bp, bf, n_p = gpu_ctx.fastq_write_device(text, line, reads, quals, offs[:n + 1], None, None, scan, bc, None, n, 1, out_p, out_f,
                                             rec_off, is_p)
    assert n_p == int(is_p.sum()) and 0.9 * n < n_p < n
    # the in-place entry points (bases and qualities read where the text has them, no gathers) give the same planes, ends and streams
    ends2, lens2 = torch.zeros_like(ends), z32(n)
    bstart, qstart = z64(n), z64(n)
    gpu_ctx.frag_text_starts_device(ss, qs, offs, None, None, n, bstart, qstart)
    assert bool((bstart == ss[:n]).all()) and bool((qstart == qs[:n]).all())
    gpu_ctx.pack_ends_text_device(text, bstart, offs[:n + 1], n, ends2, lens2)
    assert bool((ends2 == ends).all()) and bool((lens2 == lens).all())
    pw = gpu_ctx.read_planes_words(total, n)
    pl1, pl2 = z32(pw), z32(pw)
    gpu_ctx.pack_reads_device(reads, offs[:n + 1], n, total, pl1)
    gpu_ctx.pack_reads_text_device(text, ss, offs[:n + 1], n, total, pl2)
    assert bool((pl1 == pl2).all())
    out_p2, out_f2 = torch.empty(capw, dtype=torch.uint8, device=dev), torch.empty(capw, dtype=torch.uint8, device=dev)
    rec_off2, is_p2 = z64(n + 1), torch.zeros(n, dtype=torch.uint8, device=dev)
    tot2 = gpu_ctx.fastq_write_device(text, line, bstart, qstart, offs[:n + 1], None, None, scan, bc, None, n, 1, out_p2, out_f2, rec_off2,
                                      is_p2, in_text=True)
    assert tot2 == (bp, bf, n_p) and bool((out_p2[:bp] == out_p[:bp]).all()) and bool((out_f2[:bf] == out_f[:bf]).all())
    assert bool((rec_off2 == rec_off).all()) and bool((is_p2 == is_p).all())
    seq_total, hist_b, hist_q = 0, torch.zeros(256, dtype=torch.int64, device=dev), torch.zeros(256, dtype=torch.int64, device=dev)
    for buf, nbytes, want in ((out_p, bp, n_p), (out_f, bf, n - n_p)):
        k, e = gpu_ctx.fastq_index_device(buf, nbytes, line, ns, nl, ss, sl, qs, offs, cap)
        assert (k, e) == (want, 0)
        t = int(offs[k])
        sb, sq = torch.zeros(max(t, 1), dtype=torch.uint8, device=dev), torch.zeros(max(t, 1), dtype=torch.uint8, device=dev)
        gpu_ctx.fastq_gather_device(buf, ss, offs, k, sb)
        gpu_ctx.fastq_gather_device(buf, qs, offs, k, sq)
        seq_total += t
        hist_b += torch.bincount(sb[:t].to(torch.int64), minlength=256)
        hist_q += torch.bincount(sq[:t].to(torch.int64), minlength=256)
    assert seq_total == total
    h_in = torch.bincount(reads.to(torch.int64), minlength=256)
    at = lambda h: int(h[ord("A")] + h[ord("T")])  # noqa: E731
    gc = lambda h: int(h[ord("G")] + h[ord("C")])  # noqa: E731
    assert at(hist_b) == at(h_in) and gc(hist_b) == gc(h_in) and int(hist_b[ord("N")]) == int(h_in[ord("N")])
    assert bool((hist_q == torch.bincount(quals.to(torch.int64), minlength=256)).all())


def test_native_chunk_workers_equal_the_chained_entry_points(pkg, synth, sor, gpu_ctx):
    """smi_scanfastq_pass2_chunk / _pass1_chunk (one native call per chunk, host text in) == the same entry points chained
    from Python (ReadScanner), which the tests above compare with the oracle"""
    import torch

    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    used, reads = _reads(synth, 300, 981)
    chim = synth.make_chimeras(reads, 380, seed=984)
    seqs, quals = [c[0] for c in chim] + ["ACGT" * 30], [c[1] for c in chim] + ["5" * 120]
    keys = np.sort(used.numpy().astype(np.uint64))
    ranks = (np.arange(keys.size) % 50 + 1).astype(np.int32)
    text = _fastq(seqs, quals, qh=lambda i: "x" if i % 7 == 0 else "")
    gpu_ctx.set_barcode_set(keys, mode=0)
    for kw in (dict(), dict(trim_fastq=True), dict(split_chimeras=False)):
        rs = scanfastq.ReadScanner(gpu_ctx, max_ed=1, split_chimeras=kw.get("split_chimeras", True))
        exp_p, exp_f, info = rs.pass2_write_chunk(text, rank_keys=keys, rank_values=ranks, first_read_id=500, trim_fastq=kw.get("trim_fastq", False))
        got_p, got_f, ginfo = gpu_ctx.scanfastq_pass2_chunk(text, first_read_id=500, rank_keys=keys, rank_values=ranks, want_results=True, **kw)
        assert got_p == exp_p and got_f == exp_f
        assert (ginfo["n_records_in"], ginfo["n_records_out"], ginfo["n_passed"]) == (len(seqs), info["n_records"], info["n_passed"])
        assert ginfo["bc"].tobytes() == info["bc"].tobytes() and ginfo["scan"].tobytes() == info["scan"].tobytes()
    # a second, smaller chunk on the same context reuses the arena
    small = _fastq(seqs[:20], quals[:20])
    a, b, i2 = gpu_ctx.scanfastq_pass2_chunk(small)
    e_p, e_f, _ = scanfastq.ReadScanner(gpu_ctx, max_ed=1).pass2_write_chunk(small)
    assert a == e_p and b == e_f and i2["n_records_in"] == 20
    with pytest.raises(pkg.SmiError):
        gpu_ctx.scanfastq_pass2_chunk(b"@r1\nACGT\n-\nIIII\n")  # third line must start with '+': FastqReader throws
    assert gpu_ctx.scanfastq_pass2_chunk(b"")[2]["n_records_in"] == 0
    # pass 1
    wl = synth.make_whitelist(30_000, seed=985)
    wkeys = np.sort(np.unique(np.concatenate([wl.numpy().astype(np.uint64), keys])))
    gpu_ctx.set_barcode_set(wkeys, mode=1)
    h1 = torch.zeros(wkeys.size, dtype=torch.int32, device="cuda")
    h2 = torch.zeros(wkeys.size, dtype=torch.int32, device="cuda")
    n1 = scanfastq.ReadScanner(gpu_ctx).pass1_chunk(text, h1)
    n2 = gpu_ctx.scanfastq_pass1_chunk(text, h2)
    assert n1 == n2 == len(seqs) and bool((h1 == h2).all()) and int(h1.sum()) > 30


def test_chunk_worker_device_text_in_place_and_device_output(pkg, synth, sor, gpu_ctx):
    """the chunk as ONE call that never touches the host: a uint8 device tensor in (read where it lies, no copy into the arena), `passed` /
    `failed` left on the device (device_output) == the host-text call; also with --compress (the members stay on the device), for a text
    without a final newline, and on a context whose arena is reused by chunks of different sizes"""
    import gzip

    import torch

    used, reads = _reads(synth, 260, 4411)
    chim = synth.make_chimeras(reads, 300, seed=4412)
    seqs, quals = [c[0] for c in chim], [c[1] for c in chim]
    keys = np.sort(used.numpy().astype(np.uint64))
    ranks = (np.arange(keys.size) % 50 + 1).astype(np.int32)
    gpu_ctx.set_barcode_set(keys, mode=0)
    for text in (_fastq(seqs, quals), _fastq(seqs[:41], quals[:41])[:-1], _fastq(seqs, quals, eol="\r\n")):
        exp_p, exp_f, einfo = gpu_ctx.scanfastq_pass2_chunk(text, first_read_id=7, rank_keys=keys, rank_values=ranks)
        d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        before = d_text.clone()
        for copy in (True, False):
            got_p, got_f, info = gpu_ctx.scanfastq_pass2_chunk(d_text, first_read_id=7, rank_keys=keys, rank_values=ranks, device_output=True, copy=copy)
            if not copy:   # address + size inside the context's arena
                assert isinstance(got_p, pkg.DeviceSpan) and len(got_p) == len(exp_p) and len(got_f) == len(exp_f)
                got_p, got_f = got_p.tensor(), got_f.tensor()
            assert got_p.is_cuda and got_f.is_cuda and got_p.dtype == torch.uint8
            assert bytes(got_p.cpu().numpy()) == exp_p and bytes(got_f.cpu().numpy()) == exp_f
            assert {k: info[k] for k in ("n_records_in", "n_records_out", "n_passed")} == {k: einfo[k] for k in ("n_records_in", "n_records_out", "n_passed")}
        assert torch.equal(d_text, before)   # read in place, never written
        # host text in, device text out
        got_p, got_f, _ = gpu_ctx.scanfastq_pass2_chunk(text, first_read_id=7, rank_keys=keys, rank_values=ranks, device_output=True)
        assert bytes(got_p.cpu().numpy()) == exp_p and bytes(got_f.cpu().numpy()) == exp_f
        z_p, z_f, zinfo = gpu_ctx.scanfastq_pass2_chunk(d_text, first_read_id=7, rank_keys=keys, rank_values=ranks, device_output=True, compress=True)
        assert gzip.decompress(bytes(z_p.cpu().numpy())) == exp_p and gzip.decompress(bytes(z_f.cpu().numpy())) == exp_f
        assert zinfo["passed_text_bytes"] == len(exp_p) and zinfo["failed_text_bytes"] == len(exp_f)
    # pass 1 reads a device text in place as well
    gpu_ctx.set_barcode_set(keys, mode=1)
    text = _fastq(seqs, quals)
    d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    h1 = torch.zeros(keys.size, dtype=torch.int32, device="cuda")
    h2 = torch.zeros(keys.size, dtype=torch.int32, device="cuda")
    assert gpu_ctx.scanfastq_pass1_chunk(text, h1) == gpu_ctx.scanfastq_pass1_chunk(d_text, h2) == len(seqs)
    assert bool((h1 == h2).all()) and int(h1.sum()) > 30
    with pytest.raises(pkg.SmiError, match="want_results"):
        gpu_ctx.scanfastq_pass2_chunk(text, device_output=True, want_results=True)
    with pytest.raises(pkg.SmiError, match="text worker"):
        gpu_ctx.scanfastq_pass2_chunk(text, device_output=True, packed=True)
    with pytest.raises(pkg.SmiError):   # a malformed text in device memory fails as the host text does
        gpu_ctx.scanfastq_pass2_chunk(torch.frombuffer(bytearray(b"@r1\nACGT\n-\nIIII\n"), dtype=torch.uint8).cuda(), device_output=True)


def test_native_chunk_worker_5p_and_ed2(pkg, synth, sor, gpu_ctx):
    """the native worker in the other configurations: 5' barcoding without polyA requirement, and ed <= 2 (K-BC2)"""
    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    wl = synth.make_whitelist(20_000, seed=991)
    used = synth.pick_used(wl, 120, seed=992)
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    r5 = synth.gen_reads_5p(150, used, seed=993)
    text5 = _fastq(*zip(*(synth.materialize(r5, i) for i in range(150))))
    exp = scanfastq.ReadScanner(gpu_ctx, max_ed=1, five_prime=True, dont_search_polya=True).pass2_write_chunk(text5)
    got = gpu_ctx.scanfastq_pass2_chunk(text5, five_prime=True, dont_search_polya=True)
    assert got[0] == exp[0] and got[1] == exp[1] and got[2]["n_passed"] == exp[2]["n_passed"] > 80
    r3 = synth.gen_reads(200, used, seed=994, err=0.09)
    text3 = _fastq(*zip(*(synth.materialize(r3, i) for i in range(200))))
    for ed in (0, 2):
        exp = scanfastq.ReadScanner(gpu_ctx, max_ed=ed).pass2_write_chunk(text3)
        got = gpu_ctx.scanfastq_pass2_chunk(text3, max_ed=ed)
        assert got[0] == exp[0] and got[1] == exp[1]
    n_bc = {ed: gpu_ctx.scanfastq_pass2_chunk(text3, max_ed=ed)[0].count(b" cellBC=") for ed in (0, 1, 2)}
    assert n_bc[0] < n_bc[1] < n_bc[2]


def test_tiny_reads_long_names_and_headers(pkg, synth, sor, gpu_ctx):
    """the edges of the byte kernels: reads of 1 .. 70 bases (K-PACKR's partial last word with and without the wide load, K-WRITE records
    whose runs have no 16-byte aligned inside), names of 1 .. 300 characters with and without a blank, quality headers of 0 .. 200
    characters (several turns of K-WRITE's loose-byte loop), mixed with ordinary reads -- records equal the oracle's byte for byte, through
    the chained entry points and through the native chunk worker"""
    import random

    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    rng = random.Random(77)
    used, reads = _reads(synth, 40, 951)
    base = [synth.materialize(reads, i) for i in range(40)]
    seqs, quals, names, qhs = [], [], [], []
    lengths = list(range(1, 71)) + [199, 200, 201, 224, 225, 447, 448, 449]
    for j, n in enumerate(lengths):
        seqs.append("".join(rng.choice("ACGTN") for _ in range(n)))
        quals.append("".join(chr(33 + rng.randrange(40)) for _ in range(n)))
    for s, q in base:
        seqs.append(s)
        quals.append(q)
    for i in range(len(seqs)):
        ln = rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 120, 300])
        tok = "".join(rng.choice("abcdefghijklmnopqrstuvwxyz0123456789-") for _ in range(ln))
        names.append(tok if i % 3 == 0 else tok + " " + "x" * rng.choice([0, 1, 40, 150]))
        qhs.append("" if i % 2 else "h" * rng.choice([1, 30, 200]))
    text = "".join(f"@{nm}\n{s}\n+{h}\n{q}\n" for nm, s, h, q in zip(names, seqs, qhs, quals)).encode()
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    bset = sor.BarcodeSet(used.numpy())
    # oracle records (no chimera splitting for the tiny reads: they are below the splitter's minimum length anyway)
    passed, failed, rid = [], [], 1
    for nm, s, h, q in zip(names, seqs, qhs, quals):
        rc, splits, multi, _, raw = sor.chimera_split(s)
        assert rc == 0
        cuts = [0] + [p for _, p in splits] + [len(s)]
        for k in range(len(cuts) - 1):
            fs, fq = s[cuts[k]:cuts[k + 1]], q[cuts[k]:cuts[k + 1]]
            fname = sor.chimera_fragment_name(nm, raw, k) if splits else nm
            rc, sc = sor.scan_read_3p(fs, fq, "CTTCCGATCT")
            assert rc == 0
            a = None
            if sc["adapter_found"] and not multi:
                stranded = fs.encode().translate(COMP)[::-1] if sc["reverse"] else fs.encode()
                rc2, a_ = sor.assign_barcode(bset, stranded, int(sc["adapter_end"]), max_ed=1)
                if rc2 == 1:
                    a = a_
            rec, ok = sor.fastq_record(fname, h, fs, fq, sc, a, rank=0, read_id=rid, trim_fastq=False, force_failed=multi)
            assert rec is not None
            if ok:
                passed.append(rec)
                rid += 1
            else:
                failed.append(rec)
    exp_p, exp_f = b"".join(passed), b"".join(failed)
    rs = scanfastq.ReadScanner(gpu_ctx, max_ed=1)
    got_p, got_f, info = rs.pass2_write_chunk(text)
    assert got_f == exp_f
    assert got_p == exp_p and info["n_passed"] == len(passed) > 20
    nat_p, nat_f, ninfo = gpu_ctx.scanfastq_pass2_chunk(text, max_ed=1)
    assert bytes(nat_p) == exp_p and bytes(nat_f) == exp_f


def test_long_reads(pkg, synth, sor, gpu_ctx):
    """reads of up to ~30 kb (several turns of every per-read loop: K-PACKR words, K-CHIM segments of 2048 / 4096 positions, K-WRITE runs)
    equal the oracle's records, chained entry points and native chunk worker"""
    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    wl = synth.make_whitelist(20_000, seed=971)
    used = synth.pick_used(wl, 50, seed=972)
    reads = synth.gen_reads(24, used, seed=973, n_rate=0.001, max_mid=30_000)
    sq = [synth.materialize(reads, i) for i in range(24)]
    seqs, quals = [s for s, _ in sq], [q for _, q in sq]
    assert max(len(s) for s in seqs) > 15_000
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    rs = scanfastq.ReadScanner(gpu_ctx, max_ed=1)
    text = _fastq(seqs, quals)
    got_p, got_f, info = rs.pass2_write_chunk(text)
    exp_p, exp_f, n_p = _oracle_records(sor, sor.BarcodeSet(used.numpy()), seqs, quals, 1, {}, 1)
    assert got_p == exp_p and got_f == exp_f and info["n_passed"] == n_p > 10
    nat_p, nat_f, _ = gpu_ctx.scanfastq_pass2_chunk(text, max_ed=1)
    assert bytes(nat_p) == exp_p and bytes(nat_f) == exp_f


def test_very_long_reads(pkg, synth, sor, gpu_ctx):
    """ultra-long reads (a few hundred kb to over a megabase, as a genomic or concatemer-rich run can hold) among ordinary ones: text worker and packed worker
    write the oracle's records; nothing in the per-read loops (32-bit plane offsets, 2048-base segments, the writer's runs) is bound to cDNA lengths"""
    wl = synth.make_whitelist(20_000, seed=981)
    used = synth.pick_used(wl, 50, seed=982)
    reads = synth.gen_reads(10, used, seed=983, n_rate=0.001, max_mid=1_400_000)
    sq = [synth.materialize(reads, i) for i in range(10)]
    seqs, quals = [s_ for s_, _ in sq], [q for _, q in sq]
    assert max(len(s_) for s_ in seqs) > 700_000 and sum(len(s_) > 100_000 for s_ in seqs) >= 3
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    text = _fastq(seqs, quals)
    exp_p, exp_f, n_p = _oracle_records(sor, sor.BarcodeSet(used.numpy()), seqs, quals, 1, {}, 1)
    nat_p, nat_f, info = gpu_ctx.scanfastq_pass2_chunk(text, max_ed=1)
    assert bytes(nat_p) == exp_p and bytes(nat_f) == exp_f and info["n_passed"] == n_p
    pk_p, pk_f, _ = gpu_ctx.scanfastq_pass2_chunk(text, max_ed=1, packed=True, n_threads=3)
    assert bytes(pk_p) == exp_p and bytes(pk_f) == exp_f


def test_worker_lanes_share_one_barcode_set(pkg, synth, sor, gpu_ctx):
    """smi_ctx_create_lane / smi_ctx_lane_refresh: two lanes driven from two host threads at once give the records the owner gives (which
    the tests above compare with the oracle), before and after the owner loads another set; K-BC1 takes its table path on the lanes too
    (the neighbourhood bitmap and table belong to the owner, the byte buffer between its two kernels to the lane)"""
    import threading

    def texts(seed):
        used, reads = _reads(synth, 120, seed)
        chim = synth.make_chimeras(reads, 150, seed=seed + 5)
        return used, _fastq([c[0] for c in chim], [c[1] for c in chim])

    used_a, text_a = texts(981)
    used_b, text_b = texts(991)
    lanes = [gpu_ctx.lane(), gpu_ctx.lane()]
    for which, used in enumerate((used_a, used_b)):
        gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
        for ln in lanes:
            ln.refresh()
        want = [tuple(bytes(x) for x in gpu_ctx.scanfastq_pass2_chunk(t, max_ed=1)[:2]) for t in (text_a, text_b)]
        got = [None, None]

        def work(k):
            for _ in range(3):  # several chunks per lane, the two lanes interleaving on the GPU
                p, f, _info = lanes[k].scanfastq_pass2_chunk((text_a, text_b)[k], max_ed=1)
                got[k] = (bytes(p), bytes(f))

        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert got[0] == want[0] and got[1] == want[1]
        assert want[which][0].count(b"bc=") > 50  # the text whose molecules carry barcodes of the loaded list
    # ed <= 2 on a lane as well (filters and table of the owner)
    p2, f2, _ = lanes[0].scanfastq_pass2_chunk(text_b, max_ed=2)
    q2, g2, _ = gpu_ctx.scanfastq_pass2_chunk(text_b, max_ed=2)
    assert bytes(p2) == bytes(q2) and bytes(f2) == bytes(g2)
