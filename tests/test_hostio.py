"""The host side of the packed boundary (smi_hostio.hip; no GPU): smi_fastq_index_host against a line-based Python model of K-FQ's
rules, smi_pack_reads_host / smi_pack_quals_host against numpy models of K-PACKR / k_pack_quals, smi_fastq_write_host against the
oracle's restatement of FastqRecordExt.getRecordForWriting (the same expectation tests/test_write_gpu.py holds K-WRITE to), each in its
three SIMD forms (SMI_HOST_SIMD) and on one and several threads."""
import os
import random

import numpy as np
import pytest

COMP = bytes.maketrans(b"ACGTN", b"TGCAN")
FQ_BAD_SEQ_HEADER, FQ_BAD_QUAL_HEADER, FQ_LENGTH_MISMATCH, FQ_TRUNCATED = 1, 2, 4, 8


@pytest.fixture
def libmod(pkg):
    from sicelore_amd import lib as libmod

    libmod.load_library()
    return libmod


@pytest.fixture(params=[0, 1, 2])
def simd(request):
    old = os.environ.get("SMI_HOST_SIMD")
    os.environ["SMI_HOST_SIMD"] = str(request.param)
    yield request.param
    if old is None:
        del os.environ["SMI_HOST_SIMD"]
    else:
        os.environ["SMI_HOST_SIMD"] = old


def model_index(text):
    """K-FQ (smi_fastq.hip): lines end at LF, a last line without LF counts, a CR in front of the LF is stripped; record r = lines 4r..4r+3"""
    n = len(text)
    starts, p = [], 0
    while p < n:
        starts.append(p)
        q = text.find(b"\n", p)
        p = n if q < 0 else q + 1
    ends = []
    for k, s in enumerate(starts):
        e = starts[k + 1] - 1 if k + 1 < len(starts) else (n - 1 if text.endswith(b"\n") else n)
        if e > s and text[e - 1:e] == b"\r":
            e -= 1
        ends.append(e)
    err = FQ_TRUNCATED if len(starts) % 4 else 0
    recs = []
    for r in range(len(starts) // 4):
        l, e = starts[4 * r:4 * r + 4], ends[4 * r:4 * r + 4]
        if e[0] == l[0] or text[l[0]:l[0] + 1] != b"@":
            err |= FQ_BAD_SEQ_HEADER
        if e[2] == l[2] or text[l[2]:l[2] + 1] != b"+":
            err |= FQ_BAD_QUAL_HEADER
        if e[1] - l[1] != e[3] - l[3]:
            err |= FQ_LENGTH_MISMATCH
        recs.append((l[0] + 1, l[1], l[2] + 1, l[3], max(e[0] - l[0] - 1, 0), e[1] - l[1], max(e[2] - l[2] - 1, 0)))
    return recs, err


def check_index(libmod, text, n_threads):
    recs, offs, err = libmod.fastq_index_host(text, n_threads=n_threads)
    exp, exp_err = model_index(text)
    assert err == exp_err
    assert len(recs) == len(exp)
    got = list(zip(recs["name_start"].tolist(), recs["seq_start"].tolist(), recs["plus_start"].tolist(), recs["qual_start"].tolist(),
                   recs["name_len"].tolist(), recs["seq_len"].tolist(), recs["plus_len"].tolist()))
    assert got == exp
    assert offs.tolist() == np.concatenate([[0], np.cumsum([e[5] for e in exp])]).astype(np.uint64).tolist()
    return recs, offs


def test_index_small_cases(libmod):
    cases = [
        b"",
        b"@r1\nACGT\n+\nIIII\n",
        b"@r1\nACGT\n+\nIIII",                       # no final newline
        b"@r1 d\r\nACGT\r\n+x\r\nIIII\r\n@r2\r\nA\r\n+\r\nI\r\n",
        b"@r1\n\n+\n\n@r2\nAC\n+\nII\n",              # an empty read
        b"@r1\nACGT\n+\nIII\n",                       # length mismatch
        b"r1\nACGT\n+\nIIII\n",                       # bad '@'
        b"@r1\nACGT\n-\nIIII\n",                      # bad '+'
        b"@r1\nACGT\n+\nIIII\n@r2\nAC\n",             # truncated
        b"@r1\nACGT\n+\nIIII\n\n",                    # a stray empty line
        b"\n\n\n\n",
        b"@\nA\n+\nI\n",
        b"@r\n" + b"ACGT" * 100 + b"\n+\n" + b"@" * 400 + b"\n@s\nAC\n+\n@@\n",   # quality lines that start with '@'
    ]
    for text in cases:
        check_index(libmod, text, 1)
        check_index(libmod, text, 4)


def _big_text(n, seed, eol=b"\n", final_newline=True):
    rng = random.Random(seed)
    parts = []
    for i in range(n):
        ln = rng.choice([0, 1, 30, 31, 32, 33, 63, 64, 65, 200, 900, 1500, 2500])
        seq = bytes(rng.choice(b"ACGTN") for _ in range(ln)) if ln < 100 else bytes(rng.choices(b"ACGT", k=ln))
        qual = bytes(rng.choices(range(33, 74), k=ln))        # includes '@' (64) and '+' (43), also as first character
        parts.append(b"@r%d runid=%d" % (i, seed) + eol + seq + eol + (b"+" if i % 3 else b"+r%d" % i) + eol + qual + eol)
    text = b"".join(parts)
    return text if final_newline else text[:-len(eol)]


@pytest.mark.parametrize("eol,final", [(b"\n", True), (b"\r\n", True), (b"\n", False)])
def test_index_large_parallel(libmod, eol, final):
    text = _big_text(6000, 3, eol, final)
    assert len(text) > 5_000_000                      # enough for the speculative split to use several threads
    a, ao = check_index(libmod, text, 1)
    b, bo = check_index(libmod, text, 5)
    assert a.tobytes() == b.tobytes() and ao.tobytes() == bo.tobytes()
    # a malformed record deep inside: the threaded form reports exactly what the sequential one does
    k = text.index(b"@r4000 ")
    bad = text[:k] + b"#" + text[k + 1:]
    check_index(libmod, bad, 5)
    cut = text[:len(text) * 2 // 3]
    check_index(libmod, cut, 5)


def model_planes(seqs):
    """K-PACKR: planes[c][plane_start(off, r) + w] bit b = bit c of the IUPAC code of base 32 w + b (A 1 G 2 C 4 T 8, either case; else 15)"""
    n = len(seqs)
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    stride = (int(offs[-1]) >> 5) + 5 * (n + 1) + 8
    pl = np.zeros((4, stride), dtype=np.uint32)
    lut = np.full(256, 15, dtype=np.uint8)
    for ch, v in ((b"A", 1), (b"G", 2), (b"C", 4), (b"T", 8)):
        lut[ch[0]] = lut[ch.lower()[0]] = v
    for r, s in enumerate(seqs):
        codes = lut[np.frombuffer(s, dtype=np.uint8)]
        w0 = (int(offs[r]) >> 5) + 5 * r
        for c in range(4):
            bits = ((codes >> c) & 1).astype(np.uint8)
            padded = np.zeros((len(s) + 31) // 32 * 32, dtype=np.uint8)
            padded[:len(s)] = bits
            words = np.packbits(padded.reshape(-1, 32), axis=1, bitorder="little").view("<u4").reshape(-1)
            pl[c, w0:w0 + words.size] = words
    return pl.reshape(-1)


def test_pack_reads_equals_model(libmod, simd):
    rng = random.Random(11)
    seqs = [bytes(rng.choice(b"ACGTNacgtnRYKMxz*-") for _ in range(ln)) for ln in list(range(0, 140)) + [224, 255, 256, 257, 1000, 4097]]
    seqs += [bytes(rng.choices(b"ACGT", k=rng.randrange(300, 2000))) for _ in range(300)]
    text = b"".join(b"@r%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n" for i, s in enumerate(seqs))
    recs, offs, err = libmod.fastq_index_host(text, n_threads=2)
    assert err == 0 and len(recs) == len(seqs)
    exp = model_planes(seqs)
    for nt in (1, 3):
        got = libmod.pack_reads_host(text, recs, offs, n_threads=nt)
        assert got.size == exp.size and (got == exp).all()


def test_pack_quals_equals_model(libmod, simd):
    rng = random.Random(12)
    lens = list(range(0, 70)) + [223, 224, 225, 300, 1000, 5000]
    quals = [bytes(rng.choices(range(33, 127), k=ln)) for ln in lens]
    text = b"".join(b"@r%d\n" % i + b"A" * len(q) + b"\n+\n" + q + b"\n" for i, q in enumerate(quals))
    recs, offs, err = libmod.fastq_index_host(text)
    assert err == 0
    for five in (False, True):
        qt, qs = libmod.pack_quals_host(text, recs, five_prime=five, n_threads=2)
        for i, q in enumerate(quals):
            a = np.frombuffer(q, dtype=np.uint8)
            assert int(qs[i]) == int(a.sum()) - 33 * len(q)
            exp = np.full(224, 33, dtype=np.uint8)
            if five:
                exp[:min(224, len(q))] = a[:224]
            elif len(q):
                exp[224 - min(224, len(q)):] = a[-224:]
            assert (qt[i] == exp).all()


def _compact_planes(planes, pk):
    """the device's view of the fused packer's output: the segments back to back in each plane"""
    out = np.zeros((4, pk.total_words), dtype=np.uint32)
    for c in range(4):
        for k in range(pk.n_seg):
            h, d, w = int(pk.seg_host_word[k]), int(pk.seg_dev_word[k]), int(pk.seg_words[k])
            out[c, d:d + w] = planes[c * pk.stride + h:c * pk.stride + h + w]
    return out


@pytest.mark.parametrize("eol,final", [(b"\n", True), (b"\r\n", True), (b"\n", False), (b"\r\n", False)])
def test_index_pack_one_pass_equals_two_steps(libmod, simd, eol, final):
    """smi_fastq_index_pack_host: same records and offsets as smi_fastq_index_host, and every read's plane words (found through pstart in the
    compact layout) equal the K-PACKR model's; one thread and several (segments)"""
    text = _big_text(5000, 7, eol, final)
    recs0, offs0, err0 = libmod.fastq_index_host(text, n_threads=1)
    assert err0 == 0
    seqs = [text[int(r["seq_start"]):int(r["seq_start"]) + int(r["seq_len"])] for r in recs0]
    model = model_planes(seqs).reshape(4, -1)
    for nt in (1, 4):
        recs, offs, err, pstart, planes, pk = libmod.fastq_index_pack_host(text, n_threads=nt)
        assert err == 0 and pstart is not None and pk.n_seg == nt
        a, b = recs.copy(), recs0.copy()
        assert (a["reserved"] == 1).all()
        a["reserved"] = 0
        assert a.tobytes() == b.tobytes() and offs.tobytes() == offs0.tobytes()
        comp = _compact_planes(planes, pk)
        for r, s in enumerate(seqs):
            nw = (len(s) + 31) // 32 + 4
            w0 = (int(offs0[r]) >> 5) + 5 * r
            got = comp[:, int(pstart[r]):int(pstart[r]) + nw]
            assert (got == model[:, w0:w0 + nw]).all(), (r, len(s))


def test_index_pack_falls_back_on_anything_unusual(libmod):
    base = _big_text(4000, 9)
    k = base.index(b"@r2500 ")
    cases = [base[:k] + b"#" + base[k + 1:],          # a bad '@' deep inside
             base[:len(base) * 2 // 3],                # truncated
             b"@r1\nACGT\n+\nIII\n",                   # length mismatch
             b"@r1\nACGT\n-\nIIII\n", b"", b"@r\n\n+\n\n", b"@r\nAC\n+\nII"]
    for text in cases:
        for nt in (1, 4):
            recs, offs, err, pstart, planes, pk = libmod.fastq_index_pack_host(text, n_threads=nt)
            exp, exp_err = model_index(text)
            assert err == exp_err and len(recs) == len(exp)
    # tiny records (more than the one-pass packer's segments allow for): the two-step path, K-PACKR's layout
    tiny = b"@a\nA\n+\nI\n" * 50_000
    recs, offs, err, pstart, planes, pk = libmod.fastq_index_pack_host(tiny, n_threads=2)
    assert err == 0 and len(recs) == 50_000
    seqs = [b"A"] * 50_000
    if pstart is None:
        assert (planes[:libmod.read_planes_words(50_000, 50_000)] == model_planes(seqs)).all()


def test_hidden_line_end_in_a_quality_string_is_caught_later(libmod, sor):
    """the one-pass index steps over quality lines; a line end hidden inside one (placed so that the line still seems to end where it
    should) is reported by whoever reads the qualities: the writer (SMI_WR_QUAL_NEWLINE) and the pass-1 quality packer"""
    seq = "ACGT" * 60
    good = f"@r0 x\n{seq}\n+\n{'I' * 240}\n".encode()
    bad = f"@r1 x\n{seq}\n+\n{'I' * 100}\n{'I' * 139}\n".encode()      # 100 + LF + 139 = 240 bytes, then the LF the index expects
    text = good * 3 + bad + good * 2
    _, exp_err = model_index(text)
    assert exp_err != 0                                                 # the exact index (and K-FQ, and htsjdk) reject it
    recs, offs, err, pstart, planes, pk = libmod.fastq_index_pack_host(text, n_threads=1)
    assert err == 0 and len(recs) == 6 and (recs["reserved"] == 1).all()
    scan = np.zeros(6, dtype=libmod.SCAN_RESULT_DTYPE)
    bc = np.zeros(6, dtype=libmod.BC_RESULT_DTYPE)
    with pytest.raises(libmod.SmiError, match="error bits 8"):
        libmod.fastq_write_host(text, recs, offs, scan, bc)
    with pytest.raises(libmod.SmiError, match="line end inside"):
        libmod.pack_quals_host(text, recs)
    ok = good * 6
    recs, offs, err, pstart, planes, pk = libmod.fastq_index_pack_host(ok, n_threads=1)
    p, f, n_p = libmod.fastq_write_host(ok, recs, offs, scan, bc)
    assert n_p == 0 and f.count(b"_FAILED ") == 6


# ---- the writer against the oracle's records -----------------------------------------------------------------------------------------
def _scan_rec(libmod, sc):
    r = np.zeros(1, dtype=libmod.SCAN_RESULT_DTYPE)[0]
    r["flags"] = int(sc["flags"]) & 0xFFFFFFFF
    for k in ("adapter_start", "adapter_end", "polya_start", "polya_end", "scan_end", "tso_start", "adapter_nmis", "reverse", "pass1_ok", "tso_end"):
        r[k] = int(sc[k])
    r["found"] = int(sc["adapter_found"])
    return r


def _bc_rec(libmod, a):
    r = np.zeros(1, dtype=libmod.BC_RESULT_DTYPE)[0]
    if a is None:
        return r
    r["bc"], r["ed_sec"], r["found"], r["ed"] = int(a["bc"]) & 0xFFFFFFFF, int(a["ed_sec"]), 1, int(a["ed"])
    r["offset"], r["ins_minus_del"], r["n_matches"] = int(a["offset"]), int(a["ins_minus_del"]), int(a["n_matches"])
    return r


def _oracle_flow(libmod, sor, bset, names, qhs, seqs, quals, max_ed, rank_of, first_id, five_prime=False, trim=False, split=True):
    """the oracle's records and, beside them, the decisions a device would have sent back (same structs, filled from the oracle)"""
    passed, failed, rid = [], [], first_id
    scan, bc, rank, foffs, fsrc, chim = [], [], [], [0], [], []
    base = 0
    for i, (nm, qh, s, q) in enumerate(zip(names, qhs, seqs, quals)):
        splits, multi, raw = [], False, None
        ch = np.zeros(1, dtype=libmod.CHIMERA_RESULT_DTYPE)[0]
        if split:
            rc, splits, multi, nmatch, raw = sor.chimera_split(s, sor.chimera_params(22 if five_prime else 28)) if five_prime else sor.chimera_split(s)
            assert rc == 0
            ch["n_split"], ch["flags"], ch["n_matches"] = len(splits), 1 if multi else 0, nmatch
            for k, (reason, pos) in enumerate(splits):
                ch["pos"][k], ch["reason"][k] = pos, sor.SPLIT_REASONS.index(reason)
        chim.append(ch)
        cuts = [0] + [p for _, p in splits] + [len(s)]
        for k in range(len(cuts) - 1):
            fs, fq = s[cuts[k]:cuts[k + 1]], q[cuts[k]:cuts[k + 1]]
            fname = sor.chimera_fragment_name(nm, raw, k) if splits else nm
            rc, sc = (sor.scan_read_5p if five_prime else sor.scan_read_3p)(fs, fq, "CTTCCGATCT")
            assert rc == 0
            a = None
            if sc["adapter_found"] and not multi:
                stranded = fs.encode().translate(COMP)[::-1] if sc["reverse"] else fs.encode()
                rc2, a_ = sor.assign_barcode(bset, stranded, int(sc["adapter_end"]), max_ed=max_ed, five_prime=five_prime)
                if rc2 == 1:
                    a = a_
            rk = rank_of.get(int(a["bc"]) & 0xFFFFFFFF, 0) if a is not None else 0
            rec, ok = sor.fastq_record(fname, qh, fs, fq, sc, a, rank=rk, read_id=rid, five_prime=five_prime, trim_fastq=trim, force_failed=multi)
            assert rec is not None
            if ok:
                passed.append(rec)
                rid += 1
            else:
                failed.append(rec)
            scan.append(_scan_rec(libmod, sc))
            bc.append(_bc_rec(libmod, a))
            rank.append(rk)
            foffs.append(base + cuts[k + 1])
            fsrc.append(i << 2 | k)
        base += len(s)
    dec = dict(scan=np.array(scan, dtype=libmod.SCAN_RESULT_DTYPE), bc=np.array(bc, dtype=libmod.BC_RESULT_DTYPE),
               rank=np.array(rank, dtype=np.int32))
    if split:
        dec.update(frag_offsets=np.array(foffs, dtype=np.uint64), frag_src=np.array(fsrc, dtype=np.uint32),
                   chim=np.array(chim, dtype=libmod.CHIMERA_RESULT_DTYPE))
    return b"".join(passed), b"".join(failed), len(passed), dec


def _reads(synth, n, seed):
    wl = synth.make_whitelist(20_000, seed=seed)
    used = synth.pick_used(wl, 150, seed=seed + 1)
    return used, synth.gen_reads(n, used, seed=seed + 2, n_rate=0.002)


@pytest.mark.parametrize("trim", [False, True])
def test_write_host_equals_oracle_3p(libmod, synth, sor, simd, trim):
    used, reads = _reads(synth, 120, 2911)
    chim = synth.make_chimeras(reads, 160, seed=2914)
    seqs = [c[0] for c in chim] + ["ACGT" * 30, "A"]
    quals = [c[1] for c in chim] + ["5" * 120, "#"]
    names = [f"read{i} runid=x ch={i % 9}" if i % 4 else f"read{i}" for i in range(len(seqs))]
    qhs = [f"read{i} again" if i % 5 == 0 else "" for i in range(len(seqs))]
    keys = np.sort(used.numpy().astype(np.uint64))
    rank_of = {int(k): int(r) for k, r in zip(keys, np.arange(keys.size) * 7 % 1000 + 1)}
    exp_p, exp_f, n_p, dec = _oracle_flow(libmod, sor, sor.BarcodeSet(used.numpy()), names, qhs, seqs, quals, 1, rank_of, 35 ** 2, trim=trim)
    assert n_p > 100 and exp_p.count(b"sp1") > 5 and exp_p.count(b"_REV_") > 20 and exp_p.count(b"_rk=") > 50
    text = "".join(f"@{nm}\n{s}\n+{h}\n{q}\n" for nm, h, s, q in zip(names, qhs, seqs, quals)).encode()
    recs, offs, err = libmod.fastq_index_host(text)
    assert err == 0
    for nt in (1, 3):
        got_p, got_f, got_n = libmod.fastq_write_host(text, recs, offs, first_read_id=35 ** 2, trim_fastq=trim, n_threads=nt, **dec)
        assert got_p == exp_p
        assert got_f == exp_f
        assert got_n == n_p


def test_write_host_equals_oracle_5p_crlf(libmod, synth, sor, simd):
    wl = synth.make_whitelist(20_000, seed=2951)
    used = synth.pick_used(wl, 150, seed=2952)
    reads = synth.gen_reads_5p(120, used, seed=2953)
    seqs, quals = zip(*(synth.materialize(reads, i) for i in range(120)))
    names = [f"read{i} x" for i in range(120)]
    qhs = ["" for _ in range(120)]
    for trim in (False, True):
        exp_p, exp_f, n_p, dec = _oracle_flow(libmod, sor, sor.BarcodeSet(used.numpy()), names, qhs, seqs, quals, 1, {}, 1, five_prime=True,
                                              trim=trim, split=False)
        assert n_p > 60
        text = "".join(f"@{nm}\r\n{s}\r\n+\r\n{q}\r\n" for nm, s, q in zip(names, seqs, quals)).encode()
        recs, offs, err = libmod.fastq_index_host(text)
        dec.pop("rank")
        got = libmod.fastq_write_host(text, recs, offs, five_prime=True, trim_fastq=trim, n_threads=2, **dec)
        assert got == (exp_p, exp_f, n_p)


def test_write_host_reverse_complement_of_any_byte(libmod, sor, simd):
    """FastqRecordExt.REVERSE_COMPLEMENT maps IUPAC letters of either case and everything else to 0; runs of 1 .. 300 bases so that the
    64 / 32-byte blocks and their tails are all taken; qualities are mirrored"""
    rng = random.Random(5)
    names, seqs, quals, scan = [], [], [], []
    alphabet = b"ACGTNacgtnHRYMKSWBVDhrymkswbvd*-.xz0"
    for ln in list(range(46, 200)) + [255, 256, 257, 300, 1024]:
        s = bytes(rng.choice(alphabet) for _ in range(ln)).decode()
        q = bytes(rng.choices(range(33, 127), k=ln)).decode()
        sc = {k: 0 for k in ("adapter_start", "polya_start", "polya_end", "scan_end", "tso_start", "adapter_nmis", "pass1_ok", "tso_end")}
        sc.update(flags=1 << 9, adapter_found=1, adapter_end=42 + rng.randrange(0, ln - 45), reverse=1)   # PASSED_REV, window inside the read
        names.append(f"r{ln}")
        seqs.append(s)
        quals.append(q)
        scan.append(sc)
    exp = []
    for i, (nm, s, q, sc) in enumerate(zip(names, seqs, quals, scan)):
        full = np.zeros(1, dtype=sor.SCAN_RESULT_DTYPE)[0]
        for k, v in sc.items():
            full[k] = v
        rec, ok = sor.fastq_record(nm, "", s, q, full, None, rank=0, read_id=1 + i)
        assert ok and rec is not None
        exp.append(rec)
    text = "".join(f"@{nm}\n{s}\n+\n{q}\n" for nm, s, q in zip(names, seqs, quals)).encode()
    recs, offs, err = libmod.fastq_index_host(text)
    assert err == 0
    sc_arr = np.array([_scan_rec(libmod, s) for s in scan], dtype=libmod.SCAN_RESULT_DTYPE)
    bc_arr = np.zeros(len(scan), dtype=libmod.BC_RESULT_DTYPE)
    got_p, got_f, n_p = libmod.fastq_write_host(text, recs, offs, sc_arr, bc_arr, n_threads=2)
    assert got_f == b"" and n_p == len(scan)
    assert got_p == b"".join(exp)


def test_exports_and_errors(libmod):
    with pytest.raises(libmod.SmiError):
        libmod.fastq_index_host(b"@a\nA\n+\nI\n" * 10, cap_records=5)       # more records than the caller's buffers hold
