"""config.xml's knobs at run time (smi_run_knobs, round 6): every knob group through the kernels against the ORACLE CALLED WITH THE
SAME PARAMETERS -- the read scan (3' and 5'), the pass-1 filter, the chimera splitter, and the chunk workers of a context that carries the knobs
(text and packed form).  Reference: Jar/config.xml:21-61,95-183,264; ParametersReadScannerApp.java:L84-130; Parser.java:L99,L134-136;
ChimeraFindernew.java:L74-81; UsedCellBCListGenerator$Worker.java:L198-202."""
import numpy as np
import pytest
import torch

from test_scan_gpu import _ascii_batch, _compare, _t_rich_reads

pytestmark = pytest.mark.gpu

COMP = bytes.maketrans(b"ACGTN", b"TGCAN")
# another adapter (22 bases; `sequence` = its last 10) and another complete TSO (27 bases) than the shipped ones
AD_OTHER = "GTCAGATGTGTATAAGAGACAG"          # (the Nextera read-1 tail: a sequence a user could plausibly put there)
TSO_OTHER = "AAGCAGTGGTATCAACGCAGAGTGAAT"


def _scan_with(pkg, ctx, ra, qa, offs, cfg, five_prime=False):
    n = offs.size - 1
    d_reads, d_quals = torch.from_numpy(ra.copy()).cuda(), torch.from_numpy(qa.copy()).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_ends = torch.zeros((28, 2 * n), dtype=torch.int32, device="cuda")
    d_len = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_qtail = torch.zeros((n, 224), dtype=torch.uint8, device="cuda")
    d_qsum = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.pack_ends_device(d_reads, d_quals, d_offs, n, d_ends, d_len, d_qtail, d_qsum, five_prime=five_prime)
    d_out = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    ctx.scan_device(d_ends, d_len, n, cfg, d_out, None, d_qtail, d_qsum)
    torch.cuda.synchronize()
    return d_out.cpu().numpy().view(pkg.SCAN_RESULT_DTYPE).reshape(-1)


def _oracle_scan_params(sor, k):
    par = sor.default_scan_params()
    par["min_read_length"], par["polya_len"], par["polya_frac"], par["window_polya"] = k.min_read_length, k.polya_len, k.polya_frac, k.window_polya
    par["min_adapter_3p_matches"], par["min_mean_bc_qv"], par["min_mean_read_qv"] = k.min_adapter_3p_matches, k.min_mean_bc_qv, k.min_mean_read_qv
    sor.set_tso_params(par, k.tso_scan.decode(), k.tso_scan_window, k.tso_scan_max_mm, k.tso_scan_min_consec, k.tso_scan_min_two_best)
    return par


KNOB_SETS_3P = [
    dict(adapter3p_max_mm=1), dict(adapter3p_max_mm=2), dict(adapter3p_max_mm=5), dict(adapter3p_max_mm=0),
    dict(min_read_length=180), dict(min_read_length=600),
    dict(min_mean_bc_qv=10, min_mean_read_qv=10), dict(min_mean_bc_qv=14, min_mean_read_qv=5), dict(min_adapter_3p_matches=5),
    dict(min_adapter_3p_matches=10, adapter3p_max_mm=4),
    dict(adapter3p=AD_OTHER[-10:], adapter3p_complete=AD_OTHER),
    dict(adapter3p=AD_OTHER[-10:], adapter3p_complete=AD_OTHER, adapter3p_max_mm=4, min_read_length=300, polya_len=12, polya_frac=0.8, window_polya=120),
    # the TSO of the read scan (tso_for3pBarcoding: sequence, maxNeedlemanMismatches, the two rescue rules, windowForTSOsearch): the generic kernels
    dict(tso_scan_max_mm=3), dict(tso_scan_max_mm=7, tso_scan_min_consec=6), dict(tso_scan_min_consec=11, tso_scan_min_two_best=9), dict(tso_scan_window=60),
    dict(tso_scan_window=112, tso_scan_max_mm=4), dict(tso_scan=TSO_OTHER[-14:] + "GG", tso_complete=TSO_OTHER),
    dict(tso_scan="ACGTTGCAAGGCTTAC", tso_scan_window=40, tso_scan_max_mm=6, tso_scan_min_consec=5, tso_scan_min_two_best=10, min_read_length=180),
]


@pytest.mark.parametrize("over", KNOB_SETS_3P, ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items())[:60])
def test_scan_knobs_3p_equal_oracle(pkg, synth, sor, gpu_ctx, over):
    """minReadLength, minMeanBCqv / minMeanReadqv / minAdapter3pMatches, maxNeedlemanMismatches and the adapter sequences of 3' barcoding: K-SCAN
    with the configuration the chunk workers derive from the knobs (smi_scan_config_from_knobs), pass 2 and pass 1, against the oracle"""
    lib = pkg.lib if hasattr(pkg, "lib") else __import__("importlib").import_module("sicelore_amd.lib")
    k = lib.run_knobs(**over)
    par = _oracle_scan_params(sor, k)
    wl = synth.make_whitelist(50_000, seed=1201)
    used = synth.pick_used(wl, 300, seed=1202)
    n = 3000
    reads = synth.gen_reads(n, used, seed=1203, n_rate=0.003, adapter_complete=k.adapter3p_complete.decode(), tso_complete=k.tso_complete.decode())
    ra, qa, offs = _ascii_batch(synth, reads, n, short_every=40)
    found = {}
    for pass_no in (2, 1):
        cfg = gpu_ctx.scan_config(pass_no, knobs=k)
        got = _scan_with(pkg, gpu_ctx, ra, qa, offs, cfg)
        ad = (k.adapter3p if pass_no == 2 else k.adapter3p_complete).decode()
        st, exp = sor.scan_batch_3p(ra, qa, offs, ad, max_mm=k.adapter3p_max_mm, params=par, n_threads=8)
        found[pass_no] = _compare(got, st, exp, pass1=True)
        if pass_no == 1 and k.min_mean_bc_qv <= 10:
            assert exp["pass1_ok"].sum() > 0
    assert found[2] > (0.05 if k.adapter3p_max_mm == 0 else 0.3) * n
    ra, qa, offs = _t_rich_reads(3000, seed=1210)
    got = _scan_with(pkg, gpu_ctx, ra, qa, offs, gpu_ctx.scan_config(2, knobs=k))
    st, exp = sor.scan_batch_3p(ra, qa, offs, k.adapter3p.decode(), max_mm=k.adapter3p_max_mm, params=par, n_threads=8)
    _compare(got, st, exp, pass1=True)


def test_scan_knobs_change_results(pkg, synth, sor, gpu_ctx):
    """(the knobs are not decoration: the shipped values and other values disagree on the same reads)"""
    lib = __import__("importlib").import_module("sicelore_amd.lib")
    wl = synth.make_whitelist(50_000, seed=1221)
    used = synth.pick_used(wl, 300, seed=1222)
    n = 3000
    reads = synth.gen_reads(n, used, seed=1223, n_rate=0.003)
    ra, qa, offs = _ascii_batch(synth, reads, n, short_every=40)
    base = _scan_with(pkg, gpu_ctx, ra, qa, offs, gpu_ctx.scan_config(1))
    assert (base == _scan_with(pkg, gpu_ctx, ra, qa, offs, gpu_ctx.scan_config(1, knobs=lib.run_knobs()))).all()
    for over in (dict(adapter3p_max_mm=1), dict(min_read_length=600), dict(min_mean_bc_qv=14), dict(min_adapter_3p_matches=12), dict(tso_scan_max_mm=2),
                 dict(tso_scan_window=40), dict(tso_scan="AACGCAGAGTGAATGG")):
        other = _scan_with(pkg, gpu_ctx, ra, qa, offs, gpu_ctx.scan_config(1, knobs=lib.run_knobs(**over)))
        assert (other != base).any(), over


KNOB_SETS_5P = [dict(adapter5p_window=80), dict(adapter5p_window=140, adapter5p_max_mm=2), dict(adapter5p_max_mm=5), dict(adapter5p_max_mm=0, min_read_length=300),
                dict(adapter5p=AD_OTHER[-10:], adapter5p_complete=AD_OTHER, adapter5p_window=100)]


@pytest.mark.parametrize("dont", [True, False])
@pytest.mark.parametrize("over", KNOB_SETS_5P, ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items())[:60])
def test_scan_knobs_5p_equal_oracle(pkg, synth, sor, gpu_ctx, over, dont):
    """fiveprimeadapter_for5pBarcoding: sequence, maxNeedlemanMismatches (+ 1: Parser.java:L99) and AdapterSearchWindow, with and without --noPolyARequired"""
    lib = __import__("importlib").import_module("sicelore_amd.lib")
    k = lib.run_knobs(**over)
    par = _oracle_scan_params(sor, k)
    wl = synth.make_whitelist(50_000, seed=1231)
    used = synth.pick_used(wl, 300, seed=1232)
    n = 1200
    reads = synth.gen_reads_5p(n, used, seed=1233, n_rate=0.003)
    ra, qa, offs = _ascii_batch(synth, reads, n, short_every=40)
    if k.adapter5p_complete.decode() != AD_OTHER:
        pass
    else:   # plant the other adapter where the shipped one sits (read start of the forward reads)
        ra = ra.copy()
        for i in range(0, n, 2):
            b = int(offs[i])
            if offs[i + 1] - offs[i] > 60:
                ra[b:b + 22] = np.frombuffer(AD_OTHER.encode(), dtype=np.uint8)
    cfg = gpu_ctx.scan_config(2, knobs=k, five_prime=True, dont_search_polya=dont)
    got = _scan_with(pkg, gpu_ctx, ra, qa, offs, cfg, five_prime=True)
    n_found = 0
    for i in range(n):
        seq = bytes(ra[int(offs[i]):int(offs[i + 1])]).decode()
        qual = bytes(qa[int(offs[i]):int(offs[i + 1])]).decode()
        rc, e = sor.scan_read_5p(seq, qual, k.adapter5p.decode(), max_mm=k.adapter5p_max_mm + 1, params=par, window=k.adapter5p_window, dont_search_polya=dont)
        if rc != 0:
            assert got["reserved"][i] == 1
            continue
        assert int(got["flags"][i]) == int(e["flags"]), (i, hex(int(got["flags"][i])), hex(int(e["flags"])))
        assert got["found"][i] == e["adapter_found"]
        assert (got["polya_start"][i], got["polya_end"][i]) == (e["polya_start"], e["polya_end"]), i
        if e["adapter_found"]:
            n_found += 1
            for f in ("adapter_start", "adapter_end", "scan_end", "adapter_nmis", "reverse"):
                assert int(got[f][i]) == int(e[f]), (i, f)
    assert n_found > (0.02 if k.adapter5p_max_mm == 0 else 0.2) * n


def _chimera_gpu(pkg, ctx, seqs, cfg):
    n = len(seqs)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    total = int(offs[-1])
    ra = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    d_reads = torch.from_numpy(ra.copy()).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_planes = torch.full((ctx.read_planes_words(total, n),), -1, dtype=torch.int32, device="cuda")
    ctx.pack_reads_device(d_reads, d_offs, n, total, d_planes)
    d_out = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    ctx.chimera_device(d_planes, d_offs, n, total, cfg, d_out)
    torch.cuda.synchronize()
    return d_out.cpu().numpy().view(pkg.CHIMERA_RESULT_DTYPE).reshape(-1)


def _oracle_chimera_params(sor, k, five_prime=False):
    if five_prime:
        p = sor.chimera_params(tso=k.adapter5p_complete.decode(), adapter=k.adapter3p5_complete.decode(), tso_max=k.adapter5p_complete_max_mm,
                               adapter_max=k.adapter3p5_complete_max_mm, bc_umi=0)
    else:
        p = sor.chimera_params(tso=k.tso_complete.decode(), adapter=k.adapter3p_complete.decode(), tso_max=k.tso_complete_max_mm,
                               adapter_max=k.adapter3p_complete_max_mm, bc_umi=16 + k.umi_length)
    p.internal_pat_len, p.internal_pat_frac, p.window_polya = k.internal_pat_len, k.internal_pat_frac, k.window_polya
    return p


KNOB_SETS_CHIM = [
    (False, dict(tso_complete_max_mm=3)), (False, dict(tso_complete_max_mm=9, adapter3p_complete_max_mm=2)), (False, dict(adapter3p_complete_max_mm=8)),
    (False, dict(internal_pat_len=12, internal_pat_frac=0.8)), (False, dict(internal_pat_len=15, internal_pat_frac=0.6)), (False, dict(internal_pat_len=8, internal_pat_frac=1.0)),
    (False, dict(umi_length=10)), (False, dict(window_polya=100, polya_len=12)),
    (False, dict(tso_complete=TSO_OTHER, adapter3p=AD_OTHER[-10:], adapter3p_complete=AD_OTHER)),
    (True, dict(adapter5p_complete_max_mm=3, adapter3p5_complete_max_mm=7)), (True, dict(internal_pat_len=10, internal_pat_frac=0.9)),
]


@pytest.mark.parametrize("five,over", KNOB_SETS_CHIM, ids=lambda o: str(o)[:60])
def test_splitter_knobs_equal_oracle(pkg, synth, sor, gpu_ctx, five, over):
    """maxCompleteSeqNeedlemanMismatches of the complete TSO / adapters, the complete sequences, internalpATlength / internalFractionATInPolyAT,
    windowSearchForPolyA and umi_length (the barcode + UMI stretch between an internal polyA and its adapter) through K-CHIM"""
    lib = __import__("importlib").import_module("sicelore_amd.lib")
    k = lib.run_knobs(**over)
    wl = synth.make_whitelist(20000, seed=1241)
    used = synth.pick_used(wl, 200, seed=1242)
    if five:
        reads = synth.gen_reads_5p(300, used, seed=1243, n_rate=0.002)
    else:
        reads = synth.gen_reads(300, used, seed=1243, n_rate=0.002, adapter_complete=k.adapter3p_complete.decode(),
                                tso_complete=k.tso_complete.decode() + "GGG", umi_len=k.umi_length)
    seqs = [c[0] for c in synth.make_chimeras(reads, 700, seed=1244)]
    rng = np.random.default_rng(1245)
    seqs += ["".join("ACGT"[b] for b in rng.integers(0, 4, L)) for L in (1, 239, 240, 441, 5000)] + ["A" * 700, "AT" * 400]
    res = _chimera_gpu(pkg, gpu_ctx, seqs, gpu_ctx.chimera_config(five, knobs=k))
    par = _oracle_chimera_params(sor, k, five)
    n_split = 0
    for i, s in enumerate(seqs):
        rc, splits, multi, n_matches, _ = sor.chimera_split(s, par)
        got = [(sor.SPLIT_REASONS[res["reason"][i][j]], int(res["pos"][i][j])) for j in range(res["n_split"][i])]
        assert rc == 0 and not (res["flags"][i] & 6), i
        assert got == splits and bool(res["flags"][i] & 1) == multi and res["n_matches"][i] == n_matches, (i, got, splits)
        n_split += len(splits) > 0
    assert n_split > 100


def _oracle_records_knobs(sor, bset, seqs, quals, k, max_ed=1, split=True, first_id=1):
    """the records the reference writes for a chunk under the knobs k (3' barcoding): splitter, scan with `sequence`, barcode, names"""
    par = _oracle_scan_params(sor, k)
    cpar = _oracle_chimera_params(sor, k)
    passed, failed, rid = [], [], first_id
    for i, (s, q) in enumerate(zip(seqs, quals)):
        name = f"read{i} runid=x ch={i % 9}"
        splits, multi, raw = [], False, None
        if split:
            rc, splits, multi, _, raw = sor.chimera_split(s, cpar)
            assert rc == 0
        cuts = [0] + [p for _, p in splits] + [len(s)]
        for j in range(len(cuts) - 1):
            fs, fq = s[cuts[j]:cuts[j + 1]], q[cuts[j]:cuts[j + 1]]
            fname = sor.chimera_fragment_name(name, raw, j) if splits else name
            rc, sc = sor.scan_read_3p(fs, fq, k.adapter3p.decode(), max_mm=k.adapter3p_max_mm, params=par)
            assert rc == 0
            a = None
            if sc["adapter_found"] and not multi:
                stranded = fs.encode().translate(COMP)[::-1] if sc["reverse"] else fs.encode()
                rc2, a_ = sor.assign_barcode(bset, stranded, int(sc["adapter_end"]), max_ed=max_ed)
                if rc2 == 1:
                    a = a_
            rec, ok = sor.fastq_record(fname, "", fs, fq, sc, a, rank=0, read_id=rid, force_failed=multi)
            assert rec is not None
            if ok:
                passed.append(rec)
                rid += 1
            else:
                failed.append(rec)
    return b"".join(passed), b"".join(failed)


@pytest.mark.parametrize("over", [dict(adapter3p_max_mm=1, min_read_length=450, tso_complete_max_mm=3),
                                  dict(adapter3p=AD_OTHER[-10:], adapter3p_complete=AD_OTHER, adapter3p_max_mm=4, internal_pat_frac=0.6, umi_length=10)],
                         ids=["mm1_len450_tso3", "other_adapter_umi10"])
def test_chunk_workers_take_the_knobs_of_their_context(pkg, synth, sor, gpu_ctx, over):
    """smi_ctx_set_knobs: the text worker and the packed worker of pass 2, and a lane created afterwards, write the records the oracle writes with
    the same knobs (splitter + scan + barcode + names); the context goes back to the shipped file with set_knobs(None)"""
    from test_write_gpu import _fastq

    lib = __import__("importlib").import_module("sicelore_amd.lib")
    k = lib.run_knobs(**over)
    wl = synth.make_whitelist(20_000, seed=1251)
    used = synth.pick_used(wl, 100, seed=1252)
    reads = synth.gen_reads(260, used, seed=1253, n_rate=0.002, adapter_complete=k.adapter3p_complete.decode(), umi_len=k.umi_length)
    chim = synth.make_chimeras(reads, 340, seed=1254)
    seqs, quals = [c[0] for c in chim], [c[1] for c in chim]
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    text = _fastq(seqs, quals)
    base_p, base_f, _ = gpu_ctx.scanfastq_pass2_chunk(text)
    base_p, base_f = bytes(base_p), bytes(base_f)
    gpu_ctx.set_knobs(k)
    try:
        assert gpu_ctx.get_knobs().as_dict() == k.as_dict()
        got_p, got_f, _ = gpu_ctx.scanfastq_pass2_chunk(text)
        got_p, got_f = bytes(got_p), bytes(got_f)
        pk_p, pk_f, _ = gpu_ctx.scanfastq_pass2_chunk(text, packed=True, n_threads=2)
        assert bytes(pk_p) == got_p and bytes(pk_f) == got_f
        lane = gpu_ctx.lane()
        try:
            ln_p, ln_f, _ = lane.scanfastq_pass2_chunk(text)
            assert bytes(ln_p) == got_p and bytes(ln_f) == got_f
        finally:
            lane.close()
    finally:
        gpu_ctx.set_knobs(None)
    exp_p, exp_f = _oracle_records_knobs(sor, sor.BarcodeSet(used.numpy()), seqs, quals, k)
    assert got_p == exp_p and got_f == exp_f
    assert got_p != base_p
    again_p, again_f, _ = gpu_ctx.scanfastq_pass2_chunk(text)
    assert bytes(again_p) == base_p and bytes(again_f) == base_f


def test_pass1_worker_takes_the_knobs(pkg, synth, sor, gpu_ctx):
    """pass 1 of a context with knobs: the histogram over the whitelist equals the count of the oracle's pass1_ok reads whose barcode is in the list,
    text worker == packed worker"""
    from test_write_gpu import _fastq

    lib = __import__("importlib").import_module("sicelore_amd.lib")
    k = lib.run_knobs(min_mean_bc_qv=11, min_mean_read_qv=10, min_adapter_3p_matches=6, adapter3p_max_mm=4)
    par = _oracle_scan_params(sor, k)
    wl = synth.make_whitelist(30_000, seed=1261)
    used = synth.pick_used(wl, 100, seed=1262)
    n = 1500
    reads = synth.gen_reads(n, used, seed=1263, n_rate=0.002, q_mean=13.0)
    seqs, quals = zip(*(synth.materialize(reads, i) for i in range(n)))
    text = _fastq(list(seqs), list(quals))
    gpu_ctx.set_barcode_set(wl.numpy().astype(np.uint64), mode=1)
    hists = {}
    for which, kn in (("shipped", None), ("knobs", k)):
        gpu_ctx.set_knobs(kn)
        try:
            for packed in (False, True):
                d_hist = torch.zeros(gpu_ctx.n_keys, dtype=torch.int32, device="cuda")
                gpu_ctx.scanfastq_pass1_chunk(text, d_hist, packed=packed, n_threads=2)
                torch.cuda.synchronize()
                hists[(which, packed)] = d_hist.cpu().numpy()
        finally:
            gpu_ctx.set_knobs(None)
    assert (hists[("knobs", False)] == hists[("knobs", True)]).all() and (hists[("shipped", False)] == hists[("shipped", True)]).all()
    assert (hists[("knobs", False)] != hists[("shipped", False)]).any()
    wl_sorted = np.sort(wl.numpy())
    exp = np.zeros(wl_sorted.size, dtype=np.int64)
    for s, q in zip(seqs, quals):
        rc, sc = sor.scan_read_3p(s, q, k.adapter3p_complete.decode(), max_mm=k.adapter3p_max_mm, params=par)
        assert rc == 0
        if not sc["pass1_ok"]:
            continue
        stranded = s.encode().translate(COMP)[::-1] if sc["reverse"] else s.encode()
        ae = int(sc["adapter_end"])
        key = sor.revcomp(sor.encode(stranded[ae - 17:ae - 1].decode()))     # (UsedCellBCListGenerator$Worker.java:L207-219, as tests/test_scan_gpu.py::test_pass1_histogram)
        j = np.searchsorted(wl_sorted, key)
        if j < wl_sorted.size and wl_sorted[j] == key:
            exp[j] += 1
    assert exp.sum() > 20
    assert (hists[("knobs", False)].astype(np.int64) == exp).all()


def test_knobs_outside_the_build_are_refused_by_name(pkg, gpu_ctx):
    lib = __import__("importlib").import_module("sicelore_amd.lib")
    for over, word in ((dict(adapter3p="CTTCCGATCTA"), "adapter_for3pBarcoding/sequence"), (dict(tso_complete="AAGCAGTGGTATCAACGCAGAGTACATGG"), "tso_for3pBarcoding/sequence_complete"),
                       (dict(umi_length=16), "umis/umi_length"), (dict(internal_pat_len=20), "polyAT/internalpATlength"), (dict(adapter3p="CTTCCGATCN"), "adapter_for3pBarcoding/sequence"),
                       (dict(window_polya=170), "windowSearchForPolyA"), (dict(adapter5p_window=170), "AdapterSearchWindow"),
                       (dict(tso_scan="AACGCAGAGTACATGGG"), "tso_for3pBarcoding/sequence"), (dict(tso_scan="AACGCAGAGTACATG"), "tso_for3pBarcoding/sequence"),
                       (dict(tso_scan_window=130), "tso_for3pBarcoding/windowForTSOsearch"), (dict(tso_scan_max_mm=-1), "tso_for3pBarcoding/maxNeedlemanMismatches")):
        with pytest.raises(lib.SmiError) as e:
            gpu_ctx.set_knobs(lib.run_knobs(**over))
        assert word in str(e.value), (over, str(e.value))
    assert gpu_ctx.get_knobs().as_dict() == lib.run_knobs().as_dict()
    # minReadLength below what the reference cuts off each read end (175 bases with the shipped window): its run dies on the first read in between
    # (PolyATSearcher.java:L178-181 substring); refused where the effective configuration is known, i.e. by the scan
    import torch
    with pytest.raises(lib.SmiError) as e:
        gpu_ctx.scan_device(torch.zeros((28, 2), dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda"), 1,
                            gpu_ctx.scan_config(2, knobs=lib.run_knobs(min_read_length=100)), torch.zeros((1, 8), dtype=torch.int32, device="cuda"))
    assert "readscanner/minReadLength" in str(e.value)
    with pytest.raises(lib.SmiError) as e:   # the TSO scan of 3' barcoding cuts windowForTSOsearch + 26 bases (scanReadForTSOs L128-131): 138 here, above the polyA finder's 125
        gpu_ctx.scan_device(torch.zeros((28, 2), dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda"), 1,
                            gpu_ctx.scan_config(2, knobs=lib.run_knobs(min_read_length=130, window_polya=100, tso_scan_window=112)), torch.zeros((1, 8), dtype=torch.int32, device="cuda"))
    assert "readscanner/minReadLength" in str(e.value)
    gpu_ctx.scan_device(torch.zeros((28, 2), dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda"), 1,
                        gpu_ctx.scan_config(2, knobs=lib.run_knobs(min_read_length=150), five_prime=True, dont_search_polya=True),
                        torch.zeros((1, 8), dtype=torch.int32, device="cuda"))        # 5' -y cuts 110 + 10 + 4 + 5 = 129 bases only


# ---- the accuracy simulations: scanfastq -e (random barcodes), assignumis -f (random UMIs) ----------------------------------------------------------
def _splitmix(seed, k):
    z = (int(seed) + 0x9E3779B97F4A7C15 * (int(k) + 1)) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def test_random_barcode_simulation_through_the_chunk_workers(pkg, synth, sor, gpu_ctx):
    """scanfastq -e (Parser.java:L212-215; README.md:176 "Specifity can be estimated by re-running the program with the -e option"): with
    smi_ctx_set_random_barcodes the matcher of pass 2 sees random window bases drawn from (seed, read id) -- the barcode results of the text worker and
    of the packed worker equal K-BC on exactly those windows (the ORACLE's assignment on them for a sample), almost nothing is assigned any more, the
    same seed gives the same run, another seed another one, and seed 0 is the ordinary run again"""
    from test_write_gpu import _fastq

    lib = __import__("importlib").import_module("sicelore_amd.lib")
    wl = synth.make_whitelist(20_000, seed=1301)
    used = synth.pick_used(wl, 2000, seed=1302)
    n = 3000
    reads = synth.gen_reads(n, used, seed=1303, n_rate=0.002)
    seqs, quals = zip(*(synth.materialize(reads, i) for i in range(n)))
    text = _fastq(list(seqs), list(quals))
    keys = used.numpy().astype(np.uint64)
    gpu_ctx.set_barcode_set(keys, mode=0)
    _p, _f, base = gpu_ctx.scanfastq_pass2_chunk(text, split_chimeras=False, want_results=True, first_read_id=100)
    base_bc = base["bc"].copy()
    assert (base_bc["found"] == 1).sum() > 0.5 * n
    # the windows K-SCAN leaves for these reads, and their random stand-ins
    ra = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s_) for s_ in seqs])
    _scan, win = gpu_ctx.scan_batch(ra, None, offs, gpu_ctx.scan_config(2))
    runs = {}
    for seed in (7, 7, 8):
        gpu_ctx.set_random_barcodes(seed)
        try:
            res = []
            for packed in (False, True):
                p_, f_, info = gpu_ctx.scanfastq_pass2_chunk(text, split_chimeras=False, want_results=True, first_read_id=100, packed=packed, n_threads=2)
                res.append((bytes(p_), bytes(f_), info["bc"].copy()))
            assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and (res[0][2] == res[1][2]).all()
        finally:
            gpu_ctx.set_random_barcodes(0)
        rnd = win.copy()
        for i in range(n):
            if rnd["flags"][i] & 1:
                rnd["bases"][i] = _splitmix(seed, 100 + i) & ((1 << 48) - 1)
                rnd["nmask"][i] = 0
        exp = gpu_ctx.bc_match(rnd, max_ed=1)
        got = res[0][2]
        for f in ("found", "bc", "ed", "ed_sec", "offset", "ins_minus_del"):
            sel = rnd["flags"] & 1 == 1
            assert (got[f][sel] == exp[f][sel]).all(), (seed, f)
        runs.setdefault(seed, []).append(res[0])
        n_chance = int((got["found"] == 1).sum())
        assert n_chance < 0.1 * (base_bc["found"] == 1).sum()        # 2,000 barcodes x 124 x 5 of 4^16 sequences: a chance hit is rare
    assert runs[7][0][0] == runs[7][1][0] and (runs[7][0][2] == runs[7][1][2]).all()      # the same seed: the same run
    assert runs[7][0][0] != runs[8][0][0]
    # a sample of the random windows against the oracle's Parser.assignBarcode on the same bases
    bset = sor.BarcodeSet(used.numpy())
    dec = "AGCT"
    checked = 0
    for i in range(0, n, 7):
        if not (win["flags"][i] & 1):
            continue
        v = _splitmix(7, 100 + i) & ((1 << 48) - 1)
        w24 = "".join(dec[(v >> (2 * (23 - j))) & 3] for j in range(24))      # stranded[AE-22 .. AE+1]
        stranded = ("A" * 30 + w24 + "A" * 8).encode()
        rc, a = sor.assign_barcode(bset, stranded, 30 + 22, max_ed=1)
        g = runs[7][0][2][i]
        assert (rc == 1) == (g["found"] == 1), i
        if rc == 1:
            assert int(g["bc"]) == int(a["bc"]) & 0xFFFFFFFF and g["ed"] == a["ed"]
        checked += 1
    assert checked > 300
    _p2, _f2, again = gpu_ctx.scanfastq_pass2_chunk(text, split_chimeras=False, want_results=True, first_read_id=100)
    assert (again["bc"] == base_bc).all()


@pytest.mark.parametrize("five,ul", [(False, 12), (True, 10)])
def test_random_umi_simulation_in_the_umi_stage(pkg, gpu_ctx, five, ul, monkeypatch):
    """assignumis -f (ClusteringEditDistanceBase.java:L308-310): with smi_assignumis_config.random_umi_seed every read's UMI window is a random one drawn
    from (seed, position in the chunk): U7 shows it, the device stage and the host path agree, and the deep molecules of the input no longer cluster"""
    from test_umi_gpu import _make_groups, _name_with_window_len

    lib = __import__("importlib").import_module("sicelore_amd.lib")
    n = 600
    ws = _make_groups(44, [n])[:, :ul + 2]
    names = [_name_with_window_len(i, ws[i], "15", ul, five) for i in range(n)]
    pos0 = np.full(n, 100_000, dtype=np.int32)
    cig = [np.array([1000 << 4], dtype=np.uint32)] * n
    real, _nd = gpu_ctx.assignumis_chunk(names, np.zeros(n, np.uint16), pos0, cig, five_prime=five, umi_length=ul)
    assert ((real["flags"] & lib.UMI_CLUSTERED) != 0).sum() > 0.5 * n
    dec = {0: "A", 1: "G", 2: "C", 3: "T"}
    outs = []
    for host in (False, True):
        if host:
            monkeypatch.setenv("SMI_AU_HOST", "1")
        t, nd = gpu_ctx.assignumis_chunk(names, np.zeros(n, np.uint16), pos0, cig, five_prime=five, umi_length=ul, random_umi_seed=11)
        assert nd == n
        outs.append(t.copy())
    monkeypatch.delenv("SMI_AU_HOST")
    assert (outs[0] == outs[1]).all()
    for i in range(n):
        z = _splitmix(11, i)
        want = "".join(dec[(z >> (2 * k)) & 3] for k in range(1, ul + 1))       # U7 = bases 1 .. umi_length of the window
        assert outs[0]["u7"][i].decode() == want, i
    # (600 random UMIs in ONE group: the pairs within two edits by chance are what the simulation is there to count -- fewer than the real molecules give)
    assert ((outs[0]["flags"] & lib.UMI_CLUSTERED) != 0).sum() < ((real["flags"] & lib.UMI_CLUSTERED) != 0).sum()
