"""K-PACKR + K-CHIM (chimera splitter) on the GPU == oracle, and the fragment offsets kernel."""
import numpy as np
import pytest

import pymodel_chimera as pm


def _run(pkg, ctx, seqs, five_prime=False):
    import torch

    dev = torch.device("cuda:0")
    n = len(seqs)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    total = int(offs[-1])
    ra = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    d_reads = torch.from_numpy(ra.copy()).to(dev)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_planes = torch.full((ctx.read_planes_words(total, n),), -1, dtype=torch.int32, device=dev)  # garbage-filled
    ctx.pack_reads_device(d_reads, d_offs, n, total, d_planes)
    d_out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    ctx.chimera_device(d_planes, d_offs, n, total, ctx.chimera_config(five_prime), d_out)
    torch.cuda.synchronize()
    res = d_out.cpu().numpy().view(pkg.CHIMERA_RESULT_DTYPE).reshape(-1)
    return res, d_out, d_offs, offs


def _compare(sor, seqs, res):
    n_split = n_multi = 0
    for i, s in enumerate(seqs):
        rc, splits, multi, n_matches, _ = sor.chimera_split(s)
        got = [(sor.SPLIT_REASONS[res["reason"][i][k]], int(res["pos"][i][k])) for k in range(res["n_split"][i])]
        assert rc == 0 and not (res["flags"][i] & 6), i
        assert got == splits and bool(res["flags"][i] & 1) == multi and res["n_matches"][i] == n_matches, (i, got, splits)
        n_split += len(splits) > 0
        n_multi += multi
    return n_split, n_multi


@pytest.mark.gpu
def test_chimera_matches_oracle_on_synthetic_chimeras(pkg, sor, synth):
    wl = synth.make_whitelist(20000, seed=81)
    used = synth.pick_used(wl, 200, seed=82)
    reads = synth.gen_reads(500, used, seed=83, n_rate=0.002)
    chim = synth.make_chimeras(reads, 1500, seed=84)
    seqs = [c[0] for c in chim]
    rng = np.random.default_rng(5)
    seqs += ["".join("ACGT"[k] for k in rng.integers(0, 4, L)) for L in (1, 100, 239, 240, 439, 440, 441, 500, 5000)]
    seqs += ["A" * 700, "T" * 1000, "AT" * 400]
    ctx = pkg.Context(0)
    res, *_ = _run(pkg, ctx, seqs)
    n_split, n_multi = _compare(sor, seqs, res)
    assert n_split > 500 and n_multi > 50


@pytest.mark.gpu
def test_reads_with_more_than_64_internal_hits_are_computed_not_flagged(pkg, sor, synth):
    """the reference has no cap on internal TSO / adapter hits (ChimeraFindernew.java:L107-332): all-N reads, TSO concatemers,
    homopolymers and low-complexity reads go through K-CHIM-S and equal the oracle; a chunk that holds them does not fail"""
    import random

    rng = random.Random(23)
    tso, tso_rc = "AAGCAGTGGTATCAACGCAGAGTACAT", "ATGTACTCTGCGTTGATACCACTGCTT"
    rnd = lambda k: "".join(rng.choice("ACGT") for _ in range(k))  # noqa: E731
    seqs = ["N" * 600, "N" * 2300,
            rnd(150) + "".join(tso + rnd(rng.randrange(95, 140)) for _ in range(90)) + rnd(150),           # 90 spaced TSO copies
            rnd(100) + "".join((tso if k % 2 else tso_rc) + rnd(130) for k in range(80)) + rnd(100),       # both orientations
            rnd(200) + "".join(tso[:rng.randrange(20, 27)] + rnd(3) for _ in range(200)) + rnd(200),       # dense, overlapping hits
            rnd(300) + ("A" * 40 + "CTACACGACGCTCTTCCGATCT"[::-1] + rnd(60)) * 30 + rnd(300),             # many polyA stretches
            "ACGTN" * 500, "AAGCAGTGGTATCAACGCAGAGTACAT" * 60]
    reads = synth.gen_reads(40, synth.pick_used(synth.make_whitelist(5000, seed=91), 20, seed=92), seed=93)
    seqs += [synth.materialize(reads, i)[0] for i in range(40)]
    ctx = pkg.Context(0)
    res, *_ = _run(pkg, ctx, seqs)
    assert not (res["flags"] & pkg.lib.CHIM_OVERFLOW).any()
    n_big = 0
    for i, s in enumerate(seqs):
        rc, splits, multi, n_matches, _ = sor.chimera_split(s)
        got = [(sor.SPLIT_REASONS[res["reason"][i][k]], int(res["pos"][i][k])) for k in range(res["n_split"][i])]
        if rc != 0:
            assert res["flags"][i] & pkg.lib.CHIM_RANGE, i   # the reference's substring would throw
            continue
        assert got == splits and bool(res["flags"][i] & 1) == multi and res["n_matches"][i] == n_matches, (i, got, splits, n_matches)
        n_big += n_matches > 64
    assert n_big >= 2
    # the native chunk worker takes the same reads in its stride
    text = "".join(f"@r{i} x\n{s}\n+\n{'5' * len(s)}\n" for i, s in enumerate(seqs) if not (res["flags"][i] & pkg.lib.CHIM_RANGE)).encode()
    ctx.set_barcode_set(np.arange(100, dtype=np.uint64) * 977, mode=0)
    passed, failed, info = ctx.scanfastq_pass2_chunk(text)
    assert info["n_records_in"] >= 40 and info["n_records_out"] >= info["n_records_in"]


@pytest.mark.gpu
def test_chimera_hand_built_reads_and_model(pkg, sor):
    import random

    from test_oracle_chimera import AD22, molecule, rc, rnd

    rng = random.Random(17)
    seqs = []
    for _ in range(20):
        a, b = molecule(rng), molecule(rng)
        seqs += [a + b, rc(a) + b, a + rc(b), rc(a + b)]
        core = AD22 + rnd(rng, 16) + rnd(rng, 12) + "T" * rng.randrange(16, 60)
        seqs += [rnd(rng, 600) + core + rnd(rng, 600), rc(rnd(rng, 500) + core + rnd(rng, 700))]
    ctx = pkg.Context(0)
    res, *_ = _run(pkg, ctx, seqs)
    _compare(sor, seqs, res)
    for i in range(0, len(seqs), 7):  # the Python model on a subset (slow)
        m_splits, m_multi, _ = pm.find_split_positions(seqs[i])
        got = [(sor.SPLIT_REASONS[res["reason"][i][k]], int(res["pos"][i][k])) for k in range(res["n_split"][i])]
        assert got == m_splits and bool(res["flags"][i] & 1) == m_multi


@pytest.mark.gpu
def test_split_offsets(pkg, sor, synth):
    import torch

    wl = synth.make_whitelist(5000, seed=91)
    used = synth.pick_used(wl, 50, seed=92)
    reads = synth.gen_reads(200, used, seed=93, max_mid=300)
    seqs = [c[0] for c in synth.make_chimeras(reads, 3000, seed=94)]
    ctx = pkg.Context(0)
    res, d_out, d_offs, offs = _run(pkg, ctx, seqs)
    n = len(seqs)
    dev = d_out.device
    d_scratch = torch.zeros((n + 1023) // 1024 + 1, dtype=torch.int32, device=dev)
    d_nfrag = torch.zeros(1, dtype=torch.int64, device=dev)
    d_fo = torch.zeros(3 * n + 1, dtype=torch.int64, device=dev)
    d_src = torch.zeros(3 * n, dtype=torch.int32, device=dev)
    ctx.split_offsets_device(d_out, d_offs, n, d_scratch, d_nfrag, d_fo, d_src)
    torch.cuda.synchronize()
    exp_off, exp_src = [], []
    for i in range(n):
        cuts = [0] + [int(res["pos"][i][k]) for k in range(res["n_split"][i])]
        for j, c in enumerate(cuts):
            exp_off.append(int(offs[i]) + c)
            exp_src.append((i << 2) | j)
    exp_off.append(int(offs[n]))
    nf = int(d_nfrag.item())
    assert nf == len(exp_src) and nf > n
    assert d_fo.cpu().numpy()[:nf + 1].tolist() == exp_off
    assert d_src.cpu().numpy()[:nf].tolist() == exp_src


@pytest.mark.gpu
def test_chimera_5p_configuration(pkg, sor, synth):
    wl = synth.make_whitelist(20000, seed=95)
    used = synth.pick_used(wl, 200, seed=96)
    reads = synth.gen_reads_5p(400, used, seed=97, n_rate=0.002)
    seqs = [c[0] for c in synth.make_chimeras(reads, 1000, seed=98)]
    ctx = pkg.Context(0)
    res, *_ = _run(pkg, ctx, seqs, five_prime=True)
    par = sor.chimera_params(tso="CTACACGACGCTCTTCCGATCT", adapter="AAGCAGTGGTATCAACGCAGAGTAC", tso_max=5, adapter_max=5,
                             bc_umi=0)
    n_split = 0
    for i, s in enumerate(seqs):
        rc, splits, multi, n_matches, _ = sor.chimera_split(s, par)
        got = [(sor.SPLIT_REASONS[res["reason"][i][k]], int(res["pos"][i][k])) for k in range(res["n_split"][i])]
        assert rc == 0 and not (res["flags"][i] & 6)
        assert got == splits and bool(res["flags"][i] & 1) == multi and res["n_matches"][i] == n_matches, (i, got, splits)
        n_split += len(splits) > 0
    assert n_split > 300


@pytest.mark.gpu
def test_flat_read_packer_equals_the_wave_per_read_one(pkg, synth, monkeypatch):
    """K-PACKR as a thread per plane word (round 5) against the wave-per-read kernel (SMI_PACKR_WAVE) on reads of every length class -- empty, 1 .. 40 bases, around
    the 32-base word and 2048-base wave borders, tens of kilobases -- from a gathered array and from FASTQ text: the same planes word for word"""
    import torch

    from sicelore_amd import lib as libmod

    ctx = pkg.Context(0)
    rng = np.random.default_rng(77)
    lens = np.concatenate([np.arange(0, 70), rng.integers(1, 3000, 600), np.array([2047, 2048, 2049, 4096, 4097, 31, 32, 33, 63, 64, 65, 30_000, 70_001]),
                           rng.integers(200, 2000, 400)])
    rng.shuffle(lens)
    n = lens.size
    seqs = ["".join(rng.choice(list("ACGTN"), int(k), p=[0.245, 0.245, 0.245, 0.245, 0.02])) for k in lens]
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    total = int(offs[-1])
    d_reads = torch.from_numpy(np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()).cuda()
    d_offs = torch.from_numpy(offs).cuda()
    words = ctx.read_planes_words(total, n)
    # the same reads as FASTQ text (names, '+' lines, qualities between them)
    text, starts, at = [], [], 0
    for i, q in enumerate(seqs):
        head = f"@r{i} x\n"
        starts.append(at + len(head))
        rec = head + q + "\n+\n" + "I" * len(q) + "\n"
        text.append(rec)
        at += len(rec)
    d_text = torch.from_numpy(np.frombuffer("".join(text).encode(), dtype=np.uint8).copy()).cuda()
    d_starts = torch.from_numpy(np.array(starts, dtype=np.int64)).cuda()
    got = {}
    for mode in ("flat", "wave"):
        if mode == "wave":
            monkeypatch.setenv("SMI_PACKR_WAVE", "1")
        a = torch.zeros(words, dtype=torch.int32, device="cuda")
        b = torch.zeros(words, dtype=torch.int32, device="cuda")
        ctx.pack_reads_device(d_reads, d_offs, n, total, a)
        ctx.pack_reads_text_device(d_text, d_starts, d_offs, n, total, b)
        torch.cuda.synchronize()
        got[mode] = (a.cpu().numpy(), b.cpu().numpy())
    monkeypatch.delenv("SMI_PACKR_WAVE")
    assert (got["flat"][0] == got["wave"][0]).all() and (got["flat"][1] == got["wave"][1]).all() and (got["flat"][0] == got["flat"][1]).all()
    assert int((got["flat"][0] != 0).sum()) > total // 64
