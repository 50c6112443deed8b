"""The packed boundary of scanfastq on the device (smi_scanfastq_pass2_chunk_packed / _pass1_chunk_packed: bit-planes up, decisions down,
records written by host threads) against the text workers, which tests/test_write_gpu.py and tests/test_ref_exec_gpu.py hold to the
oracle and to the reference's own records: same bytes, same counters, in every configuration; k_ends_from_planes against K-PACK."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fastq(seqs, quals, eol="\n", qh=lambda i: "", name=lambda i: f"read{i} runid=x ch={i % 9}"):
    return "".join(f"@{name(i)}{eol}{s}{eol}+{qh(i)}{eol}{q}{eol}" for i, (s, q) in enumerate(zip(seqs, quals))).encode()


def _reads(synth, n, seed):
    wl = synth.make_whitelist(20_000, seed=seed)
    used = synth.pick_used(wl, 150, seed=seed + 1)
    return used, synth.gen_reads(n, used, seed=seed + 2, n_rate=0.002)


def _both(ctx, text, **kw):
    a = ctx.scanfastq_pass2_chunk(text, want_results=True, **kw)
    b = ctx.scanfastq_pass2_chunk(text, want_results=True, packed=True, n_threads=3, **kw)
    assert bytes(a[0]) == bytes(b[0]) and bytes(a[1]) == bytes(b[1])
    for k in ("n_records_in", "n_records_out", "n_passed"):
        assert a[2][k] == b[2][k], k
    if a[2]["n_records_out"]:
        assert a[2]["scan"].tobytes() == b[2]["scan"].tobytes() and a[2]["bc"].tobytes() == b[2]["bc"].tobytes()
    return a


def test_ends_from_planes_equal_k_pack(pkg, synth, gpu_ctx):
    """read ends cut out of the planes the HOST packed == K-PACK's ends from ASCII, whole reads and fragments, every length around the
    224-base end and the 32-base word"""
    import torch

    from sicelore_amd import lib as libmod

    rng = random.Random(3)
    dev = torch.device("cuda", gpu_ctx.device)
    lens = list(range(1, 70)) + list(range(190, 260)) + [447, 448, 449, 1000, 3000] + [rng.randrange(300, 2500) for _ in range(400)]
    seqs = ["".join(rng.choice("ACGTN" if i % 7 == 0 else "ACGT") for _ in range(ln)) for i, ln in enumerate(lens)]
    text = _fastq(seqs, ["I" * len(s) for s in seqs])
    recs, offs, err = libmod.fastq_index_host(text, n_threads=2)
    assert err == 0
    n, total = len(seqs), int(offs[-1])
    planes = libmod.pack_reads_host(text, recs, offs, n_threads=2)
    d_text = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).to(dev)
    d_ss = torch.from_numpy(recs["seq_start"].astype(np.int64)).to(dev)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    # K-PACKR on the device gives the same planes (gap words excepted: the host zeroes them, the device leaves them)
    d_pl = torch.zeros(planes.size, dtype=torch.int32, device=dev)
    gpu_ctx.pack_reads_text_device(d_text, d_ss, d_offs, n, total, d_pl)
    assert (d_pl.cpu().numpy().view(np.uint32) == planes).all()
    # fragments: cut every read longer than 600 bases into two or three pieces
    foffs, fsrc = [0], []
    for i, s in enumerate(seqs):
        cuts = [0] + ([len(s) // 3, 2 * len(s) // 3 + 5] if len(s) > 1200 else [len(s) // 2 + 7] if len(s) > 600 else []) + [len(s)]
        for k in range(len(cuts) - 1):
            foffs.append(int(offs[i]) + cuts[k + 1])
            fsrc.append(i << 2 | k)
    m = len(fsrc)
    d_planes = torch.from_numpy(planes.view(np.int32)).to(dev)
    for frag in (False, True):
        mm = m if frag else n
        d_ro = torch.from_numpy(np.array(foffs, dtype=np.int64)).to(dev) if frag else d_offs
        d_fs = torch.from_numpy(np.array(fsrc, dtype=np.uint32).view(np.int32)).to(dev) if frag else None
        ends_a, ends_b = torch.zeros((28, 2 * mm), dtype=torch.int32, device=dev), torch.zeros((28, 2 * mm), dtype=torch.int32, device=dev)
        len_a, len_b = torch.zeros(mm, dtype=torch.int32, device=dev), torch.zeros(mm, dtype=torch.int32, device=dev)
        bstart = torch.zeros(mm, dtype=torch.int64, device=dev)
        gpu_ctx.frag_text_starts_device(d_ss, None, d_offs, d_ro if frag else None, d_fs, mm, bstart, None)
        gpu_ctx.pack_ends_text_device(d_text, bstart, d_ro, mm, ends_a, len_a)
        gpu_ctx.ends_from_planes_device(d_planes, d_offs, n, total, d_ro, d_fs, mm, ends_b, len_b)
        torch.cuda.synchronize()
        assert bool((len_a == len_b).all())
        assert bool((ends_a == ends_b).all())


def test_packed_equals_text_worker_3p(pkg, synth, gpu_ctx):
    used, reads = _reads(synth, 300, 981)
    chim = synth.make_chimeras(reads, 380, seed=984)
    seqs, quals = [c[0] for c in chim] + ["ACGT" * 30], [c[1] for c in chim] + ["5" * 120]
    keys = np.sort(used.numpy().astype(np.uint64))
    ranks = (np.arange(keys.size) % 50 + 1).astype(np.int32)
    text = _fastq(seqs, quals, qh=lambda i: "x" if i % 7 == 0 else "")
    gpu_ctx.set_barcode_set(keys, mode=0)
    for kw in (dict(), dict(trim_fastq=True), dict(split_chimeras=False), dict(max_ed=0), dict(max_ed=2)):
        p, f, info = _both(gpu_ctx, text, first_read_id=500, rank_keys=keys, rank_values=ranks, **kw)
        assert info["n_passed"] > 200
    p, f, info = _both(gpu_ctx, text)
    assert p.count(b"sp1") > 10 and p.count(b"_REV_") > 50
    # CR LF, names without a blank, a second (smaller) chunk on the same context, the empty chunk, a malformed one
    _both(gpu_ctx, _fastq(seqs[:50], quals[:50], eol="\r\n", name=lambda i: f"read{i}"))
    assert gpu_ctx.scanfastq_pass2_chunk(b"", packed=True)[2]["n_records_in"] == 0
    with pytest.raises(pkg.SmiError):
        gpu_ctx.scanfastq_pass2_chunk(b"@r1\nACGT\n-\nIIII\n", packed=True)


def test_packed_equals_text_worker_5p(pkg, synth, gpu_ctx):
    wl = synth.make_whitelist(20_000, seed=991)
    used = synth.pick_used(wl, 120, seed=992)
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    r5 = synth.gen_reads_5p(150, used, seed=993)
    text5 = _fastq(*zip(*(synth.materialize(r5, i) for i in range(150))))
    for kw in (dict(dont_search_polya=True), dict(dont_search_polya=False), dict(dont_search_polya=True, trim_fastq=True)):
        _both(gpu_ctx, text5, five_prime=True, **kw)


def test_packed_tiny_reads_long_names_long_reads(pkg, synth, gpu_ctx):
    rng = random.Random(77)
    used, reads = _reads(synth, 40, 951)
    base = [synth.materialize(reads, i) for i in range(40)]
    seqs, quals, names, qhs = [], [], [], []
    for n in list(range(1, 71)) + [199, 200, 201, 224, 225, 447, 448, 449]:
        seqs.append("".join(rng.choice("ACGTN") for _ in range(n)))
        quals.append("".join(chr(33 + rng.randrange(40)) for _ in range(n)))
    for s, q in base:
        seqs.append(s)
        quals.append(q)
    long_reads = synth.gen_reads(12, used, seed=973, n_rate=0.001, max_mid=30_000)
    for i in range(12):
        s, q = synth.materialize(long_reads, i)
        seqs.append(s)
        quals.append(q)
    for i in range(len(seqs)):
        ln = rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 120, 300])
        tok = "".join(rng.choice("abcdefghijklmnopqrstuvwxyz0123456789-") for _ in range(ln))
        names.append(tok if i % 3 == 0 else tok + " " + "x" * rng.choice([0, 1, 40, 150]))
        qhs.append("" if i % 2 else "h" * rng.choice([1, 30, 200]))
    text = "".join(f"@{nm}\n{s}\n+{h}\n{q}\n" for nm, s, h, q in zip(names, seqs, qhs, quals)).encode()
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    _both(gpu_ctx, text)
    _both(gpu_ctx, text[:-1])     # no final newline


def test_packed_large_chunk_and_lanes(pkg, synth, gpu_ctx):
    """60 k reads with 10 % chimeras through both workers (several host threads, the speculative index split), and two lanes at once"""
    import threading

    import torch

    dev = torch.device("cuda", gpu_ctx.device)
    wl = synth.make_whitelist(50_000, seed=971, device=dev)
    used = synth.pick_used(wl, 300, seed=972)
    gpu_ctx.set_barcode_set_device(used.to(torch.int32), mode=0)
    rd = synth.gen_reads(60_000, used, seed=973, device=dev)
    text = synth.fastq_text_device(rd, chimera_frac=0.10)[0].cpu().numpy()
    a = gpu_ctx.scanfastq_pass2_chunk(text, copy=True)
    b = gpu_ctx.scanfastq_pass2_chunk(text, copy=True, packed=True, n_threads=8)
    assert a[0] == b[0] and a[1] == b[1] and a[2]["n_passed"] == b[2]["n_passed"] > 40_000
    lanes = [gpu_ctx.lane(), gpu_ctx.lane()]
    got = [None, None]

    def work(k):
        for _ in range(2):
            p, f, _ = lanes[k].scanfastq_pass2_chunk(text, copy=True, packed=True, n_threads=4)
            got[k] = (p, f)

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got[0] == (a[0], a[1]) and got[1] == (a[0], a[1])
    for ln in lanes:
        ln.close()


def test_packed_pass1_equals_text_worker(pkg, synth, gpu_ctx):
    import torch

    used, reads = _reads(synth, 400, 985)
    seqs, quals = zip(*(synth.materialize(reads, i) for i in range(400)))
    seqs, quals = list(seqs) + ["ACGT" * 30, "A" * 230], list(quals) + ["5" * 120, "I" * 230]
    text = _fastq(seqs, quals)
    wl = synth.make_whitelist(30_000, seed=986)
    wkeys = np.sort(np.unique(np.concatenate([wl.numpy().astype(np.uint64), used.numpy().astype(np.uint64)])))
    gpu_ctx.set_barcode_set(wkeys, mode=1)
    h1 = torch.zeros(wkeys.size, dtype=torch.int32, device="cuda")
    h2 = torch.zeros(wkeys.size, dtype=torch.int32, device="cuda")
    n1 = gpu_ctx.scanfastq_pass1_chunk(text, h1)
    n2 = gpu_ctx.scanfastq_pass1_chunk(text, h2, packed=True, n_threads=3)
    assert n1 == n2 == len(seqs) and bool((h1 == h2).all()) and int(h1.sum()) > 50
    # 5' (the quality filter then reads the FIRST qualities)
    r5 = synth.gen_reads_5p(300, used, seed=987)
    text5 = _fastq(*zip(*(synth.materialize(r5, i) for i in range(300))))
    h1.zero_()
    h2.zero_()
    torch.cuda.synchronize()
    gpu_ctx.scanfastq_pass1_chunk(text5, h1, five_prime=True, dont_search_polya=True)
    gpu_ctx.scanfastq_pass1_chunk(text5, h2, five_prime=True, dont_search_polya=True, packed=True)
    assert bool((h1 == h2).all()) and int(h1.sum()) > 20
