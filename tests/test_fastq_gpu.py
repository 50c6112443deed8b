"""K-FQ: FASTQ text on the device -> record index + contiguous read / quality buffers == a plain Python parse."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _index(ctx, text, cap=None):
    n_bytes = len(text)
    d_text = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).cuda() if n_bytes else torch.zeros(1, dtype=torch.uint8, device="cuda")
    cap = cap or text.count(b"\n") // 4 + 2
    bufs = dict(line=torch.zeros(4 * cap + 8, dtype=torch.int64, device="cuda"))
    for k in ("name_start", "seq_start", "qual_start"):
        bufs[k] = torch.zeros(cap, dtype=torch.int64, device="cuda")
    for k in ("name_len", "seq_len"):
        bufs[k] = torch.zeros(cap, dtype=torch.int32, device="cuda")
    bufs["offsets"] = torch.zeros(cap + 1, dtype=torch.int64, device="cuda")
    n_rec, err = ctx.fastq_index_device(d_text, n_bytes, bufs["line"], bufs["name_start"], bufs["name_len"], bufs["seq_start"],
                                        bufs["seq_len"], bufs["qual_start"], bufs["offsets"], cap)
    return n_rec, err, d_text, bufs


def _records(n, rng, crlf=False, final_newline=True):
    recs = []
    for i in range(n):
        L = int(rng.integers(1, 3000)) if i % 50 else int(rng.integers(1, 5))
        seq = "".join("ACGTN"[k] for k in rng.integers(0, 5 if i % 7 == 0 else 4, L))
        qual = "".join(chr(33 + int(k)) for k in rng.integers(2, 40, L))
        recs.append((f"read{i} runid=ab{i % 13} ch={i % 512}", seq, qual))
    eol = "\r\n" if crlf else "\n"
    text = "".join(f"@{nm}{eol}{s}{eol}+{eol}{q}{eol}" for nm, s, q in recs)
    if not final_newline:
        text = text[:-len(eol)]
    return recs, text.encode()


@pytest.mark.parametrize("crlf,final_newline", [(False, True), (False, False), (True, True)])
def test_fastq_index_and_gather(pkg, gpu_ctx, crlf, final_newline):
    rng = np.random.default_rng(3 + crlf + 2 * final_newline)
    recs, text = _records(3000, rng, crlf, final_newline)
    n_rec, err, d_text, b = _index(gpu_ctx, text)
    assert n_rec == len(recs) and err == 0
    t = np.frombuffer(text, dtype=np.uint8)
    ns, nl = b["name_start"].cpu().numpy(), b["name_len"].cpu().numpy()
    ss, sl, qs = b["seq_start"].cpu().numpy(), b["seq_len"].cpu().numpy(), b["qual_start"].cpu().numpy()
    offs = b["offsets"].cpu().numpy()
    for i in (0, 1, 49, 50, 51, 1234, len(recs) - 1):
        nm, s, q = recs[i]
        assert bytes(t[ns[i]:ns[i] + nl[i]]).decode() == nm
        assert bytes(t[ss[i]:ss[i] + sl[i]]).decode() == s and bytes(t[qs[i]:qs[i] + sl[i]]).decode() == q
    assert (sl[:n_rec] == [len(r[1]) for r in recs]).all()
    assert (offs[:n_rec + 1] == np.concatenate([[0], np.cumsum([len(r[1]) for r in recs])])).all()
    total = int(offs[n_rec])
    d_reads = torch.zeros(total, dtype=torch.uint8, device="cuda")
    d_quals = torch.zeros(total, dtype=torch.uint8, device="cuda")
    gpu_ctx.fastq_gather_device(d_text, b["seq_start"], b["offsets"], n_rec, d_reads)
    gpu_ctx.fastq_gather_device(d_text, b["qual_start"], b["offsets"], n_rec, d_quals)
    torch.cuda.synchronize()
    assert bytes(d_reads.cpu().numpy()).decode() == "".join(r[1] for r in recs)
    assert bytes(d_quals.cpu().numpy()).decode() == "".join(r[2] for r in recs)


def test_fastq_index_one_sweep_equals_two_sweeps(pkg, gpu_ctx, monkeypatch):
    """the line index from one sweep over the text (the count sweep keeps the newline flags, the line starts come from those) against the two
    sweeps it replaced (SMI_FQ_TWO_SWEEPS), on a text of ~5,400 blocks with short and long lines mixed; and the capacity checks that moved
    to the device"""
    rng = np.random.default_rng(77)
    recs, text = _records(60_000, rng)
    assert len(text) > 80_000_000
    n1, e1, _t, b1 = _index(gpu_ctx, text)
    monkeypatch.setenv("SMI_FQ_TWO_SWEEPS", "1")
    n2, e2, _t, b2 = _index(gpu_ctx, text)
    monkeypatch.delenv("SMI_FQ_TWO_SWEEPS")
    assert (n1, e1) == (n2, e2) == (len(recs), 0)
    n_lines = 4 * n1
    assert torch.equal(b1["line"][:n_lines + 1], b2["line"][:n_lines + 1])
    for k in ("name_start", "name_len", "seq_start", "seq_len", "qual_start"):
        assert torch.equal(b1[k][:n1], b2[k][:n1]), k
    assert torch.equal(b1["offsets"][:n1 + 1], b2["offsets"][:n1 + 1])
    assert int(b1["offsets"][n1]) == sum(len(r[1]) for r in recs)
    for bad_cap in (n1, n1 // 2):   # record buffers need n_records + 1 entries
        with pytest.raises(pkg.SmiError, match="too small"):
            _index(gpu_ctx, text, cap=bad_cap)
    assert _index(gpu_ctx, text, cap=n1 + 1)[:2] == (n1, 0)


def test_fastq_errors_are_reported(pkg, gpu_ctx):
    ok = b"@r1\nACGT\n+\nIIII\n@r2\nAC\n+\nII\n"
    assert _index(gpu_ctx, ok)[:2] == (2, 0)
    assert _index(gpu_ctx, b"")[:2] == (0, 0)
    assert _index(gpu_ctx, ok.replace(b"@r2", b"r2@"))[1] == 1          # SMI_FQ_BAD_SEQ_HEADER
    assert _index(gpu_ctx, ok.replace(b"+\nII", b"-\nII"))[1] == 2      # SMI_FQ_BAD_QUAL_HEADER
    assert _index(gpu_ctx, ok.replace(b"IIII", b"III"))[1] == 4         # SMI_FQ_LENGTH_MISMATCH
    n, err, *_ = _index(gpu_ctx, ok + b"@r3\nACG\n")
    assert n == 2 and err == 8                                           # SMI_FQ_TRUNCATED


def test_fastq_to_scan_pipeline(pkg, synth, sor, gpu_ctx):
    """FASTQ text -> K-FQ -> K-PACK -> K-SCAN gives the oracle's scan of the same reads"""
    wl = synth.make_whitelist(20_000, seed=241)
    used = synth.pick_used(wl, 200, seed=242)
    n = 800
    reads = synth.gen_reads(n, used, seed=243)
    seqs, quals = zip(*(synth.materialize(reads, i) for i in range(n)))
    text = "".join(f"@r{i} x\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(zip(seqs, quals))).encode()
    n_rec, err, d_text, b = _index(gpu_ctx, text)
    assert (n_rec, err) == (n, 0)
    total = int(b["offsets"][n_rec].item())
    d_reads = torch.zeros(total, dtype=torch.uint8, device="cuda")
    gpu_ctx.fastq_gather_device(d_text, b["seq_start"], b["offsets"], n_rec, d_reads)
    d_ends = torch.zeros((28, 2 * n), dtype=torch.int32, device="cuda")
    d_len = torch.zeros(n, dtype=torch.int32, device="cuda")
    gpu_ctx.pack_ends_device(d_reads, None, b["offsets"], n, d_ends, d_len)
    d_out = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    gpu_ctx.scan_device(d_ends, d_len, n, gpu_ctx.scan_config(2), d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(pkg.SCAN_RESULT_DTYPE).reshape(-1)
    ra = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    st, exp = sor.scan_batch_3p(ra, None, offs, "CTTCCGATCT", n_threads=4)
    assert (st == 0).all() and (got["flags"].astype(np.uint64) == exp["flags"]).all()
    assert (got["adapter_end"] == exp["adapter_end"]).all() and (got["found"] == exp["adapter_found"]).all()
