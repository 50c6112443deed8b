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
                    noname_blank=False):
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
                rc, sc = sor.scan_read_3p(fs, fq, "CTTCCGATCT")
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
