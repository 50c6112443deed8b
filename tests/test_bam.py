"""BAM ingest (host C++: smi_bgzf_inflate, smi_bam_header, smi_bam_index_records) against the independent Python model of
the SAM specification in tests/bammodel.py; no GPU involved."""
import random
import struct

import numpy as np
import pytest

import bammodel


def _records(rng, n):
    recs, meta = [], []
    pos = 1000
    for i in range(n):
        ref = 0 if i < n // 2 else 2
        if i == n // 2:
            pos = 50
        pos += rng.randint(0, 300)
        L = rng.randint(1, 400)
        seq = "".join(rng.choice("ACGTN") for _ in range(L))
        cigar = [("S", rng.randint(1, 9))] * rng.randint(0, 1) + [("M", L)] + [("I", 2), ("D", 3), ("N", 700), ("=", 4), ("X", 1)][:rng.randint(0, 5)]
        aux = b"NMC\x03" + b"RGZgrp" + bytes([48 + i % 10]) + b"\0" if i % 3 else b""
        name = f"r{i}_FWD_PS={100 + i}_PE={120 + i}_AE={160 + i}_X=ACGT_Q=20.5_{i:x}"
        flag = rng.choice([0, 16, 4, 256, 2048 + 16])
        qual = bytes(rng.randint(0, 60) for _ in range(L)) if i % 4 else None
        recs.append(bammodel.bam_record(name, flag, ref if not flag & 4 else -1, pos if not flag & 4 else -1, rng.randint(0, 60),
                                        cigar if not flag & 4 else [], seq, qual, aux))
        meta.append((name, flag, cigar, seq))
    return recs, meta


def test_bgzf_and_record_index_equal_model(pkg):
    from sicelore_amd import lib as libmod

    rng = random.Random(5)
    recs, _ = _records(rng, 700)
    refs = [("chr1", 248956422), ("chrUn_KI270302v1", 2274), ("chr12", 133275309)]
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:248956422\n"
    raw = bammodel.bam_bytes(text, refs, recs)
    for block in (0xFF00, 777):  # many small blocks: records straddle block boundaries
        z = bammodel.bgzf_compress(raw, block=block)
        got, used = libmod.bgzf_inflate(np.frombuffer(z, dtype=np.uint8), n_threads=3)
        assert used == len(z) and got.tobytes() == raw == bammodel.bgzf_decompress(z)
    bam = np.frombuffer(raw, dtype=np.uint8)
    t, r, start = libmod.bam_header(bam)
    mt, mr, mrecs = bammodel.parse_bam(raw)
    assert t == text == mt and r == refs == mr
    idx, end = libmod.bam_index_records(bam, start, cap=len(recs) + 5)
    assert end == len(raw) and idx.size == len(mrecs) == 700
    for a, m in zip(idx, mrecs):
        assert (int(a["rec_off"]), int(a["rec_len"])) == (m["off"], m["length"])
        assert (int(a["ref_id"]), int(a["pos"]), int(a["flag"]), int(a["mapq"])) == (m["ref_id"], m["pos0"], m["flag"], m["mapq"])
        assert raw[int(a["name_off"]):int(a["name_off"]) + int(a["l_read_name"]) - 1].decode() == m["name"]
        cg = np.frombuffer(raw, dtype="<u4", count=int(a["n_cigar"]), offset=int(a["cigar_off"]))
        assert [(bammodel.CIGAR_OPS[v & 15], int(v >> 4)) for v in cg] == m["cigar"]
        assert int(a["l_seq"]) == len(m["seq"]) and raw[int(a["qual_off"]):int(a["qual_off"]) + int(a["l_seq"])] == m["qual"]
        assert raw[int(a["aux_off"]):int(a["aux_off"]) + int(a["aux_len"])] == m["aux"]
    # a capped call and a buffer that ends inside a record stop in front of it
    part, e2 = libmod.bam_index_records(bam, start, cap=10)
    assert part.size == 10 and e2 == int(idx[10]["rec_off"])
    cut = int(idx[20]["rec_off"]) + 17
    part, e3 = libmod.bam_index_records(bam[:cut], start, cap=100)
    assert part.size == 20 and e3 == int(idx[20]["rec_off"])


def test_bgzf_errors_and_partial_input(pkg):
    from sicelore_amd import lib as libmod

    raw = bytes(range(256)) * 300
    z = bytearray(bammodel.bgzf_compress(raw, block=5000))
    # a trailing incomplete block is not consumed
    got, used = libmod.bgzf_inflate(np.frombuffer(bytes(z[:-9]), dtype=np.uint8))
    assert used < len(z) - 9 and raw.startswith(got.tobytes()) and got.size >= len(raw) - 5000
    bad = bytearray(z)
    bad[40] ^= 0x55  # inside the first deflate stream: inflate error or CRC mismatch
    with pytest.raises(libmod.SmiError):
        libmod.bgzf_inflate(np.frombuffer(bytes(bad), dtype=np.uint8))
    with pytest.raises(libmod.SmiError):
        libmod.bgzf_inflate(np.frombuffer(b"\x1f\x8b\x08\x00" + bytes(30), dtype=np.uint8))  # plain gzip, no BC field
    with pytest.raises(libmod.SmiError):
        libmod.bam_header(np.frombuffer(b"BAM\2" + bytes(20), dtype=np.uint8))
    short = bammodel.bam_bytes("", [], []) + struct.pack("<I", 8) + bytes(8)
    with pytest.raises(libmod.SmiError):
        libmod.bam_index_records(np.frombuffer(short, dtype=np.uint8), 12, cap=4)


def test_scan_data_from_name_follows_reference_rules(pkg):
    import importlib

    au = importlib.import_module("sicelore_amd.assignumis")
    from sicelore_amd import lib as libmod

    nm = "a1b2_REV_PS=1361_PE=1390_AE=1430_T=44_bc=ACGTACGTACGTACGT_ed=1_ed_sec=2147483647_bcStart=1429_bcEnd=1414_rk=17_X=ACGTTTGACCA_Q=27.1_1z"
    d = au.scan_data_from_name(nm)
    assert d["reverse"] and (d["ps"], d["pe"], d["ae"], d["tso"]) == (1361, 1390, 1430, 44) and d["x"] == "ACGTTTGACCA"
    assert d["bc"] == dict(seq="ACGTACGTACGTACGT", ed=1, ed_sec=2147483647, start=1429, end=1414, rank=17) and abs(d["q"] - 27.1) < 1e-6
    assert au.scan_data_from_name(nm, bc_edit_limit=0)["bc"] is None          # -b 0: the ed=1 barcode is not taken (L450-456)
    assert au.scan_data_from_name("plain_read_name") is None                   # no _REV_ / _FWD_ (L412-418)
    with pytest.raises(libmod.SmiError):
        au.scan_data_from_name("x_FWD_PS=3_PE=9_")                             # AdapterInfoNotFoundInReadException (L442-443)
    assert au.parse_name(nm)["cell"] == "ACGTACGTACGTACGT" and au.parse_name("x_FWD_PS=3_PE=9_AE=40_X=AC_Q=9_1") is None


def test_gz_inflate_multi_member_and_errors(pkg):
    import gzip

    from sicelore_amd import lib as libmod

    a = (b"@r1\nACGT\n+\nIIII\n" * 30000)
    b = b"@r2\nTTTT\n+\n5555\n" * 7
    z = gzip.compress(a, 6) + gzip.compress(b"") + gzip.compress(b, 1)  # cat a.gz empty.gz b.gz
    got = libmod.gz_inflate(np.frombuffer(z, dtype=np.uint8))
    assert got.tobytes() == a + b
    assert libmod.gz_inflate(np.frombuffer(bammodel.bgzf_compress(a), dtype=np.uint8)).tobytes() == a  # BGZF is gzip too
    with pytest.raises(libmod.SmiError):
        libmod.gz_inflate(np.frombuffer(z[:-5], dtype=np.uint8))  # truncated
    bad = bytearray(z)
    bad[len(bad) // 3] ^= 0xFF
    with pytest.raises(libmod.SmiError):
        libmod.gz_inflate(np.frombuffer(bytes(bad), dtype=np.uint8))


def _parse_aux(aux):
    """independent reader of the aux types this project writes or passes through"""
    out, p = [], 0
    while p < len(aux):
        tag, ty = aux[p:p + 2].decode(), chr(aux[p + 2])
        p += 3
        if ty == "Z":
            q = aux.index(b"\0", p)
            out.append((tag, ty, aux[p:q].decode()))
            p = q + 1
        else:
            fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "A": "<c", "f": "<f"}[ty]
            out.append((tag, ty, struct.unpack_from(fmt, aux, p)[0]))
            p += struct.calcsize(fmt)
    return out


def test_tag_sets_and_attribute_order(pkg):
    import importlib

    au = importlib.import_module("sicelore_amd.assignumis")
    nm = "a_REV_PS=100_PE=130_AE=170_T=44_bc=ACGTACGTACGTACGT_ed=1_ed_sec=2147483647_bcStart=169_bcEnd=154_rk=17_X=ACGT_Q=27.1_1z"
    d = au.scan_data_from_name(nm)
    assert d["read_id"] == int("1z", 36)
    calls, has_bc, clustered = au.record_tag_sets(d, dict(U8="AAAACCCCGGGG", U7="AAAACCCCGGGT", U1=1, U2=3), "AAAACCCCGGGT")
    assert has_bc and clustered
    # ReadScanResult.writeSamFlags L205-237, writeBCSamFlags L254-279, ClusterOneBase L145-164 -- in call order
    assert calls == [("PE", 130), ("PS", 100), ("AE", 170), ("RE", ""), ("TE", 44), ("BU", "ACGTACGTACGTACGT"), ("BV", "169"),
                     ("BE", "154"), ("BW", 1), ("BX", "N.A."), ("SX", "71"), ("BH", "17"), ("BC", "ACGTACGTACGTACGT"), ("BZ", ""),
                     ("BB", "169"), ("BF", "154"), ("B1", 1), ("B2", "2147483647"), ("BZ", "ACGTACGTACGTACGT"), ("BH", "17"),
                     ("U8", "AAAACCCCGGGG"), ("U7", "AAAACCCCGGGT"), ("UC", ""), ("U1", "1"), ("U2", "3")]
    # no clustering result: U7 from the read, copied to U8 with UZ -- unless the group was skipped (DONT_ASSIGN_UMI)
    c2, _, cl2 = au.record_tag_sets(d, None, "TTTTGGGGCCCC")
    assert not cl2 and c2[-3:] == [("U7", "TTTTGGGGCCCC"), ("U8", "TTTTGGGGCCCC"), ("UZ", "")]
    c3, _, _ = au.record_tag_sets(d, dict(skipped=True), "TTTTGGGGCCCC")
    assert c3[-1] == ("U7", "TTTTGGGGCCCC")
    d0 = au.scan_data_from_name("b_FWD_PS=5_PE=9_AE=40_X=AC_Q=9_1")
    c4, has_bc4, _ = au.record_tag_sets(d0, None, None)
    assert not has_bc4 and c4 == [("PE", 9), ("PS", 5), ("AE", 40)]
    # attribute list: ordered by binary tag (second char major) from the moment htsjdk decodes it; a repeated tag keeps its last value; integers
    # read from the file are written back in the smallest type (AS:i:16 -> c)
    aux = b"NMC\x03" + b"ASi\x10\x00\x00\x00" + b"tpAP" + b"BZZold\0"
    fields = au.apply_tag_sets(au.split_aux(aux), calls)
    tags = [t for t, _ in fields]
    assert tags == sorted(set(tags), key=lambda t: (ord(t[1]) << 8) | ord(t[0])) and len(tags) == len(set(tags))
    assert tags[:6] == ["B1", "U1", "B2", "U2", "U7", "U8"] and tags[-2:] == ["BZ", "tp"] and set(tags) >= {"NM", "AS", "tp", "BZ", "PE"}
    parsed = {}
    for t, ty, v in _parse_aux(b"".join(r for _, r in fields)):
        assert t not in parsed
        parsed[t] = (ty, v)
    assert parsed["PE"] == ("C", 130) and parsed["AE"] == ("C", 170) and parsed["B1"] == ("c", 1) and parsed["TE"] == ("c", 44)
    assert parsed["BZ"] == ("Z", "ACGTACGTACGTACGT") and parsed["RE"] == ("Z", "") and parsed["NM"] == ("c", 3) and parsed["tp"] == ("A", b"P")
    assert parsed["AS"] == ("c", 16)
    # integer types follow BinaryTagCodec.getIntegerType
    for v, ty in ((-1, "c"), (127, "c"), (128, "C"), (255, "C"), (256, "s"), (-129, "s"), (32768, "S"), (65536, "i"), (-40000, "i"), (2 ** 31, "I")):
        assert chr(au._aux_bytes("XY", v)[2]) == ty


def test_tag_sets_equal_the_reference_calls(pkg):
    """the setAttribute calls of ReadScanResult.writeSamFlags + writeBCSamFlags and the U7 value of OneNanoporeResult.getPostBCUMIseq, executed
    from the reference's class files on 178 read names (tests/golden/ref_exec_samtags.json), against record_tag_sets / umi_window"""
    import importlib
    import json
    import os

    au = importlib.import_module("sicelore_amd.assignumis")
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_exec_samtags.json")))
    n = n_none = n_u7 = 0
    for sec in gold["sections"]:
        five = sec["five_prime"]
        for c in sec["cases"]:
            assert "throws" not in c
            d = au.scan_data_from_name(c["name"])
            if "calls" not in c:
                assert c["scan_data"] is None and d is None, c["name"]
                n_none += 1
                continue
            calls, has_bc, clustered = au.record_tag_sets(d, None, None)
            assert not clustered
            want = [(t, int(v) if ty == "int" else v) for t, v, ty in c["calls"]]
            assert calls == want and [type(v) for _, v in calls] == [type(v) for _, v in want], c["name"]
            assert has_bc == ("u7" in c)
            if has_bc:
                w = au.umi_window(d["x"], d["ae"], d["bc"]["end"], five) if d["bc"]["end"] is not None and d["x"] else None
                u7 = None if w is None else "".join("AGCT"[{1: 0, 2: 1, 4: 2, 8: 3}[x]] if x != 15 else "N" for x in w[1:13])
                assert u7 == c["u7"], c["name"]
                n_u7 += 1
            n += 1
    assert n >= 150 and n_none >= 6 and n_u7 >= 130


def test_batch_order_equals_the_coordinate_comparator(pkg):
    """the order in which a batch is written: a stable sort by htsjdk's SAMRecordCoordinateComparator, executed comparison by comparison
    (tests/golden/ref_exec_bamorder.json), against sorted(key=_coordinate_key)"""
    import importlib
    import json
    import os

    au = importlib.import_module("sicelore_amd.assignumis")
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_exec_bamorder.json")))
    n = 0
    for c in gold["sections"][0]["cases"]:
        recs = [dict(ref_id=r["ref"], pos=r["pos0"], flag=r["flag"], mapq=r["mapq"], next_ref_id=r["mate_ref"], next_pos=r["mate_pos0"], tlen=r["tlen"])
                for r in c["records"]]
        names = [r["name"] for r in c["records"]]
        got = sorted(range(len(recs)), key=lambda k: au._coordinate_key(recs[k], names[k]))
        assert got == c["order"]
        n += len(recs)
    assert n > 1000


def test_attribute_list_equals_htsjdk_executed(pkg):
    """BinaryTagCodec.readTags + SAMRecord.setAttribute executed on 60 records (tests/golden/ref_exec_auxorder.json): the tags of the written
    record, their order, values and type characters, against split_aux / apply_tag_sets"""
    import importlib
    import json
    import os

    au = importlib.import_module("sicelore_amd.assignumis")
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_exec_auxorder.json")))
    n = n_retyped = 0
    for c in gold["sections"][0]["cases"]:
        aux = bytes.fromhex(c["aux_hex"])
        fields = au.apply_tag_sets(au.split_aux(aux), [(t, v) for t, v in c["calls"]])
        got = [[t, (v.decode() if ty == "A" else v), ty] for t, ty, v in _parse_aux(b"".join(r for _, r in fields))]
        want = []
        for t, v, ty in c["final"]:
            if ty == "A":
                v = chr(v)
            elif ty == "f":
                v = float(np.float32(v))
            want.append([t, v, ty])
        assert got == want, (c["aux"], c["calls"])
        kinds = {t: k for t, k, _ in c["aux"]}
        n_retyped += sum(1 for t, _, ty in c["final"] if t in kinds and kinds[t] in "cCsSiI" and kinds[t] != ty and t not in dict(c["calls"]))
        n += len(want)
    assert n > 600 and n_retyped > 40


def _write_batch_case(libmod, au, seed, five_prime=False, truncate=False, bc_edit_limit=None, gene_tag="GE", carried=()):
    """a BAM of mixed records + synthetic clustering results -> (native bytes, Python-mirror bytes, gene-count texts of both)"""
    import json
    import os

    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sec = json.load(open(os.path.join(gold_dir, "ref_exec_samtags.json")))["sections"][1 if five_prime else 0]
    names_pool = [c["name"] for c in sec["cases"]]
    aux_pool = [bytes.fromhex(c["aux_hex"]) for c in json.load(open(os.path.join(gold_dir, "ref_exec_auxorder.json")))["sections"][0]["cases"]]
    rng = np.random.default_rng(seed)
    n = 400
    rows = []
    for k in range(n):
        nm = names_pool[int(rng.integers(0, len(names_pool)))]
        if rng.random() < 0.5:
            nm = nm.replace("read", f"r{int(rng.integers(0, 30))}x", 1)          # repeated names: ties down to flags / mate fields
        unm = rng.random() < 0.06
        ref = -1 if unm else int(rng.integers(0, 2))
        flag = int(rng.choice([0, 16, 256, 272, 2048, 2064])) | (4 if unm else 0)
        cig = [] if unm else ([("S", int(rng.integers(1, 400)))] if rng.random() < 0.5 else []) + [("M", int(rng.integers(20, 300)))] + \
            ([("H", int(rng.integers(100, 300)))] if rng.random() < 0.3 else [])
        L = sum(ln for op, ln in cig if op in "MIS=X") or 10
        aux = aux_pool[int(rng.integers(0, len(aux_pool)))]
        for t in carried:                  # attributes the input already carries (a gene tag of an earlier tool under GE / under the -g attribute)
            if rng.random() < 0.5 and t.encode() not in (aux[k2:k2 + 2] for k2 in range(len(aux))):
                aux += t.encode() + b"Z" + f"OLD{k % 7}".encode() + b"\0"
        rows.append((ref, -1 if unm else int(rng.integers(0, 40)), nm, flag, cig, L, aux, int(rng.choice([0, 20, 60]))))
    brecs = [bammodel.bam_record(nm, fl, ref, p0, mq, cg, "C" * L, aux=aux) for ref, p0, nm, fl, cg, L, aux, mq in rows]
    header = bammodel.bam_bytes("@HD\tVN:1.6\tSO:unsorted\n", [("chr1", 10 ** 6), ("chr2", 10 ** 6)], [])
    data = bammodel.bgzf_compress(header + b"".join(brecs), block=8192)
    _text, _refs, bam, recs = au.load_bam(data, n_threads=2)
    assert recs.size == n
    names = [au.read_name(bam, r) for r in recs]
    scans = [au.scan_data_from_name(nm, bc_edit_limit) for nm in names]
    tags = np.zeros(n, dtype=libmod.UMI_TAG_DTYPE)
    umis = [None] * n
    for i, d in enumerate(scans):
        u7 = None
        if d is not None and d["bc"] is not None and d["bc"]["seq"] is not None:
            tags[i]["flags"] |= libmod.UMI_HAS_BC
            if d["bc"]["end"] is not None and d["x"]:
                w = au.umi_window(d["x"], d["ae"], d["bc"]["end"], five_prime)
                u7 = None if w is None else "".join("AGCT"[{1: 0, 2: 1, 4: 2, 8: 3}[c]] if c != 15 else "N" for c in w[1:13])
        if u7 is not None:
            tags[i]["flags"] |= libmod.UMI_HAS_U7
            tags[i]["u7"] = u7.encode()
            kind = rng.random()
            if kind < 0.5:
                u8 = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 12))
                u2 = int(rng.integers(-1, 4))
                tags[i]["flags"] |= libmod.UMI_CLUSTERED
                tags[i]["u8"], tags[i]["u1"], tags[i]["u2"] = u8.encode(), int(rng.integers(0, 3)), u2
                umis[i] = dict(U8=u8, U7=u7, U1=int(tags[i]["u1"]), U2=None if u2 < 0 else u2)
            elif kind < 0.6:
                tags[i]["flags"] |= libmod.UMI_SKIPPED
                umis[i] = dict(skipped=True)
    gene_list = []
    for i in range(n):
        k = rng.random()
        gene_list.append((None, None, None) if k < 0.1 else (None, None, "INTERGENIC") if k < 0.3 else
                         (rng.choice(["G1", "G2,G3", "G4", ",", "G5,,"]), rng.choice(["+", "-", "+,-"]), "CODING"))
    off, buf = [0], b""
    for ge, gs, xf in gene_list:
        for v in (ge, gs, xf):
            buf += (v or "").encode()
            off.append(len(buf))
    gene_raw = (np.frombuffer(buf + b"\0", dtype=np.uint8).copy(), np.array(off, dtype=np.uint32))
    region = rng.integers(-1, 30, n).astype(np.int64)
    nth = (rng.random(n) < 0.15).astype(np.uint8)
    batch = rng.permutation(n).astype(np.int32)[: n - 7]                        # a batch is a subset, in whatever order it was collected
    gc1, gc2 = libmod.GeneCounts(), libmod.GeneCounts()
    bc, umi, order = libmod.bam_write_batch(bam, recs, batch, tags, gene=gene_raw, bc_edit_limit=bc_edit_limit, truncate_read_name=truncate,
                                            five_prime=five_prime, n_threads=3, gene_counts=gc1, region=region, nth_record=nth, gene_tag=gene_tag)
    exp_order = sorted(batch.tolist(), key=lambda k: au._coordinate_key(recs[k], names[k]))
    exp_bc, exp_umi, grows = [], [], []
    for i in exp_order:
        res = au.tagged_record(bam, recs[i], names[i], scans[i], umis[i], gene_list[i], five_prime, truncate, gene_tag)
        if res is None:
            continue
        exp_bc.append(res[0])
        if res[1]:
            exp_umi.append(res[0])
        row = au.gene_count_row(bam, recs[i], res[2], int(region[i]), int(nth[i]), gene_tag)
        if row is not None:
            grows.append(row)
    if grows:
        gc2.add(**au._count_columns(grows, five_prime))
    assert order.tolist() == exp_order
    return bc.tobytes(), b"".join(exp_bc), umi.tobytes(), b"".join(exp_umi), (gc1.genecounts_tsv(), gc1.umi_depths_tsv(), gc1.info()), \
        (gc2.genecounts_tsv(), gc2.umi_depths_tsv(), gc2.info())


@pytest.mark.parametrize("five_prime,truncate,limit", [(False, False, None), (True, False, None), (False, True, None), (False, False, 0)])
def test_native_batch_writer_equals_the_python_mirror(pkg, five_prime, truncate, limit):
    """smi_bam_write_batch (host threads) against tagged_record / _coordinate_key / gene_count_row, the mirror the reference-executed
    fixtures pin: same order, same bytes in both outputs, same gene-count tables"""
    import importlib

    from sicelore_amd import lib as libmod

    au = importlib.import_module("sicelore_amd.assignumis")
    for seed in (1, 2):
        bc, exp_bc, umi, exp_umi, g1, g2 = _write_batch_case(libmod, au, seed, five_prime, truncate, limit)
        assert bc == exp_bc and umi == exp_umi and len(bc) > 50_000 and 0 < len(umi) < len(bc)
        assert g1 == g2 and g1[2]["gene_entries"] > 5 and g1[2]["records_skipped_clipping"] > 0


def test_gene_attribute_other_than_ge(pkg):
    """`assignumis -g XG` (UmiFinderMain.java:L239-246 -> TagReadBase.TAG, OneNanoporeResult.java:L522): the gene name goes under XG and is counted from XG; GS and XF
    stay; a GE the input carries is neither touched nor counted.  Native writer == Python mirror, and against the default run only the tag differs."""
    import importlib

    from sicelore_amd import lib as libmod

    au = importlib.import_module("sicelore_amd.assignumis")
    bc, exp_bc, umi, exp_umi, g1, g2 = _write_batch_case(libmod, au, 5, gene_tag="XG", carried=("GE", "XG"))
    assert bc == exp_bc and umi == exp_umi and g1 == g2 and g1[2]["gene_entries"] > 5
    assert b"XGZG" in bc and b"GEZOLD" in bc and b"GEZG" not in bc and b"XGZOLD" in bc      # (XGZOLD: records without an XF keep what they carried)
    d_bc, d_exp, _u, _ue, d1, d2 = _write_batch_case(libmod, au, 5, carried=("GE", "XG"))
    assert d_bc == d_exp and d1 == d2 and b"GEZG" in d_bc and b"XGZOLD" in d_bc
    assert len(d_bc) != 0 and g1[0] != "" and sorted(g1[0].split("\n")[0].split("\t")) == sorted(d1[0].split("\n")[0].split("\t"))
    for bad in ("G", "G1", "GEN"):
        with pytest.raises(libmod.SmiError, match="two letters"):
            _write_batch_case(libmod, au, 5, gene_tag=bad)


def test_chunk_bounds_equal_the_reader_loop(pkg):
    """BamReader.run's cuts as positions (assignumis.chunk_bounds, what the native and the streamed pipeline use) against the record-by-record
    loop that mirrors the reference (assignumis._run_chunks): chunk sizes 1 - 11, up to four chromosomes, empty and one-record inputs"""
    import importlib

    au = importlib.import_module("sicelore_amd.assignumis")
    rng = np.random.default_rng(3)
    n_cases = 0
    for t in range(400):
        n = int(rng.integers(0, 70))
        ref = np.sort(rng.integers(0, 4, n))
        if t % 9 == 0 and n > 3:
            ref[-2:] = -1                                             # an unmapped tail is one more "chromosome"
        cs = int(rng.integers(1, 12))
        recs = np.zeros(n, dtype=[("ref_id", "<i4")])
        recs["ref_id"] = ref
        got = []

        def flush(cur, keep):
            got.append((cur[-1] + 1, keep))
            return []

        au._run_chunks(recs, cs, flush)
        assert (got[:-1] if n else got) == au.chunk_bounds(ref, cs), (t, n, cs)     # (the loop's last flush is the end of the file)
        n_cases += 1
    assert n_cases == 400


def test_segment_cuts_of_a_streamed_bam_equal_the_whole_file_cuts(pkg):
    """assignumis_stream's cut positions (segment_cuts, segment after segment, with whatever the grouping holds back carried over) against
    chunk_bounds on the whole file"""
    import importlib

    au = importlib.import_module("sicelore_amd.assignumis")
    rng = np.random.default_rng(4)
    for t in range(300):
        n = int(rng.integers(1, 120))
        ref = np.sort(rng.integers(0, 4, n)).astype(np.int64)
        step = int(rng.integers(1, 15))
        want = au.chunk_bounds(ref, step)
        got = []
        g0, i0, prev_ref, cur_n, pos = 0, -1, None, 0, 0          # pos: global index of the first record not yet read
        while pos < n or cur_n:
            take = int(rng.integers(0, 25))                       # records the next segment adds (0: a segment that ends inside a record)
            hi = min(n, pos + take)
            seg = ref[g0:hi]                                      # pending records + the new ones
            cuts, i0 = au.segment_cuts(seg, cur_n, g0, i0, prev_ref, step)
            k_first = 0                                           # local index of the first record that is still pending after the cuts
            for at, keep in cuts:
                got.append((g0 + at, keep))
                held = int(rng.integers(0, at - k_first + 1)) if keep else 0      # what the grouping holds back for the next chunk
                k_first = at - held
            if hi == n and pos >= n:
                break
            m = len(seg)
            if m:
                prev_ref = int(seg[m - 1])
            cur_n = m - k_first
            g0 += k_first
            pos = hi
            if pos >= n:                                          # the end of the file: what is pending is flushed, no further cut positions
                break
        assert got == want, (t, n, step, got, want)
