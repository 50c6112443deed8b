"""Read-name writer (FastqRecordExt.getRecordForWriting): product (host C++) == oracle == Python model, and the two
read-name examples of the reference's README reproduce character for character."""
import decimal
import random

import numpy as np
import pytest

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def model_dec1(f):
    """DecimalFormat("##.#") on the exact value of the float, HALF_EVEN"""
    d = decimal.Decimal(float(np.float32(f))).quantize(decimal.Decimal("0.1"), rounding=decimal.ROUND_HALF_EVEN)
    s = format(d, "f")
    ip, fp = s.split(".")
    if fp == "0":
        return ip.lstrip("-") if d == 0 else ip
    return (ip if ip not in ("0", "-0") else ip.replace("0", "")) + "." + fp


def model_name(name, seq, qual, flags, ps, pe, ae, tso_end, bc, rank, read_id, five_prime=False):
    base = name.split(" ")[0]
    rev, fwd = "PASSED_REV" in flags, "PASSED_FWD" in flags
    if not (rev or fwd):
        return base + "_FAILED "
    add = "_REV_" if rev else "_FWD_"
    if pe:
        add += f"PS={ps}_PE={pe}_"
    if ae:
        add += f"AE={ae}_"
    if tso_end:
        add += f"T={tso_end}_"
    if bc:
        add += f"bc={bc['seq']}_ed={bc['ed']}_ed_sec={bc['ed_sec']}_bcStart={bc['start']}_bcEnd={bc['end']}_"
        if rank > 0:
            add += f"rk={rank}_"
    if ae:
        begin, end = (ae - 3, ae + 39) if five_prime else (ae - 41, ae + 2)
        if begin < 0:
            return base
        stranded = "".join(COMP[c] for c in reversed(seq)) if rev else seq
        squal = qual[::-1] if rev else qual
        add += "X=" + stranded[begin:end] + "_"
        q = np.float32(np.mean([ord(c) - 33 for c in squal[begin - 1:end]], dtype=np.float64))
        digits = "0123456789abcdefghijklmnopqrstuvwxyz"
        rid, s36 = read_id, ""
        while True:
            s36 = digits[rid % 36] + s36
            rid //= 36
            if rid == 0:
                break
        add += "Q=" + model_dec1(q) + "_" + s36
        if bc:
            add += " cellBC=" + bc["seq"]
    return base + add


def test_decimal_format(sor):
    for v, exp in [(27.1, "27.1"), (27.0, "27"), (27.25, "27.2"), (27.75, "27.8"), (0.04, "0"), (0.5, ".5"),
                   (15.94, "15.9"), (9.96, "10"), (12.35, "12.4"), (12.45, "12.4"), (0.0, "0")]:
        assert sor.fmt_dec1(v) == exp == model_dec1(v), v
    rng = random.Random(1)
    for _ in range(2000):
        v = rng.randrange(0, 44 * 41) / 44.0  # means of 44 qualities
        assert sor.fmt_dec1(v) == model_dec1(v)


def test_readme_read_name_examples(pkg, sor):
    """/root/reference/README.md:400 and :452"""
    from sicelore_amd import lib as libmod

    cases = [
        dict(rev=False, PS=566, PE=590, AE=619, T=0, bc="TCCGATCGTGCCAAGA", ed=0, offset=0, imd=0, rk=2987,
             X="AAAAAAAAAAAATGGCGTGTATTGTCTTGGCACGATCGGAAGA", qsum=1192,
             suffix="_FWD_PS=566_PE=590_AE=619_bc=TCCGATCGTGCCAAGA_ed=0_ed_sec=2147483647_bcStart=618_bcEnd=603_rk=2987_"
                    "X=AAAAAAAAAAAATGGCGTGTATTGTCTTGGCACGATCGGAAGA_Q=27.1"),
        dict(rev=True, PS=1257, PE=1305, AE=1327, T=40, bc="GAGTGAGGTTGGGTAG", ed=1, offset=0, imd=0, rk=3883,
             X="AAAAAAAAAAACAAACCAAGTAACCAACCCAACCTCACTCAGA", qsum=700,
             suffix="_REV_PS=1257_PE=1305_AE=1327_T=40_bc=GAGTGAGGTTGGGTAG_ed=1_ed_sec=2147483647_bcStart=1326_bcEnd=1311_"
                    "rk=3883_X=AAAAAAAAAAACAAACCAAGTAACCAACCCAACCTCACTCAGA_Q=15.9"),
    ]
    for c in cases:
        stranded = "C" * (c["AE"] - 41) + c["X"] + "TCGGAAGAGCGTCGTGTAG"
        squal = ["5"] * len(stranded)
        # 44 qualities over stranded[AE-41 .. AE+2] (1-based) with the wanted sum
        lo = c["AE"] - 42
        vals = [c["qsum"] // 44] * 44
        for k in range(c["qsum"] - sum(vals)):
            vals[k] += 1
        for k, v in enumerate(vals):
            squal[lo + k] = chr(33 + v)
        squal = "".join(squal)
        raw_seq = "".join(COMP[x] for x in reversed(stranded)) if c["rev"] else stranded
        raw_qual = squal[::-1] if c["rev"] else squal
        scan = np.zeros(1, dtype=pkg.SCAN_RESULT_DTYPE)[0]
        scan["flags"] = (1 << sor.FLAG_BITS["PASSED_REV"]) if c["rev"] else (1 << sor.FLAG_BITS["PASSED_FWD"])
        scan["polya_start"], scan["polya_end"], scan["adapter_end"], scan["found"] = c["PS"], c["PE"], c["AE"], 1
        scan["reverse"], scan["tso_end"] = int(c["rev"]), c["T"]
        bc = np.zeros(1, dtype=pkg.BC_RESULT_DTYPE)[0]
        bc["bc"], bc["found"], bc["ed"], bc["ed_sec"] = sor.encode(c["bc"]), 1, c["ed"], 2147483647
        bc["offset"], bc["ins_minus_del"] = c["offset"], c["imd"]
        got = libmod.format_read_name("read1 runid=abc", raw_seq, raw_qual, scan, bc, rank=c["rk"], read_id=46655)
        assert got == "read1" + c["suffix"] + "_zzz cellBC=" + c["bc"]


def test_product_oracle_model_on_scanned_reads(pkg, sor, synth):
    from sicelore_amd import lib as libmod

    wl = synth.make_whitelist(20000, seed=401)
    used = synth.pick_used(wl, 100, seed=402)
    reads = synth.gen_reads(250, used, seed=403)
    bset = sor.BarcodeSet(used.numpy())
    n_bc = n_failed = 0
    for i in range(250):
        seq, qual = synth.materialize(reads, i)
        if i % 31 == 0:
            seq, qual = seq[:150], qual[:150]
        rc, sc = sor.scan_read_3p(seq, qual, "CTTCCGATCT")
        flags = {k for k, b in sor.FLAG_BITS.items() if (int(sc["flags"]) >> b) & 1}
        a = None
        if sc["adapter_found"]:
            stranded = "".join(COMP[c] for c in reversed(seq)) if sc["reverse"] else seq
            rc2, a_ = sor.assign_barcode(bset, stranded, int(sc["adapter_end"]), max_ed=1)
            if rc2 == 1:
                a = a_
        name = f"r{i} ch=5 start_time=2024"
        rank = (i % 50) + 1 if i % 3 else 0
        o = sor.format_read_name(name, seq, qual, sc, a, rank=rank, read_id=1000 + i)
        # product structs from the oracle records
        ps = np.zeros(1, dtype=pkg.SCAN_RESULT_DTYPE)[0]
        ps["flags"] = int(sc["flags"]) & 0xFFFFFFFF
        for f in ("polya_start", "polya_end", "adapter_start", "adapter_end", "scan_end", "adapter_nmis", "reverse",
                  "tso_start", "tso_end"):
            ps[f] = sc[f]
        ps["found"] = sc["adapter_found"]
        pb = None
        mb = None
        if a is not None:
            pb = np.zeros(1, dtype=pkg.BC_RESULT_DTYPE)[0]
            pb["bc"], pb["found"], pb["ed"], pb["ed_sec"] = int(a["bc"]) & 0xFFFFFFFF, 1, a["ed"], a["ed_sec"]
            pb["offset"], pb["ins_minus_del"] = a["offset"], a["ins_minus_del"]
            mb = dict(seq=sor.decode(int(a["bc"])), ed=int(a["ed"]), ed_sec=int(a["ed_sec"]), start=int(a["bc_start"]),
                      end=int(a["bc_end"]))
            n_bc += 1
        p = libmod.format_read_name(name, seq, qual, ps, pb, rank=rank, read_id=1000 + i)
        m = model_name(name, seq, qual, flags, int(sc["polya_start"]), int(sc["polya_end"]),
                       int(sc["adapter_end"]) if sc["adapter_found"] else 0, int(sc["tso_end"]), mb, rank, 1000 + i)
        assert o == p == m, (i, o, p, m)
        n_failed += o.endswith("_FAILED ")
    assert n_bc > 100 and n_failed > 5


def test_product_oracle_model_5p(pkg, sor, synth):
    from sicelore_amd import lib as libmod

    wl = synth.make_whitelist(20000, seed=411)
    used = synth.pick_used(wl, 100, seed=412)
    reads = synth.gen_reads_5p(200, used, seed=413)
    bset = sor.BarcodeSet(used.numpy())
    n_bc = 0
    for i in range(200):
        seq, qual = synth.materialize(reads, i)
        rc, sc = sor.scan_read_5p(seq, qual, "CTTCCGATCT", max_mm=4, dont_search_polya=bool(i % 2))
        if rc != 0:
            continue
        flags = {k for k, b in sor.FLAG_BITS.items() if (int(sc["flags"]) >> b) & 1}
        a = None
        if sc["adapter_found"]:
            stranded = "".join(COMP[c] for c in reversed(seq)) if sc["reverse"] else seq
            rc2, a_ = sor.assign_barcode(bset, stranded, int(sc["adapter_end"]), max_ed=1, five_prime=True)
            if rc2 == 1:
                a = a_
        name = f"r{i} ch=5"
        o = sor.format_read_name(name, seq, qual, sc, a, rank=i % 7, read_id=50000 + i, five_prime=True)
        ps = np.zeros(1, dtype=pkg.SCAN_RESULT_DTYPE)[0]
        ps["flags"] = int(sc["flags"]) & 0xFFFFFFFF
        for f in ("polya_start", "polya_end", "adapter_start", "adapter_end", "scan_end", "adapter_nmis", "reverse",
                  "tso_start", "tso_end"):
            ps[f] = sc[f]
        ps["found"] = sc["adapter_found"]
        pb = mb = None
        if a is not None:
            pb = np.zeros(1, dtype=pkg.BC_RESULT_DTYPE)[0]
            pb["bc"], pb["found"], pb["ed"], pb["ed_sec"] = int(a["bc"]) & 0xFFFFFFFF, 1, a["ed"], a["ed_sec"]
            pb["offset"], pb["ins_minus_del"] = a["offset"], a["ins_minus_del"]
            mb = dict(seq=sor.decode(int(a["bc"])), ed=int(a["ed"]), ed_sec=int(a["ed_sec"]), start=int(a["bc_start"]),
                      end=int(a["bc_end"]))
            n_bc += 1
        p = libmod.format_read_name(name, seq, qual, ps, pb, rank=i % 7, read_id=50000 + i, five_prime=True)
        m = model_name(name, seq, qual, flags, int(sc["polya_start"]), int(sc["polya_end"]),
                       int(sc["adapter_end"]) if sc["adapter_found"] else 0, 0, mb, i % 7, 50000 + i, five_prime=True)
        assert o == p == m, (i, o, p, m)
    assert n_bc > 80


def model_record(name, qhdr, seq, qual, flags, ps, pe, ae, tso_end, bc, rank, read_id, five_prime=False, trim=False):
    """getRecordForWriting L209-311 + htsjdk BasicFastqWriter, in Python on top of model_name"""
    rev, fwd = "PASSED_REV" in flags, "PASSED_FWD" in flags
    nm = model_name(name, seq, qual, flags, ps, pe, ae, tso_end, bc, rank, read_id, five_prime)
    if not (rev or fwd):
        return f"@{nm}\n{seq}\n+{qhdr}\n{qual}\n", False
    s = "".join(COMP[c] for c in reversed(seq)) if rev else seq
    begin = (ae - 3) if five_prime else (ae - 41)
    q = (qual[::-1] if rev else qual) if (ae and begin >= 0) else None
    if trim and bc:
        b = bc["start"] + 30 if five_prime else (tso_end or 1)
        e = ps if pe else len(seq)
        if b < e:
            s, q = s[b - 1:e], q[b - 1:e]
    return f"@{nm}\n{s}\n+{qhdr}\n{'null' if q is None else q}\n", True


@pytest.mark.parametrize("trim", [False, True])
def test_oracle_record_equals_model(sor, synth, trim):
    wl = synth.make_whitelist(20000, seed=421)
    used = synth.pick_used(wl, 100, seed=422)
    reads = synth.gen_reads(150, used, seed=423, n_rate=0.002)
    bset = sor.BarcodeSet(used.numpy())
    n_passed = n_trimmed = 0
    for i in range(150):
        seq, qual = synth.materialize(reads, i)
        if i % 50 == 7:
            seq, qual = seq[:150], qual[:150]  # too short: FAILED
        rc, sc = sor.scan_read_3p(seq, qual, "CTTCCGATCT")
        assert rc == 0
        flags = {k for k, b in sor.FLAG_BITS.items() if (int(sc["flags"]) >> b) & 1}
        a = bc = None
        if sc["adapter_found"]:
            stranded = "".join(COMP[c] for c in reversed(seq)) if sc["reverse"] else seq
            rc2, a_ = sor.assign_barcode(bset, stranded, int(sc["adapter_end"]), max_ed=1)
            if rc2 == 1:
                a = a_
                bc = dict(seq=sor.decode(int(a["bc"]) & 0xFFFFFFFF, 16), ed=int(a["ed"]), ed_sec=int(a["ed_sec"]),
                          start=int(a["bc_start"]), end=int(a["bc_end"]))
        name, qh = f"r{i} ch=5", ("" if i % 3 else f"r{i}")
        got, ok = sor.fastq_record(name, qh, seq, qual, sc, a, rank=i % 5, read_id=1000 + i, trim_fastq=trim)
        exp, ok_m = model_record(name, qh, seq, qual, flags, int(sc["polya_start"]), int(sc["polya_end"]),
                                 int(sc["adapter_end"]) if sc["adapter_found"] else 0, int(sc["tso_end"]), bc, i % 5, 1000 + i,
                                 trim=trim)
        assert got == exp.encode() and ok == ok_m
        n_passed += ok
        n_trimmed += ok and len(got) < 2 * len(seq)
    assert n_passed > 120 and (n_trimmed > 40) == trim
    # a read marked MULTI_CHIMERIC_READS_DISCARDED is written as FAILED with its raw bases whatever its scan says
    got, ok = sor.fastq_record("x y", "", seq, qual, sc, a, force_failed=True)
    assert not ok and got == f"@x_FAILED \n{seq}\n+\n{qual}\n".encode()
