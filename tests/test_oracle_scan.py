"""CPU tests of the ORACLE for the 3' read scan: hand-derived vectors + C oracle == independent Python model."""
import random

import numpy as np
import pytest

import pymodel_scan as pms

AD10 = "CTTCCGATCT"              # Jar/config.xml:111 (pass 2)
AD22 = "CTACACGACGCTCTTCCGATCT"  # Jar/config.xml:113 (pass 1)


def test_needleman_hand_vectors(sor):
    # perfect match: 10 diagonal steps
    a1, d, a2, ne, ins, dl, sub, e5 = sor.nw_strings(AD10, AD10)
    assert (a1, d, a2) == (AD10, "." * 10, AD10) and ne == 0 and (ins, dl, sub) == (0, 0, 0) and e5 == 0
    # one substitution in the middle
    a1, d, a2, ne, ins, dl, sub, e5 = sor.nw_strings(AD10, "CTTCAGATCT")
    assert d == "....x....." and ne == 1 and (ins, dl, sub) == (0, 0, 1)
    # N in the read matches anything (IUPAC AND)
    assert sor.nw_strings(AD10, "CTTCNGATCT")[1] == "." * 10
    # substitution within the last two read bases weighs 1.2 in countIndelsMismatchesEndOfRead
    a1, d, a2, ne, ins, dl, sub, e5 = sor.nw_strings(AD10, "CTTCCGATCA")
    assert d == ".........x" and abs(e5 - 1.2) < 1e-6
    # C oracle strings == Python model strings on random slices
    rng = random.Random(3)
    for _ in range(300):
        ad = rng.choice([AD10, AD22])
        sl = "".join(rng.choice("ACGTN" if rng.random() < 0.1 else "ACGT") for _ in range(len(ad)))
        if rng.random() < 0.6:  # noisy copy of the adapter
            sl = list(ad)
            for _ in range(rng.randrange(4)):
                p = rng.randrange(len(sl))
                op = rng.choice("sid")
                if op == "s":
                    sl[p] = rng.choice("ACGT")
                elif op == "i":
                    sl.insert(p, rng.choice("ACGT"))
                else:
                    del sl[p]
            sl = ("".join(sl) + "ACGTACGT")[:len(ad)]
        a1, d, a2, ne, ins, dl, sub, e5 = sor.nw_strings(ad, sl)
        m = pms.needleman(pms.enc(ad), pms.enc(sl))
        assert (a1, d, a2) == m
        assert ne == float(pms.count_errors(m))
        nm = pms.NeedlemanMatch(m)
        assert (ins, dl, sub) == (nm.ins, nm.dele, nm.sub) and e5 == float(nm.end_of_read(5))


def test_polyt_finder(sor):
    rng = random.Random(11)
    hits = 0
    for it in range(400):
        n_pre = rng.randrange(0, 120)
        n_t = rng.randrange(5, 70)
        s = [rng.choice("ACGT") for _ in range(n_pre)] + ["T"] * n_t
        for _ in range(n_t // 8):  # impurities
            s[n_pre + rng.randrange(n_t)] = rng.choice("ACGN")
        s += [rng.choice("ACGT") for _ in range(200)]
        codes = pms.enc("".join(s[:175]))
        got = sor.find_polyt(codes)
        exp = pms.find_polyt(codes)
        assert got == exp, (it, "".join(s[:175]))
        hits += got is not None
    assert 100 < hits < 400
    # 11 of 15 is below 0.75, 12 of 15 passes; the window of entry `pos` is [pos+1, pos+15]
    base = "ACGACGACGA" + "T" * 4 + "A" + "T" * 4 + "A" + "T" * 4 + "ACG" * 60
    assert sor.find_polyt(pms.enc(base[:175])) == pms.find_polyt(pms.enc(base[:175]))


@pytest.mark.parametrize("adapter,n_reads", [(AD10, 250), (AD22, 250)])
def test_scan_c_oracle_equals_python_model(sor, synth, adapter, n_reads):
    wl = synth.make_whitelist(5000, seed=81)
    used = synth.pick_used(wl, 50, seed=82)
    reads = synth.gen_reads(n_reads, used, seed=83, n_rate=0.004)
    n_found = n_pass1 = n_tso = 0
    for i in range(n_reads):
        seq, qual = synth.materialize(reads, i)
        if i % 25 == 0:
            seq, qual = seq[:150], qual[:150]  # too short
        if i % 25 == 1:
            seq = seq[:300] + seq[-300:]       # adapter at both ends is impossible, but exercises short reads
            qual = qual[:300] + qual[-300:]
        rc, r = sor.scan_read_3p(seq, qual, adapter)
        m = pms.scan_read_3p(seq, qual, adapter)
        assert rc == 0
        flags = {k for k, b in sor.FLAG_BITS.items() if (int(r["flags"]) >> b) & 1}
        assert flags == m["flags"], (i, flags, m["flags"])
        assert int(r["adapter_found"]) == m["adapter_found"]
        assert (int(r["tso_start"]), int(r["tso_end"])) == (m["tso_start"], m["tso_end"]), i
        n_tso += bool(m["tso_start"] or m["tso_end"])
        if m["adapter_found"]:
            n_found += 1
            for f in ("adapter_start", "adapter_end", "polya_start", "polya_end", "reverse", "adapter_nmis", "pass1_ok"):
                assert int(r[f]) == m[f], (f, i)
            n_pass1 += m["pass1_ok"]
    assert n_found > n_reads * 0.6
    assert n_pass1 > 10
    assert n_tso > n_reads * 0.5


def test_scan_then_assign_recovers_barcodes(sor, synth):
    """end to end on the oracle: scan -> AE -> assignBarcode against the used list gives the planted barcode"""
    wl = synth.make_whitelist(20000, seed=91)
    used = synth.pick_used(wl, 100, seed=92)
    reads = synth.gen_reads(300, used, seed=93)
    bset = sor.BarcodeSet(used.numpy())
    ok = tot = 0
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    for i in range(300):
        seq, qual = synth.materialize(reads, i)
        rc, r = sor.scan_read_3p(seq, qual, AD10)
        if not r["adapter_found"]:
            continue
        stranded = "".join(comp[c] for c in reversed(seq)) if r["reverse"] else seq
        rc2, a = sor.assign_barcode(bset, stranded, int(r["adapter_end"]), max_ed=1)
        assert rc2 >= 0
        if rc2 == 1:
            tot += 1
            ok += int(a["bc"]) == int(reads["truth"][i])
    assert tot > 150 and ok / tot > 0.97


def _cmp_5p(sor, seq, qual, dont):
    rc, o = sor.scan_read_5p(seq, qual, AD10, max_mm=4, dont_search_polya=dont)
    m = pms.scan_read_5p(seq, qual, AD10, max_mm=4, dont_search_polya=dont)
    assert rc == 0
    flags = {k for k, b in sor.FLAG_BITS.items() if (int(o["flags"]) >> b) & 1}
    assert flags == m["flags"], (flags, m["flags"])
    assert int(o["adapter_found"]) == m["adapter_found"]
    assert (int(o["polya_start"]), int(o["polya_end"])) == (m["polya_start"], m["polya_end"])
    if m["adapter_found"]:
        assert (int(o["adapter_start"]), int(o["adapter_end"]), int(o["adapter_nmis"]), int(o["reverse"])) == \
            (m["adapter_start"], m["adapter_end"], m["adapter_nmis"], m["reverse"])
        assert int(o["pass1_ok"]) == m["pass1_ok"]
    return o


@pytest.mark.parametrize("dont", [True, False])
def test_scan_5p_c_oracle_equals_python_model(sor, synth, dont):
    wl = synth.make_whitelist(20000, seed=301)
    used = synth.pick_used(wl, 100, seed=302)
    reads = synth.gen_reads_5p(120, used, seed=303, n_rate=0.003)
    n_found = n_rev = n_right = 0
    for i in range(120):
        seq, qual = synth.materialize(reads, i)
        if i % 17 == 0:
            seq, qual = seq[:180], qual[:180]  # too short
        o = _cmp_5p(sor, seq, qual, dont)
        if o["adapter_found"]:
            n_found += 1
            n_rev += int(o["reverse"])
            n_right += int(o["reverse"]) == int(reads["reverse"][i])
            # the barcode sits right behind the adapter on the stranded read
            stranded = pms.revcomp_str(seq) if o["reverse"] else seq
            ae = int(o["adapter_end"])
            assert 10 <= ae <= 140 and len(stranded[ae:ae + 16]) == 16
    assert n_found > 80 and 20 < n_rev < n_found - 20 and n_right > 0.95 * n_found
