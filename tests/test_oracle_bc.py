"""CPU tests of the ORACLE for the barcode matcher: the reference's own known answers (README read-name
examples), hand-derived vectors for the primitives, and the C oracle against the independent Python model."""
import random

import numpy as np
import pytest

import pymodel as pm

# /root/reference/README.md:400 and :452 -- the only known-answer material the reference holds for this path.
# X= is stranded[AE-40 .. AE+2]; the examples fix window geometry, codec order, reverse complement and the
# bcStart/bcEnd arithmetic.
README_KATS = [
    dict(X="AAAAAAAAAAAATGGCGTGTATTGTCTTGGCACGATCGGAAGA", AE=619, bc="TCCGATCGTGCCAAGA", ed=0, bcStart=618, bcEnd=603),
    dict(X="AAAAAAAAAAACAAACCAAGTAACCAACCCAACCTCACTCAGA", AE=1327, bc="GAGTGAGGTTGGGTAG", ed=1, bcStart=1326,
         bcEnd=1311),
]


def _read_with_x(X, AE):
    return "C" * (AE - 41) + X + "TCGGAAGAGCGTCGTGTAG"


@pytest.mark.parametrize("kat", README_KATS)
def test_readme_known_answers(sor, kat):
    read = _read_with_x(kat["X"], kat["AE"])
    assert read[kat["AE"] - 41:kat["AE"] + 2] == kat["X"]
    bset = sor.BarcodeSet([sor.encode(kat["bc"]), sor.encode("ACGTACGTACGTACGT")])
    rc, r = sor.assign_barcode(bset, read, kat["AE"], max_ed=1)
    assert rc == 1 and r["found"] == 1
    assert sor.decode(int(r["bc"])) == kat["bc"]
    assert r["ed"] == kat["ed"] and r["ed_sec"] == 2147483647
    assert r["bc_start"] == kat["bcStart"] and r["bc_end"] == kat["bcEnd"]
    # 5 offsets x (1 + 48 substitutions + 60 insertions + 15 deletions) set probes
    assert r["n_probes"] == 620
    m = pm.assign_barcode({sor.encode(kat["bc"])}, read, kat["AE"], 1)
    assert m["found"] == 1 and m["bc"] == int(r["bc"]) and m["bc_start"] == kat["bcStart"] and m["bc_end"] == kat["bcEnd"]


def test_codec_vectors(sor):
    # A=0 G=1 C=2 T=3, first base most significant
    assert sor.encode("A" * 16) == 0
    assert sor.encode("T" * 16) == 0xFFFFFFFF
    assert sor.encode("AGCT") == 0b00011011
    assert sor.decode(0b00011011, 4) == "AGCT"
    assert sor.decode(sor.revcomp(sor.encode("AAGCT"), 5), 5) == "AGCTT"
    # a non-ACGT char ORs (long)-2: everything above it becomes 1-bits
    v = sor.encode("ACNGT") & 0xFFFFFFFFFFFFFFFF
    assert v == ((0xFFFFFFFFFFFFFFFE << 4) | 0b0111) & 0xFFFFFFFFFFFFFFFF
    # ... and reverseComplement reads only 2*len bits: bases before the N read as T, the N as C
    assert sor.decode(sor.revcomp(sor.encode("ACNGT"), 5), 5) == sor.decode(sor.revcomp(sor.encode("TTCGT"), 5), 5)
    assert sor.encode("ACNGT") == pm.to_signed(pm.encode("ACNGT"))


def test_mutate_vectors(sor):
    s = sor.encode("ACGTACGTACGTACGT")
    dec = lambda v: sor.decode(int(v))  # noqa: E731
    assert [dec(v) for v in sor.replace_deg(s, 0)] == [b + "CGTACGTACGTACGT" for b in "AGCT"]
    assert [dec(v) for v in sor.replace_deg(s, 15)] == ["ACGTACGTACGTACG" + b for b in "AGCT"]
    # "insertion" after position 3: keeps bases 0..3, new base at 4, rest shifted right, last base dropped
    assert [dec(v) for v in sor.insert_deg(s, 3)] == ["ACGT" + b + "ACGTACGTACG" for b in "AGCT"]
    # "deletion" of position 3: rest shifted left, the 4-bit post base appended (A=1,G=2,C=4,T=8; N=15 -> A)
    assert dec(sor.delete_byte(s, 8, 3)) == "ACGACGTACGTACGTT"
    assert dec(sor.delete_byte(s, 15, 3)) == "ACGACGTACGTACGTA"
    assert dec(sor.delete_byte(s, 4, 0)) == "CGTACGTACGTACGTC"
    # Java shift-count wrap at pos == len-2: (hash << 62) >>> 64 == hash << 62, so the dropped last base stays
    # in bits 62..63 and the value can only be a barcode when that base is A
    ins14 = sor.insert_deg(s, 14)
    assert all((int(v) >> 62) & 3 == 3 for v in ins14)  # last base T = 3 (sign bit included)
    s_a = sor.encode("ACGTACGTACGTACGA")
    assert [dec(v) for v in sor.insert_deg(s_a, 14)] == ["ACGTACGTACGTACG" + b for b in "AGCT"]
    for pos in range(15):
        assert list(sor.insert_deg(s, pos)) == [pm.to_signed(v) for v in pm.insert_deg(s, pos, 16)]
        assert sor.delete_byte(s, 2, pos) == pm.to_signed(pm.delete_byte(s, 2, pos, 16))
    for pos in range(16):
        assert list(sor.replace_deg(s, pos)) == [pm.to_signed(v) for v in pm.replace_deg(s, pos, 16)]


def test_enumeration_order_first_hit_wins(sor):
    w = "ACGTACGTACGTACGT"
    s = sor.encode(w)
    post = [1, 2, 4, 8, 1]
    # two different barcodes one substitution away: position 2 is enumerated before position 9
    b_early = sor.encode("ACTTACGTACGTACGT")
    b_late = sor.encode("ACGTACGTAAGTACGT")
    ms, probes = sor.bc_match(sor.BarcodeSet([b_early, b_late]), s, 1, post4=post)
    assert probes == 124
    assert len(ms) == 1 and ms[0]["matching_bc"] == b_early and ms[0]["ed"] == 1 and ms[0]["subs"] == 1
    # at one position: substitutions, then insertions (A,G,C,T), then the deletion
    b_ins = sor.encode("ACGGTACGTACGTACG")  # G inserted after position 2
    b_del = sor.encode("ACTACGTACGTACGTA")  # position 2 deleted, post[1]=A appended
    ms, _ = sor.bc_match(sor.BarcodeSet([b_ins, b_del]), s, 1, post4=post)
    assert len(ms) == 1 and ms[0]["matching_bc"] == b_ins
    assert (ms[0]["ins"], ms[0]["dels"]) == (0, 1)  # insertions() bumps nDeletions (BarcodeMatchTester.java:L289)
    ms, _ = sor.bc_match(sor.BarcodeSet([b_del]), s, 1, post4=post)
    assert len(ms) == 1 and (ms[0]["ins"], ms[0]["dels"]) == (1, 0)  # deletions() bumps nInsertions (L346)
    # exact hit and a level-1 hit are both kept
    ms, _ = sor.bc_match(sor.BarcodeSet([s, b_late]), s, 1, post4=post)
    assert sorted(int(m["ed"]) for m in ms) == [0, 1]


def test_best_second_rule(sor):
    # read region: ...UMI rc, BC rc, then the adapter (rc) starts at AE
    bc = "TCCGATCGTGCCAAGA"
    rc_bc = sor.decode(sor.revcomp(sor.encode(bc)))
    read = "C" * 60 + "A" * 12 + "GGTTGGTTGGTT" + rc_bc + "AGATCGGAAGAGCGTCGTGTAG"
    AE = 60 + 12 + 12 + 16 + 1
    other = "GCCGATCGTGCCAAGA"  # one substitution (position 0) away from bc
    # only `bc` -> accepted with ed 0
    rc, r = sor.assign_barcode(sor.BarcodeSet([sor.encode(bc)]), read, AE)
    assert rc == 1 and r["found"] == 1 and r["ed"] == 0 and r["offset"] == 0
    # bc (ed 0) and a second barcode at ed 1 -> accepted, ed_sec = 1
    rc, r = sor.assign_barcode(sor.BarcodeSet([sor.encode(bc), sor.encode(other)]), read, AE)
    assert rc == 1 and r["found"] == 1 and r["ed"] == 0 and r["ed_sec"] == 1
    # a late neighbour is shadowed: bc ends with A, so "insert A after position 14" regenerates bc itself and is
    # the first level-1 hit; the substitution at position 15 is never recorded (first hit per level wins)
    late = "TCCGATCGTGCCAAGT"
    rc, r = sor.assign_barcode(sor.BarcodeSet([sor.encode(bc), sor.encode(late)]), read, AE)
    assert rc == 1 and r["ed"] == 0 and r["ed_sec"] == 2147483647 and r["n_matches"] == 2
    # only neighbours at ed 1, two different ones -> whatever the model says
    n1 = "ACCGATCGTGCCAAGA"
    rc, r = sor.assign_barcode(sor.BarcodeSet([sor.encode(other), sor.encode(n1)]), read, AE)
    assert rc == 1 or rc == 0
    m = pm.assign_barcode({sor.encode(other), sor.encode(n1)}, read, AE)
    assert int(r["found"]) == m["found"]
    # window not inside the read -> the reference throws (String.substring)
    rc, _ = sor.assign_barcode(sor.BarcodeSet([sor.encode(bc)]), read, 20)
    assert rc == -1
    assert pm.assign_barcode({sor.encode(bc)}, read, 20) is None


def _rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def _mutate_str(rng, s, k):
    s = list(s)
    for _ in range(k):
        op = rng.choice("sid")
        p = rng.randrange(len(s))
        if op == "s":
            s[p] = rng.choice("ACGT")
        elif op == "i":
            s.insert(p, rng.choice("ACGT"))
        elif len(s) > 1:
            del s[p]
    return "".join(s)


@pytest.mark.parametrize("ed,five_prime", [(1, False), (1, True), (2, False), (2, True), (0, False)])
def test_c_oracle_equals_python_model(sor, ed, five_prime):
    rng = random.Random(1000 + ed * 2 + five_prime)
    n_bc = 40
    bcs = [_rand_seq(rng, 16) for _ in range(n_bc)]
    # near-duplicates make second-best / ambiguity paths fire
    bcs += [_mutate_str(rng, b, 1)[:16].ljust(16, "A") for b in bcs[:10]]
    keys = [sor.encode(b) for b in bcs]
    bset = sor.BarcodeSet(keys)
    pset = set(k & pm.M64 for k in keys)
    n_reads = 60 if ed < 2 else 25
    n_found = 0
    for i in range(n_reads):
        bc = rng.choice(bcs)
        noisy = _mutate_str(rng, bc, rng.choice([0, 0, 1, 1, 2, 3]))
        alphabet = "ACGT" if i % 7 else "ACGTN"
        if not five_prime:
            rc_noisy = sor.decode(sor.revcomp(sor.encode(noisy.ljust(16, "A")[:len(noisy)]), len(noisy)), len(noisy)) \
                if "N" not in noisy else noisy
            read = _rand_seq(rng, 40, alphabet) + "A" * 10 + _rand_seq(rng, 12, alphabet) + rc_noisy + \
                "AGATCGGAAGAGCGTCGTGTAG"
            ae = 40 + 10 + 12 + len(noisy) + 1 + rng.choice([0, 0, 0, -1, 1])
        else:
            read = _rand_seq(rng, 4, alphabet) + "CTACACGACGCTCTTCCGATCT" + noisy + _rand_seq(rng, 40, alphabet)
            ae = 4 + 22 + rng.choice([0, 0, 0, -1, 1])
        rc, r = sor.assign_barcode(bset, read, ae, max_ed=ed, five_prime=five_prime)
        m = pm.assign_barcode(pset, read, ae, max_ed=ed, five_prime=five_prime)
        if m is None:
            assert rc == -1
            continue
        assert rc == m["found"], (i, read, ae)
        if m["found"]:
            n_found += 1
            for f in ("bc", "ed", "ed_sec", "offset", "ins_minus_del", "bc_start", "bc_end"):
                assert int(r[f]) == m[f], (f, i, read, ae)
    assert n_found > n_reads // 4


def test_match_level_sets_equal_python_model(sor):
    """per-offset Matches sets (incl. HashSet iteration order) for assign and collision modes, ed 1 and 2"""
    rng = random.Random(7)
    bcs = [_rand_seq(rng, 16) for _ in range(30)]
    bcs += [_mutate_str(rng, b, 1)[:16].ljust(16, "C") for b in bcs]
    bcs += [_mutate_str(rng, b, 2)[:16].ljust(16, "G") for b in bcs[:30]]
    keys = [sor.encode(b) for b in bcs]
    bset, pset = sor.BarcodeSet(keys), set(keys)
    for ed in (1, 2):
        for mode in ("assign", "collision"):
            for b in bcs[:12]:
                s = sor.encode(b)
                if mode == "assign":
                    post = [rng.choice([1, 2, 4, 8, 15]) for _ in range(5)]
                    ms, probes = sor.bc_match(bset, s, ed, post4=post, offset=1)
                    pmatches, pprobes = pm.bc_match(pset, s, 16, ed, False, True, post, 1, True)
                else:
                    # BarcodeDatasetColissionTester.java:L213-225: skipFullMatches, no post, no descent on hit
                    ms, probes = sor.bc_match(bset, s, ed, post4=None, skip_full=True, do_next=False)
                    pmatches, pprobes = pm.bc_match(pset, s, 16, ed, True, True, None, 0, False)
                assert probes == pprobes
                assert [(int(m["matching_bc"]), int(m["ed"]), int(m["subs"]), int(m["ins"]), int(m["dels"])) for m in ms] == \
                    [(pm.to_signed(m["matching_bc"]), m["ed"], m["subs"], m["ins"], m["dels"]) for m in pmatches]


def test_hashset_tie_order(sor):
    """equal ed at two non-zero offsets: the winner is decided by HashSet bucket order of the window hashes"""
    rng = random.Random(99)
    hits = 0
    for _ in range(300):
        core = _rand_seq(rng, 16)
        # homopolymer-free flanks; the same barcode reachable at offsets -1 and +1 only
        left, right = _rand_seq(rng, 30), "AGATCGGAAGAGCGTCGTGTAG"
        read = left + core + right
        ae = 30 + 16 + 1
        w_m1 = sor.decode(sor.revcomp(sor.encode(read[ae - 16 - 1 - 1:ae - 1 - 1])))
        w_p1 = sor.decode(sor.revcomp(sor.encode(read[ae - 16 + 1 - 1:ae - 1 + 1])))
        keys = [sor.encode(w_m1), sor.encode(w_p1)]
        rc, r = sor.assign_barcode(sor.BarcodeSet(keys), read, ae)
        m = pm.assign_barcode(set(keys), read, ae)
        assert rc == m["found"]
        if rc == 1:
            hits += 1
            assert int(r["bc"]) == m["bc"] and int(r["offset"]) == m["offset"]
    # two different barcodes at the same ed are ambiguous -> almost always rejected; the assertion above is the point
    assert hits >= 0


def test_golden_fixture(sor):
    """tests/golden/bc_assign_v1.json (made by tests/golden/make_golden.py from the oracle; regression pin)"""
    import json
    import os

    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bc_assign_v1.json")))
    bset = sor.BarcodeSet(g["whitelist"])
    for c in g["cases"]:
        for i, read in enumerate(c["reads"]):
            rc, r = sor.assign_barcode(bset, read, c["ae"][i], max_ed=c["max_ed"], five_prime=c["five_prime"])
            assert rc == c["status"][i]
            if rc == 1:
                for f in ("bc", "ed", "ed_sec", "offset", "ins_minus_del", "bc_start", "bc_end"):
                    assert int(r[f]) == c[f][i]
