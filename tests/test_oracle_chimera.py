"""Chimera splitter (ChimeraFindernew.findSplitPositions): oracle == independent Python model; hand-built reads."""
import random

import numpy as np

import pytest

import pymodel_chimera as pm

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
TSO30 = "AAGCAGTGGTATCAACGCAGAGTACATGGG"
AD22 = "CTACACGACGCTCTTCCGATCT"


def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def rnd(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def molecule(rng, cdna=500, polya=30):
    """transcript-sense 3' read: TSO cDNA polyA rc(UMI) rc(BC) rc(adapter)"""
    return TSO30 + rnd(rng, cdna) + "A" * polya + rnd(rng, 12) + rnd(rng, 16) + rc(AD22)


def check(sor, read):
    rc_, splits, multi, n_matches, raw = sor.chimera_split(read)
    m_splits, m_multi, m_ms = pm.find_split_positions(read)
    assert rc_ == 0
    assert splits == m_splits and multi == m_multi and n_matches == len(m_ms), (splits, m_splits, multi, m_multi)
    return splits, multi, raw


def test_short_and_clean_reads_are_not_split(sor):
    rng = random.Random(5)
    assert check(sor, rnd(rng, 239))[0] == []
    assert check(sor, molecule(rng))[0] == []
    assert check(sor, rc(molecule(rng)))[0] == []


def test_two_molecules_head_to_tail(sor):
    rng = random.Random(6)
    a, b = molecule(rng), molecule(rng)
    splits, multi, raw = check(sor, a + b)
    # rc(adapter) of molecule 1 (a REVERSE adapter after the polyA) directly followed by the TSO of molecule 2
    assert not multi and len(splits) == 1 and splits[0][0] == "REV_ADAPTER_FWD_TSO"
    assert abs(splits[0][1] - len(a)) <= 15
    names = [sor.chimera_fragment_name("r1 runid=x ch=5", raw, k) for k in range(2)]
    assert names == ["r1_RA_FTsp1 runid=x ch=5", "r1_RA_FTsp2 runid=x ch=5"]
    assert [f[0] for f in pm.fragments("r1 runid=x ch=5", a + b, "#" * len(a + b), splits)] == names
    assert sor.chimera_fragment_name("noblank", raw, 1) == "noblank"


def test_isolated_internal_adapter(sor):
    rng = random.Random(7)
    # forward adapter + BC + UMI + polyT in the middle of a read, no TSO partner
    core = AD22 + rnd(rng, 16) + rnd(rng, 12) + "T" * 30
    read = rnd(rng, 600) + core + rnd(rng, 600)
    splits, multi, _ = check(sor, read)
    assert splits == [("FWD_ADAPTER", 601 - 25)] and not multi
    splits, multi, _ = check(sor, rc(read))
    assert [s[0] for s in splits] == ["REV_ADAPTER"] and not multi


def test_four_molecules_are_discarded(sor):
    rng = random.Random(8)
    read = "".join(molecule(rng) for _ in range(4))
    splits, multi, _ = check(sor, read)
    assert multi and splits == []


def test_random_chimeras_oracle_equals_model(sor, synth):
    wl = synth.make_whitelist(5000, seed=71)
    used = synth.pick_used(wl, 50, seed=72)
    reads = synth.gen_reads(60, used, seed=73, max_mid=400)
    chim = synth.make_chimeras(reads, 60, seed=74)
    n_split = n_multi = 0
    for seq, _q, _k in chim:
        splits, multi, _ = check(sor, seq)
        n_split += len(splits) > 0
        n_multi += multi
    assert n_split >= 15 and n_multi >= 1


def test_product_fragment_names(pkg, sor):
    from sicelore_amd import lib as libmod

    for reasons in ([3], [0, 5], [4, 1]):
        r = np.zeros(1, dtype=pkg.CHIMERA_RESULT_DTYPE)[0]
        raw = sor._ChimeraResult()
        raw.n_split = r["n_split"] = len(reasons)
        for k, x in enumerate(reasons):
            r["reason"][k] = raw.reason[k] = x
            r["pos"][k] = raw.pos[k] = 300 * (k + 1)
        for name in ("r7 runid=abc ch=1", "plain"):
            for k in range(len(reasons) + 1):
                assert libmod.chimera_fragment_name(name, r, k) == sor.chimera_fragment_name(name, raw, k)
    with pytest.raises(libmod.SmiError):
        libmod.chimera_fragment_name("x y", np.zeros(1, dtype=pkg.CHIMERA_RESULT_DTYPE)[0], 0)


def test_5p_configuration_oracle_equals_model(sor, synth):
    """5' barcoding: the 5' adapter is searched like the TSO, the 3' adapter next to internal polyA/T, no BC + UMI gap"""
    ad5, ad3 = "CTACACGACGCTCTTCCGATCT", "AAGCAGTGGTATCAACGCAGAGTAC"
    wl = synth.make_whitelist(5000, seed=75)
    used = synth.pick_used(wl, 50, seed=76)
    reads = synth.gen_reads_5p(50, used, seed=77, max_mid=400)
    chim = synth.make_chimeras(reads, 50, seed=78)
    par = sor.chimera_params(tso=ad5, adapter=ad3, tso_max=5, adapter_max=5, bc_umi=0)
    n_split = 0
    for seq, _q, _k in chim:
        rc_, splits, multi, n_matches, _ = sor.chimera_split(seq, par)
        m_splits, m_multi, m_ms = pm.find_split_positions(seq, ad5, ad3, 5, 5, bc_umi=0)
        assert rc_ == 0 and splits == m_splits and multi == m_multi and n_matches == len(m_ms)
        n_split += len(splits) > 0
    assert n_split >= 10
