"""The banded Needleman-Wunsch fill of K-SCAN / K-CHIM (csrc/smi_nw.h "Band") against the full matrix, in plain Python.

The kernels fill only the cells |row - column| <= W, with W from the guarantee of the reference's 4-mer gate (> 1 matching 4-mers on
the main diagonal of the aligned slice = at least MIN_DIAG = 5 matching bases there; 6 behind the three-4-mer gate of the internal
adapter scan).  This test restates both fills independently of the C++ (objects and strings, as tests/pymodel_scan.py does for the
full matrix) and checks on gate-passing inputs -- random, mutated copies of the pattern, shifted copies that pull the optimal path
off the diagonal, reads with N -- that the banded fill yields the SAME alignment strings as the reference's full fill
(NeedlemanWunsch.fillInCell L55-80 tie order, traceback from the bottom-right cell).
"""
import random

import pytest

from pymodel_scan import ENC4, needleman

NEG = -10 ** 9


def nw_band(n, min_diag):
    """largest |r - c| kept: the same recurrence as nw_band<N, MIN_DIAG>() in smi_nw.h"""
    w = 0
    while w + 1 < n and 14 * (w + 1) <= 10 * (n - min_diag):
        w += 1
    return w


def needleman_banded(seq1, seq2, w, match=5, mismatch=-5, space=-5, lead1=-4, lead2=-5):
    """same scores and tie order, cells with |r - c| > w never computed; a cell at the edge of the band only compares the
    predecessors inside it (what the kernel's two-operand max does).  -> (tmpl, dots, read) strings like pymodel_scan.needleman"""
    n1, n2 = len(seq1), len(seq2)
    assert n1 == n2
    score = {}
    prev = {}
    for c in range(0, min(n1, w) + 1):
        score[(0, c)] = c * lead2
        prev[(0, c)] = (0, c - 1) if c else None
    for r in range(1, min(n2, w) + 1):
        score[(r, 0)] = r * lead1
        prev[(r, 0)] = (r - 1, 0)
    for r in range(1, n2 + 1):
        for c in range(max(1, r - w), min(n1, r + w) + 1):
            diag = score[(r - 1, c - 1)] + (match if (seq2[r - 1] & seq1[c - 1]) != 0 else mismatch)
            above = score[(r - 1, c)] + space if (r - 1, c) in score else None
            left = score[(r, c - 1)] + space if (r, c - 1) in score else None
            # fillInCell: rowSpace >= colSpace ? (mm >= rowSpace ? diag : above) : (mm >= colSpace ? diag : left)
            if above is not None and (left is None or above >= left):
                best, frm = (diag, (r - 1, c - 1)) if diag >= above else (above, (r - 1, c))
            elif left is not None:
                best, frm = (diag, (r - 1, c - 1)) if diag >= left else (left, (r, c - 1))
            else:
                best, frm = diag, (r - 1, c - 1)
            score[(r, c)] = best
            prev[(r, c)] = frm
    a1, a2 = [], []
    cur = (n2, n1)
    while prev[cur] is not None:
        p = prev[cur]
        a2.insert(0, seq2[cur[0] - 1] if cur[0] - p[0] == 1 else 0)
        a1.insert(0, seq1[cur[1] - 1] if cur[1] - p[1] == 1 else 0)
        cur = p
    dec = {v: k for k, v in ENC4.items()}
    dots = "".join("x" if (b1 == 0 or b2 == 0 or (b1 & b2) == 0) else "." for b1, b2 in zip(a1, a2))
    return "".join(dec.get(b, "?") for b in a1), dots, "".join(dec.get(b, "?") for b in a2)


def n_kmers_matching(pat, sl):
    """Kmers.nKmersMatching_4mer on the main diagonal"""
    return sum(1 for i in range(len(pat) - 3) if all((pat[i + v] & sl[i + v]) != 0 for v in range(4)))


def diag_matches(pat, sl):
    return sum(1 for a, b in zip(pat, sl) if (a & b) != 0)


PATTERNS = {
    10: "CTTCCGATCT",                      # adapter of pass 2 (config.xml:111)
    16: "AACGCAGAGTACATGG",                # TSO of K-SCAN (config.xml:155)
    22: "CTACACGACGCTCTTCCGATCT",          # complete adapter (pass 1, internal scans)
    27: "AAGCAGTGGTATCAACGCAGAGTACAT",     # complete TSO (K-CHIM)
}


def slices_for(pat_s, rng, n_cases):
    """gate-passing read slices of len(pat): mutated copies, shifted copies with a planted diagonal stretch, random with a plant"""
    n = len(pat_s)
    bases = "AGCT"
    out = []
    while len(out) < n_cases:
        kind = rng.randrange(4)
        if kind == 0:  # copy with substitutions / indels (a true site)
            s = list(pat_s)
            for _ in range(rng.randrange(0, 5)):
                p = rng.randrange(len(s))
                op = rng.randrange(3)
                if op == 0:
                    s[p] = rng.choice(bases)
                elif op == 1:
                    s.insert(p, rng.choice(bases))
                elif len(s) > 1:
                    del s[p]
            s = (s + [rng.choice(bases) for _ in range(n)])[:n]
        elif kind == 1:  # the pattern shifted by k (optimal path off the diagonal), with a 5-base diagonal plant to pass the gate
            k = rng.randrange(1, max(2, n // 2))
            if rng.random() < 0.5:
                s = [rng.choice(bases) for _ in range(k)] + list(pat_s[: n - k])
            else:
                s = list(pat_s[k:]) + [rng.choice(bases) for _ in range(k)]
            p = rng.randrange(0, n - 4)
            s[p:p + 5] = pat_s[p:p + 5]
        elif kind == 2:  # random with a 5-base plant
            s = [rng.choice(bases) for _ in range(n)]
            p = rng.randrange(0, n - 4)
            s[p:p + 5] = pat_s[p:p + 5]
        else:  # two separate 4-base plants + an N
            s = [rng.choice(bases) for _ in range(n)]
            p = rng.randrange(0, n - 8)
            s[p:p + 4] = pat_s[p:p + 4]
            q = rng.randrange(p + 4, n - 3)
            s[q:q + 4] = pat_s[q:q + 4]
            s[rng.randrange(n)] = "N"
        out.append("".join(s))
    return out


@pytest.mark.parametrize("n,min_kmers,min_diag", [(10, 2, 5), (16, 2, 5), (22, 2, 5), (27, 2, 5), (22, 3, 6)])
def test_banded_fill_equals_full_fill_on_gated_slices(n, min_kmers, min_diag):
    rng = random.Random(1000 + 7 * n + min_kmers)
    pat_s = PATTERNS[n]
    pat = [ENC4[c] for c in pat_s]
    w = nw_band(n, min_diag)
    assert w == {(10, 5): 3, (16, 5): 7, (22, 5): 12, (27, 5): 15, (22, 6): 11}[(n, min_diag)]
    checked = 0
    for s in slices_for(pat_s, rng, 1500 if n <= 16 else 500):
        sl = [ENC4[c] for c in s]
        if n_kmers_matching(pat, sl) < min_kmers:
            continue  # the kernels never align such a slice
        assert diag_matches(pat, sl) >= min_diag  # what the band is derived from
        assert needleman_banded(pat, sl, w) == needleman(pat, sl), s
        checked += 1
    assert checked > 200


def test_band_is_not_vacuous():
    """one cell narrower and the alignments do change: the bound is doing work, and is tight to within a few cells"""
    n = 16
    pat_s = PATTERNS[n]
    pat = [ENC4[c] for c in pat_s]
    rng = random.Random(5)
    differs = 0
    for s in slices_for(pat_s, rng, 3000):
        sl = [ENC4[c] for c in s]
        if n_kmers_matching(pat, sl) < 2:
            continue
        if needleman_banded(pat, sl, 2) != needleman(pat, sl):
            differs += 1
    assert differs > 0
