"""CPU tests of the ORACLE for UMI pair distances (bounded Levenshtein, 3x3 offsets, packing, window geometry)."""
import random

import numpy as np

CODE = {"A": 1, "G": 2, "C": 4, "T": 8, "N": 15}


def lev(a, b):
    """textbook Levenshtein (independent of the banded two-row implementation under test)"""
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def rand_codes(rng, n, p_n=0.02):
    return [15 if rng.random() < p_n else rng.choice([1, 2, 4, 8]) for _ in range(n)]


def noisy(rng, s, k):
    s = list(s)
    for _ in range(k):
        op, p = rng.choice("sid"), rng.randrange(len(s))
        if op == "s":
            s[p] = rng.choice([1, 2, 4, 8])
        elif op == "i":
            s.insert(p, rng.choice([1, 2, 4, 8]))
        else:
            del s[p]
    return (s + rand_codes(rng, 14))[:14]


def test_limited_compare_is_bounded_levenshtein(sor):
    rng = random.Random(5)
    seen = set()
    for _ in range(3000):
        a = rand_codes(rng, 12)
        b = noisy(rng, a, rng.randrange(0, 8))[:12] if rng.random() < 0.8 else rand_codes(rng, 12)
        d = lev(a, b)
        got = sor.limited_compare(a, b, 4)
        assert got == (d if d <= 4 else -1)
        seen.add(got)
    assert seen == {-1, 0, 1, 2, 3, 4}


def model_pair(w1, w2):
    eds = [[0] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(3):
            d = lev(w1[i:i + 12], w2[j:j + 12])
            eds[i][j] = d if d <= 4 else 5
    best, b1, b2 = 127, 0, 0
    for i in (1, 2, 0):        # ZERO, PLUSONE, MINUSONE by value
        for v in (1, 2, 0):
            if eds[i][v] < best:
                best, b1, b2 = eds[i][v], i, v
    return best | (b1 << 4) | (b2 << 6)


def test_pair_and_matrix_against_model(sor):
    rng = random.Random(9)
    ws = []
    for _ in range(12):
        base = rand_codes(rng, 14)
        ws.append(base)
        for _ in range(4):
            ws.append(noisy(rng, base, rng.choice([0, 1, 1, 2, 3])))
    ws = np.array(ws, dtype=np.uint8)
    m = sor.umi_matrix(ws)
    n = ws.shape[0]
    for i in range(n):
        assert m[i, i] == (0 | (1 << 4) | (1 << 6))  # equality entry: (0, ZERO, ZERO)
        for v in range(i, n):
            r = model_pair(list(ws[i]), list(ws[v]))
            assert m[i, v] == r == sor.umi_pair(ws[i], ws[v])
            assert m[v, i] == ((r & 15) | (((r >> 6) & 3) << 4) | (((r >> 4) & 3) << 6))
    eds = m & 15
    assert set(np.unique(eds)) == {0, 1, 2, 3, 4, 5}
    # the first strict minimum in the order ZERO, PLUSONE, MINUSONE wins: for a copy shifted by one base the exact
    # pairs are (ZERO, MINUSONE) and (PLUSONE, ZERO); (ZERO, MINUSONE) is visited first
    a = rand_codes(random.Random(1), 16, 0)
    r = sor.umi_pair(a[0:14], a[1:15])
    assert r & 15 == 0 and ((r >> 4) & 3, (r >> 6) & 3) == (1, 0)


def test_window_from_read_name(sor):
    # /root/reference/README.md:400: AE=619 bcEnd=603 -> the barcode ends at position 19 of revcomp(X)
    X = "AAAAAAAAAAAATGGCGTGTATTGTCTTGGCACGATCGGAAGA"
    w = sor.umi_window_3p(X, 619, 603)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = "".join(comp[c] for c in reversed(X))
    assert rc[:3] == "TCT" and rc[3:19] == "TCCGATCGTGCCAAGA"
    assert [CODE[c] for c in rc[18:32]] == list(w)
    assert sor.umi_window_3p(X, 619, 592) is not None and sor.umi_window_3p(X, 619, 591) is None  # slice would leave the 43-base string
