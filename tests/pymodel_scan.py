"""Independent plain-Python restatement of the reference's 3' read scan (small cases only), written from the
bytecode listing with Java object semantics kept (strings for alignments, a dict for AdapterScanRslt, numpy
float32 for Java float arithmetic) -- deliberately NOT sharing code with oracle/sor_scan.c.

Cites: FJ!nanopore/analyzers/{PolyATSearcher,AdapterTSOanalyzer,Match,NeedlemanMatch,PolyATadapterAnalyzerBase}.java,
FJ!nanoporereadscanner/analyzers/PolyATadapterAnalyzer_3pBCUMI.java, TB!nuc/alignment/needleman/*.java.
"""
import numpy as np

f32 = np.float32
ENC4 = {"-": 0, "A": 1, "G": 2, "C": 4, "T": 8, "N": 15}
DEC4 = {v: k for k, v in ENC4.items()}
COMP = {"A": "T", "T": "A", "G": "C", "C": "G", "N": "N"}
T = 8


def enc(s):
    return [ENC4[c] for c in s.upper()]


def revcomp_str(s):
    return "".join(COMP[c] for c in reversed(s.upper()))


def find_polyt(seq, minlen=15, minfrac=f32(0.75), window=150):
    """PolyATSearcher.findpolyAT L56-252 on a list of 4-bit codes -> None or (begin1, end1)"""
    scores = []
    cur = f32(0.0)
    for i in range(minlen):
        if seq[i] == T:
            cur = f32(cur + f32(1.0))
    pos = -1
    while pos < window - 1:
        pos += 1
        d = (-1 if seq[pos] == T else 0) + (1 if seq[pos + minlen] == T else 0)
        cur = f32(cur + f32(d))
        scores.append((pos, f32(cur / f32(minlen))))
    count_t = lambda s, st, w: sum(1 for i in range(st, st + w) if s[i] == T)  # noqa: E731
    first = None
    for p, v in scores:
        if not (v < minfrac) and seq[p] == T and count_t(seq, p, 5) > 2:
            first = p
            break
    if first is None:
        return None
    start = first
    for inc in (20, 15, 10, 5, 4, 3, 2, 1):
        while start + inc < window and not (float(scores[start + inc][1]) < float(minfrac) - 0.1):
            start += inc
    endpos = start + minlen - 1
    while endpos > 4:
        nts, tot = [], 0
        for i in range(5):
            if seq[endpos - i] == T:
                tot += 1
            nts.append(tot)
        if nts[0] != 0 and nts[1] >= 2 and nts[3] >= 3 and nts[4] >= 4:
            break
        endpos -= 1
    n = len(seq)
    while n > endpos + 6 and count_t(seq, endpos + 1, 5) > 3:
        endpos += 5
    while n > endpos + 4 and count_t(seq, endpos + 1, 3) > 1:
        endpos += 3
    while endpos < n - 1 and seq[endpos + 1] == T:
        endpos += 1
    return first + 1, endpos + 1


class Cell:
    __slots__ = ("row", "col", "score", "prev")

    def __init__(self, r, c):
        self.row, self.col, self.score, self.prev = r, c, 0, None


def needleman(seq1, seq2, match=5, mismatch=-5, space=-5, lead1=-4, lead2=-5):
    """NeedlemanWunsch(sequence1 = template on columns, sequence2 = read on rows) -> (tmpl, dots, read) strings"""
    n1, n2 = len(seq1), len(seq2)
    tab = [[Cell(r, c) for c in range(n1 + 1)] for r in range(n2 + 1)]
    for r in range(n2 + 1):
        for c in range(n1 + 1):
            cell = tab[r][c]
            if r == 0 and c != 0:
                cell.score, cell.prev = c * lead2, tab[r][c - 1]
            elif c == 0 and r != 0:
                cell.score, cell.prev = r * lead1, tab[r - 1][c]
    for r in range(1, n2 + 1):
        for c in range(1, n1 + 1):
            above, left, diag = tab[r - 1][c], tab[r][c - 1], tab[r - 1][c - 1]
            row_space = above.score + space
            col_space = left.score + space
            mm = diag.score + (match if (seq2[r - 1] & seq1[c - 1]) != 0 else mismatch)
            cur = tab[r][c]
            if row_space >= col_space:
                if mm >= row_space:
                    cur.score, cur.prev = mm, diag
                else:
                    cur.score, cur.prev = row_space, above
            else:
                if mm >= col_space:
                    cur.score, cur.prev = mm, diag
                else:
                    cur.score, cur.prev = col_space, left
    a1, a2 = [], []
    cur = tab[n2][n1]
    while cur.prev is not None:
        a2.insert(0, seq2[cur.row - 1] if cur.row - cur.prev.row == 1 else 0)
        a1.insert(0, seq1[cur.col - 1] if cur.col - cur.prev.col == 1 else 0)
        cur = cur.prev
    dots = ""
    for b1, b2 in zip(a1, a2):
        dots += "x" if (b1 == 0 or b2 == 0 or (b1 & b2) == 0) else "."
    dec = lambda a: "".join(DEC4.get(b, "?") for b in a)  # noqa: E731
    return dec(a1), dots, dec(a2)


def count_errors(aln):
    tmpl, dots, _ = aln
    lead = 0
    while tmpl[lead] == "-":
        lead += 1
    return f32(f32(dots.count("x")) - f32(f32(0.9) * f32(lead)))


class NeedlemanMatch:
    def __init__(self, aln):
        self.match, self.pattern, self.read = aln
        ins = dele = sub = 0
        for i in range(len(self.match)):
            if self.pattern[i] == "x":
                if self.match[i] == "-":
                    ins += 1
                elif self.read[i] == "-":
                    dele += 1
                else:
                    sub += 1
        i = len(self.read)
        while self.read[i - 1] == "-":
            i -= 1
        dele -= len(self.read) - i
        self.ins, self.dele, self.sub = ins, dele, sub
        self.nmis = ins + dele + sub

    def end_of_read(self, n):
        ret = f32(0.0)
        i = len(self.read) - 1
        k = i
        while k >= len(self.read) - n and i >= 0:
            if self.pattern[i] == "x":
                if k >= len(self.read) - 2:
                    ret = f32(float(ret) + 1.2)
                else:
                    ret = f32(ret + f32(1.0))
            if self.read[i] != "-":
                k -= 1
            i -= 1
        return ret

    def has_3p_matches(self, n):
        c = 0
        i = len(self.pattern) - 1
        while i >= len(self.pattern) - n and self.pattern[i] == ".":
            c += 1
            i -= 1
        return c == n


def kmers_matching(read, ad, pos1):
    m, p, k = 0, pos1 - 1, 0
    while p < len(read) - 3 and k < len(ad) - 3:
        if all(read[p + v] & ad[k + v] for v in range(4)):
            m += 1
        p += 1
        k += 1
    return m


def jround(x):
    """Math.round(float)"""
    import math

    return int(math.floor(float(f32(f32(x) + f32(0.5)))))


def scan_adapter(read, begin, end, ad, max_errors=None):
    """AdapterTSOanalyzer.scanForAdapterOrTSOseq L84-110"""
    rslt = {}
    pos = begin
    while pos <= min(len(read) - len(ad), end):
        delta = 1
        if kmers_matching(read, ad, pos) > 1:
            ne = count_errors(needleman(ad, read[pos - 1:pos - 1 + len(ad)]))
            if max_errors is None or not (jround(ne) > max_errors):
                rslt.setdefault(float(ne), []).append(pos)
            if max_errors is not None and max_errors < ne:
                delta = max(1, jround(f32(ne - f32(max_errors))) - 1)
        pos += delta
    return rslt


def n_consecutive(pattern):
    ret = cur = 0
    for ch in pattern:
        if ch == ".":
            cur += 1
        else:
            ret, cur = max(ret, cur), 0
    return ret


def best_two(pattern):
    runs, cur = [], 0
    for ch in pattern:
        if ch == ".":
            cur += 1
        else:
            if cur > 4:
                runs.append(cur)
            cur = 0
    return sum(sorted(runs)[:2])


def scan_tsos(read):
    """PolyATadapterAnalyzer_3pBCUMI.scanReadForTSOs L122-190 -> (flags set, tso_start, tso_end)"""
    tso = enc("AACGCAGAGTACATGG")
    L = len(read)
    fwd = enc(read[:116])
    rev = enc(revcomp_str(read[L - 116:]))

    def one(seq):
        r = scan_adapter(seq, 1, 90, tso, 5)
        if not r:
            return None
        pos = r[min(r)][0]
        nm = NeedlemanMatch(needleman(tso, seq[pos - 1:pos + 15]))
        nm.passed = nm.nmis <= 5
        nm.end_scan = pos + 15 + nm.ins - nm.dele
        return nm

    f, r = one(fwd), one(rev)
    found = lambda m: m is not None and m.passed  # noqa: E731
    if not found(f) and not found(r):
        for m in (f, r):
            if m is not None:
                m.passed = n_consecutive(m.pattern) >= 8
        if not found(f) and not found(r):
            for m in (f, r):
                if m is not None:
                    m.passed = best_two(m.pattern) >= 12
    if found(f) and found(r) and abs(f.nmis - r.nmis) > 3:
        if f.nmis > r.nmis:
            f = None
        else:
            r = None
    ff, rf = found(f), found(r)
    flags = set()
    if ff and not rf:
        flags.add("TSO_5P")
    elif rf and not ff:
        flags.add("TSO_3P")
    elif ff and rf:
        flags.add("TSO_5P_AND_3P")
    return flags, (f.end_scan if ff else 0), (r.end_scan if rf else 0)


def scan_read_3p(read, qual, adapter, max_mm=3, min_len=200, min_3p=8, min_bc_qv=8, min_read_qv=8):
    out = dict(flags=set(), adapter_found=0, pass1_ok=0)
    L = len(read)
    out["tso_start"] = out["tso_end"] = 0
    if L < min_len:
        out["flags"] |= {"READ_TOO_SHORT", "FAILED"}
        return out
    tf, out["tso_start"], out["tso_end"] = scan_tsos(read)
    out["flags"] |= tf
    ad = enc(adapter)
    fwd = enc(read[:175])
    rev = enc(revcomp_str(read[L - 175:]))
    pf, pr = find_polyt(fwd), find_polyt(rev)
    out["flags"].add("POLY_A_NOT_FOUND" if not pf and not pr else "POLY_T_5P" if pf and not pr else
                     "POLY_A_3P" if pr and not pf else "POLY_T_5P_POLY_A_3P")
    sf = scan_adapter(fwd[:pf[1]], 1, pf[1] - 12, ad) if pf else None
    sr = scan_adapter(rev[:pr[1]], 1, pr[1] - 12, ad) if pr else None
    use_fwd = None
    if sf is not None or sr is not None:
        if sf and sr:
            if abs(f32(min(sf)) - f32(min(sr))) >= 2.0:
                out["flags"].add("ADAPTER_SELECTED_DESP_BOTH")
                use_fwd = not (min(sf) >= min(sr))
            else:
                out["flags"].add("ADAPTER_5P_AND_3P")
        elif sf and not sr:
            use_fwd = True
        elif not sf and sr:
            use_fwd = False
    if use_fwd is None:
        out["flags"].add("FAILED")
        return out
    p = pf if use_fwd else pr
    out["polya_start"], out["polya_end"] = L - (p[1] - 1), L - (p[0] - 1)
    s = sf if use_fwd else sr
    test = (fwd if use_fwd else rev)[:p[1]]
    offsets = s[min(s)]

    def create(pos):
        nm = NeedlemanMatch(needleman(ad, test[pos - 1:pos - 1 + len(ad)]))
        if nm.nmis > max_mm and not nm.has_3p_matches(6):
            return None
        nm.start, nm.end = pos, pos + len(ad) - 1 + nm.ins - nm.dele
        return nm

    if len(offsets) == 1:
        m = create(offsets[0])
        lst = [m] if m else []
    else:
        groups = {}
        for o in offsets:
            m = create(o)
            if m:
                groups.setdefault(float(m.end_of_read(5)), []).append(m)
        lst = groups[min(groups)] if groups else []
    if not lst:
        out["flags"].add("FAILED")
        return out
    m = lst[0]
    out["adapter_found"] = 1
    out["adapter_start"], out["adapter_end"] = L - (m.start - 1), L - (m.end - 1)
    out["adapter_nmis"] = m.nmis
    out["reverse"] = 1 if use_fwd else 0
    out["flags"] |= {"ADAPTER_5P", "PASSED_REV"} if use_fwd else {"ADAPTER_3P", "PASSED_FWD"}
    if qual is not None:
        mean = lambda a, b: f32(np.mean([ord(c) - 33 for c in qual[a - 1:b]], dtype=np.float64))  # noqa: E731
        ae = out["adapter_end"]
        out["pass1_ok"] = int(m.end_of_read(min_3p) == 0.0 and not (mean(ae - 16, ae - 1) < min_bc_qv)
                              and not (mean(1, L) < min_read_qv))
    return out


def scan_read_5p(read, qual, adapter, max_mm=4, window=110, dont_search_polya=True, min_len=200, min_3p=8, min_bc_qv=8,
                 min_read_qv=8):
    """PolyATadapterAnalyzer_5pBCUMI.search L43-76 + analyze for 5' barcoding (scan coordinates kept as they are)"""
    out = dict(flags=set(), adapter_found=0, pass1_ok=0, polya_start=0, polya_end=0)
    L = len(read)
    if L < min_len:
        out["flags"] |= {"READ_TOO_SHORT", "FAILED"}
        return out
    ad = enc(adapter)
    pf = pr = None
    if not dont_search_polya:
        pf, pr = find_polyt(enc(read[:175])), find_polyt(enc(revcomp_str(read[L - 175:])))
        out["flags"].add("POLY_A_NOT_FOUND" if not pf and not pr else "POLY_T_5P" if pf and not pr else
                         "POLY_A_3P" if pr and not pf else "POLY_T_5P_POLY_A_3P")
    n_end = window + len(ad) + max_mm + 5
    sr = sf = None
    three = five = None
    if pf is not None or dont_search_polya:
        three = enc(revcomp_str(read[L - n_end:]))
        sr = scan_adapter(three, 1, window, ad)
    if pr is not None or dont_search_polya:
        five = enc(read[:n_end])
        sf = scan_adapter(five, 1, window, ad)
    use_fwd = None
    if sf is not None or sr is not None:
        if sf and sr:
            if abs(f32(min(sf)) - f32(min(sr))) >= 2.0:
                out["flags"].add("ADAPTER_SELECTED_DESP_BOTH")
                use_fwd = not (min(sf) >= min(sr))
            else:
                out["flags"].add("ADAPTER_5P_AND_3P")
        elif sf and not sr:
            use_fwd = True
        elif not sf and sr:
            use_fwd = False
    if use_fwd is None:
        out["flags"].add("FAILED")
        return out
    if not dont_search_polya:
        p = pr if use_fwd else pf  # useforward == false selects the FORWARD polyT result for 5' (L170-171)
        out["polya_start"], out["polya_end"] = L - (p[1] - 1), L - (p[0] - 1)
    s = sf if use_fwd else sr
    test = five if use_fwd else three
    offsets = s[min(s)]

    def create(pos):
        nm = NeedlemanMatch(needleman(ad, test[pos - 1:pos - 1 + len(ad)]))
        if nm.nmis > max_mm and not nm.has_3p_matches(6):
            return None
        nm.start, nm.end = pos, pos + len(ad) - 1 + nm.ins - nm.dele
        return nm

    if len(offsets) == 1:
        m = create(offsets[0])
        lst = [m] if m else []
    else:
        groups = {}
        for o in offsets:
            m = create(o)
            if m:
                groups.setdefault(float(m.end_of_read(5)), []).append(m)
        lst = groups[min(groups)] if groups else []
    if not lst:
        out["flags"].add("FAILED")
        return out
    m = lst[0]
    out["adapter_found"] = 1
    out["adapter_start"], out["adapter_end"] = m.start, m.end
    out["adapter_nmis"] = m.nmis
    out["reverse"] = 0 if use_fwd else 1
    out["flags"] |= {"ADAPTER_5P", "PASSED_FWD"} if use_fwd else {"ADAPTER_3P", "PASSED_REV"}
    if qual is not None:
        mean = lambda a, b: f32(np.mean([ord(c) - 33 for c in qual[a - 1:b]], dtype=np.float64))  # noqa: E731
        ae = out["adapter_end"]
        out["pass1_ok"] = int(m.end_of_read(min_3p) == 0.0 and not (mean(ae - 16, ae - 1) < min_bc_qv)
                              and not (mean(1, L) < min_read_qv))
    return out
