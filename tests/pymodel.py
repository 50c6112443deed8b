"""Second, independent restatement of the reference's barcode matcher in plain Python (small cases only).

Written statement-by-statement from the bytecode listing (tools/classfold.py) with Java long semantics made
explicit, and with java.util.HashMap modelled as real bucket lists (put / resize / split), i.e. NOT sharing
code or shortcuts with oracle/sor_bc.c.  Tests check the C oracle against this model; both cite the same
reference lines (FJ!nanoporereadscanner/analyzers/BarcodeMatchTester.java, Parser.java:L195-315,
TB!nuc/encoding/TwoBit/NucleicAcidTwoBitPerBase.java).
"""
M64 = (1 << 64) - 1


def to_signed(v):
    v &= M64
    return v - (1 << 64) if v >> 63 else v


def lshl(v, s):
    return (v << (s & 63)) & M64


def lushr(v, s):
    return (v & M64) >> (s & 63)


BASE2 = {"A": 0, "a": 0, "G": 1, "g": 1, "C": 2, "c": 2, "T": 3, "t": 3}
ENC4 = {"-": 0, "A": 1, "G": 2, "C": 4, "T": 8, "N": 15, "H": 13, "R": 3, "Y": 12, "M": 5, "K": 10, "S": 6, "W": 9,
        "B": 14, "V": 7, "D": 11}
RC4 = {0: 0, 1: 8, 8: 1, 2: 4, 4: 2, 15: 15, 13: 11, 11: 13, 3: 12, 12: 3, 5: 10, 10: 5, 6: 6, 9: 9, 14: 7, 7: 14}


def encode(s):  # getLongHashForSeq L183-187
    r = 0
    for ch in s:
        r = lshl(r, 2) | ((BASE2[ch] if ch in BASE2 else -2) & M64)
    return r


def revcomp(seq, n):  # reverseComplement L477-484
    rc = [3, 2, 1, 0]
    t = 0
    for _ in range(n):
        t = lshl(t, 2) | rc[seq & 3]
        seq = lushr(seq, 2)
    return t


CLEAR = []
_l = (-4) & M64
for _i in range(32):
    CLEAR.append(_l)
    _l = lshl(_l, 2) | 3
SETB = [[0, 1, 3, 2]]
for _i in range(31):
    SETB.append([lshl(x, 2) for x in SETB[-1]])
B2L0 = {1: 0, 2: 1, 4: 2, 8: 3}  # BYTE_TO_2BITLONG_ARRAY[0]; all other codes stay 0


def replace_deg(seq, pos, n):  # L228-233
    seq &= CLEAR[n - pos - 1]
    sh = (n - (pos + 1)) << 1
    return [seq | lshl(b, sh) for b in range(4)]


def insert_deg(h, pos, n):  # L300-309
    sh = (n - pos - 1) << 1
    upper = lshl(lushr(h, sh), sh)
    sh = 64 - sh
    h = lshl(h, sh)
    h = lushr(h, sh + 2)
    return [upper | h | SETB[n - (pos + 1) - 1][j] for j in (0, 1, 3, 2)]


def delete_byte(h, base4, pos, n):  # L321-327
    sh = (n - pos) << 1
    upper = lshl(lushr(h, sh), sh)
    sh = 64 - sh
    h = lshl(h, sh + 2)
    h = lushr(h, sh)
    return upper | h | B2L0.get(base4, 0)


class JHashSet:
    """java.util.HashSet over (key, hash, eq-tuple): real buckets, JDK 9+ putVal/resize behaviour"""

    def __init__(self):
        self.tab = [[] for _ in range(16)]
        self.size = 0

    def _resize(self):
        old = self.tab
        oc = len(old)
        new = [[] for _ in range(oc * 2)]
        for j, b in enumerate(old):
            for e in b:  # split preserving order: lo stays at j, hi goes to j + oc
                (new[j] if (e[0] & oc) == 0 else new[j + oc]).append(e)
        self.tab = new

    def add(self, h, eqkey, value):
        h ^= h >> 16
        b = self.tab[h & (len(self.tab) - 1)]
        for e in b:
            if e[0] == h and e[1] == eqkey:
                return False
        bin_count_before = len(b)
        b.append((h, eqkey, value))
        if bin_count_before >= 8:  # binCount >= TREEIFY_THRESHOLD - 1
            assert len(self.tab) < 64, "treeification not modelled"
            self._resize()
        self.size += 1
        if self.size > (len(self.tab) * 3) // 4:
            self._resize()
        return True

    def __iter__(self):
        for b in self.tab:
            for e in b:
                yield e[2]


def one_match_hash(read_seq):
    return (read_seq ^ (read_seq >> 32)) & 0xFFFFFFFF


def bc_match(search, seq, n, ed, skip_full, allow_indels, post, offset, do_next):
    """BarcodeMatchTester.doJob L198-244.  search: python set of unsigned 64-bit ints.  Returns (list, n_probes)."""
    matches = JHashSet()
    probes = [0]
    use_tested = ed >= 2
    is_long = n > 16
    tested = set()

    def tkey(s):
        return s if is_long else (s & 0xFFFFFFFF)

    def check(m):
        if skip_full and m["seq"] == seq:
            return None
        probes[0] += 1
        if m["seq"] not in search:
            return None
        return dict(read_seq=seq, matching_bc=m["seq"], ed=m["level"], offset=offset, subs=m["ns"], ins=m["ni"],
                    dels=m["nd"])

    def add(r):
        matches.add(one_match_hash(r["read_seq"]), (r["read_seq"], r["ed"], r["offset"]), r)

    deque = []

    def go_next(m):
        if ed <= m["level"]:
            return
        c = dict(m)
        c["prev"] = m["pos"]
        c["pos"] = -1
        c["level"] = m["level"] + 1
        deque.append(c)

    parent = dict(seq=seq, prev=-1, pos=-1, level=0, ns=0, ni=0, nd=0)
    r = check(parent)
    if r:
        add(r)
    if ed == 0:
        return list(matches), probes[0]
    parent["level"] = 1
    deque.append(parent)
    last = n - 1
    while deque:
        cur = deque.pop()
        cur["pos"] += 1
        if cur["pos"] < last:
            deque.append(dict(cur))
        if cur["prev"] == cur["pos"]:
            continue
        # substitutions L257-269
        for s in replace_deg(cur["seq"], cur["pos"], n):
            if s == cur["seq"] or (use_tested and tkey(s) in tested):
                continue
            m = dict(cur)
            m["ns"] += 1
            m["seq"] = s
            r = check(m)
            if r:
                add(r)
            if r or do_next:
                go_next(m)
        if allow_indels and cur["pos"] < last:
            # insertions L284-296 (nDeletions++)
            for s in insert_deg(cur["seq"], cur["pos"], n):
                if use_tested and tkey(s) in tested:
                    continue
                m = dict(cur)
                m["seq"] = s
                m["nd"] += 1
                r = check(m)
                if r:
                    add(r)
                if (not r) or do_next:
                    go_next(m)
            # deletions L313-352 (nInsertions++)
            if not (post is not None and cur["nd"] + 1 > len(post)):
                last_base = post[cur["nd"] + 1 - 1] if post is not None else 0
                m0 = delete_byte(cur["seq"], last_base, cur["pos"], n)
                variants = [m0] if post is not None else [m0, m0 | 1, m0 | 2, m0 | 3]
                for s in variants:
                    if use_tested and tkey(s) in tested:
                        continue
                    m = dict(cur)
                    m["seq"] = s
                    m["ni"] += 1
                    r = check(m)
                    if r:
                        add(r)
                    if (not r) or do_next:
                        go_next(m)
        if use_tested:
            tested.add(tkey(cur["seq"]))
    return list(matches), probes[0]


def assign_barcode(search, stranded, adapterpos, max_ed=1, test_pm=2, five_prime=False, bc_len=16):
    """Parser.assignBarcode L195-315.  Returns None where Java would throw, else a dict (found=0/1)."""
    allm = JHashSet()
    offsets = sorted(range(-test_pm, test_pm + 1), key=abs)  # stable: 0,-1,1,-2,2
    for off in offsets:
        if not five_prime:
            bs, be = adapterpos - bc_len + off, adapterpos - 1 + off
        else:
            bs, be = adapterpos + 1 + off, adapterpos + bc_len + off
        if bs - 1 < 0 or be > len(stranded) or bs - 1 > be:
            return None
        bc = encode(stranded[bs - 1:be])
        if not five_prime:
            if bs - 5 < 0 or bs > len(stranded):
                return None
            post = [RC4[ENC4[c.upper()]] for c in reversed(stranded[bs - 5:bs])]
            bc = revcomp(bc, bc_len)
        else:
            if be + 5 > len(stranded):
                return None
            post = [ENC4[c.upper()] for c in stranded[be:be + 5]]
        ms, _ = bc_match(search, bc, bc_len, max_ed, False, True, post, off, True)
        for r in ms:
            allm.add(one_match_hash(r["read_seq"]), (r["read_seq"], r["ed"], r["offset"]), r)
    lst = list(allm)
    if not lst:
        return dict(found=0)
    import functools

    def cmp(a, b):
        if a["ed"] != b["ed"]:
            return -1 if a["ed"] < b["ed"] else 1
        if a["offset"] == 0 and b["offset"] != 0:
            return -1
        if a["offset"] != 0 and b["offset"] == 0:
            return 1
        return 0

    lst.sort(key=functools.cmp_to_key(cmp))  # list.sort is stable like Stream.sorted()
    seen, distinct = set(), []
    for r in lst:
        if r["matching_bc"] not in seen:
            seen.add(r["matching_bc"])
            distinct.append(r)
    best = distinct[0]
    second = distinct[1] if len(distinct) > 1 else None
    if best["ed"] > max_ed or (second is not None and best["ed"] >= second["ed"]):
        return dict(found=0)
    imd = best["ins"] - best["dels"]
    if not five_prime:
        start = adapterpos - 1 + best["offset"]
        end = start - (bc_len - 1) - imd
    else:
        start = adapterpos + 1 + best["offset"]
        end = start + (bc_len - 1) + imd
    return dict(found=1, bc=best["matching_bc"], ed=best["ed"], ed_sec=second["ed"] if second else 2147483647,
                offset=best["offset"], ins_minus_del=imd, bc_start=start, bc_end=end)
