"""Independent plain-Python restatement of the reference's chimera splitter (3' barcoding), Java object semantics
kept (dict for AdapterScanRslt, lists with in-place removal, iterator walk) -- NOT sharing code with
oracle/sor_chimera.c.  Small cases only.

Cites: FJ!nanoporereadscanner/analyzers/ChimeraFindernew.java:L107-332,
FJ!nanopore/analyzers/PolyATadapterInternalSearcherBase.java:L78-270, FJ!nanopore/analyzers/AdapterTSOanalyzer.java:L130-156,L278-308.
"""
from pymodel_scan import NeedlemanMatch, count_errors, enc, f32, jround, kmers_matching, needleman

TSO_COMPLETE = "AAGCAGTGGTATCAACGCAGAGTACAT"
ADAPTER_COMPLETE = "CTACACGACGCTCTTCCGATCT"
A, T = 1, 8
TAGS = {"REV_ADAPTER": "RA", "FWD_ADAPTER": "FA", "REV_ADAPTER_FWD_ADAPTER": "RA_FA", "REV_ADAPTER_FWD_TSO": "RA_FT",
        "REV_TSO_FWD_ADAPTER": "RT_FA", "REV_TSO_FWD_TSO": "RT_FT", "READSTART": ""}
MIN_INT = -2 ** 31


def rc4(codes):
    sw = {0: 0, 1: 8, 8: 1, 2: 4, 4: 2, 15: 15}
    return [sw[b] for b in reversed(codes)]


def scan_kmers_internal(seq, begin, end, ad, max_errors, min_kmers):
    """scanForAdapterOrTSOseqKMERsForInternal -> AdapterScanRslt as {float key: [positions]} (insertion ordered)"""
    rslt = {}
    pos = begin
    while pos <= min(len(seq) - len(ad), end):
        delta = 1
        if kmers_matching(seq, ad, pos) >= min_kmers:
            ne = count_errors(needleman(ad, seq[pos - 1:pos - 1 + len(ad)]))
            if not (jround(ne) > max_errors):
                rslt.setdefault(float(ne), []).append(pos)
            if max_errors < ne:
                delta = max(1, jround(f32(ne - f32(max_errors))) - 1)
        pos += delta
    return rslt


def pos_below_max(rslt, max_mm):
    pairs = [(p, k) for k, lst in rslt.items() if not (k > max_mm) for p in lst]
    pairs.sort(key=lambda x: x[1])  # stable, Float.compare on the score
    return pairs or None


def internal_tso(seq, tso, max_mm, is_reverse, result):
    lst = pos_below_max(scan_kmers_internal(seq, 70, len(seq) - 70, tso, max_mm, 2), max_mm)
    if lst is not None and len(lst) > 1:
        remove = [i for i in range(len(lst) - 1, 0, -1) if abs(lst[i][0] - lst[i - 1][0]) < 3]
        for i in remove:
            del lst[i]
    prev = MIN_INT
    for p, _score in (lst or []):
        begin = p + len(tso) - 1 if is_reverse else p
        old, prev = prev, begin
        if begin > old + 120:
            result.append(dict(rev=is_reverse, begin=begin, kind="TSO"))


def search_at_end(seq, pos, base, cur, off, minlen, minfrac):
    pb = pos + 1
    while pb < len(seq) - off - minlen - 1 and not (f32(cur / f32(minlen)) < minfrac):
        if seq[pb] == base:
            cur = f32(cur - f32(1))
        if seq[pb + minlen] == base:
            cur = f32(cur + f32(1))
        if not (f32(cur / f32(minlen)) < minfrac):
            pos = pb
        if seq[pb + minlen - 1] != base and seq[pb + minlen - 2] != base:
            break
        pb += 1
    end = pos + minlen - 1
    while sum(seq[end - i] == base for i in range(4)) < 2:
        end -= 4
    while seq[end] != base:
        end -= 1
    return end


def at_scan(seq, minlen=15, minfrac=f32(0.70), window=150):
    off = window + 70
    out = []
    cur_t = cur_a = f32(0)
    end_t = end_a = 0
    i = off - 1
    while i < off + minlen - 1 and i < len(seq):
        if seq[i] == A:
            cur_a = f32(cur_a + f32(1))
        if seq[i] == T:
            cur_t = f32(cur_t + f32(1))
        i += 1
    pos = off - 1
    while pos < len(seq) - off:
        if seq[pos] == A:
            cur_a = f32(cur_a - f32(1))
        if seq[pos] == T:
            cur_t = f32(cur_t - f32(1))
        if seq[pos + minlen - 1] == A:
            cur_a = f32(cur_a + f32(1))
        if seq[pos + minlen - 1] == T:
            cur_t = f32(cur_t + f32(1))
        if not (f32(cur_t / f32(minlen)) < minfrac) and pos > end_t and seq[pos] == T and seq[pos + 1] == T:
            at = dict(begin=pos + 1, end=search_at_end(seq, pos, T, cur_t, off, minlen, minfrac) + 1, nuc=T, matches=None)
            end_t = at["end"]
            out.append(at)
        if not (f32(cur_a / f32(minlen)) < minfrac) and pos > end_a and seq[pos] == A and seq[pos + 1] == A:
            at = dict(begin=pos + 1, end=search_at_end(seq, pos, A, cur_a, off, minlen, minfrac) + 1, nuc=A, matches=None)
            end_a = at["end"]
            out.append(at)
        pos += 1
    return out


def adapter_scan(seq, at, ad, max_errors, bc_umi=28):
    if at["nuc"] == T:
        start_range = at["begin"] - bc_umi - 30 - 10
        end_range = start_range + 30 + 20
    else:
        end_range = at["end"] + bc_umi + 30 + 10
        start_range = end_range - 30 - 20
    sub = seq[start_range - 1:end_range]
    assert len(sub) == end_range - start_range + 1
    if at["nuc"] == A:
        sub = rc4(sub)
    rslt = scan_kmers_internal(sub, 1, len(sub) - len(ad), ad, max_errors, 3)
    if not rslt:
        return
    best = min(rslt)
    offs = list(rslt[best])
    i = len(offs) - 1
    while i > 0:
        if abs(offs[i] - offs[i - 1]) < 2:
            del offs[i]
        i -= 1
    matches = []
    for o in offs:
        nm = NeedlemanMatch(needleman(ad, sub[o - 1:o - 1 + len(ad)]))
        if nm.nmis > max_errors:
            continue
        matches.append(start_range + o - 1 if at["nuc"] == T else start_range + len(sub) - o)
    if matches:
        at["matches"] = matches


def find_split_positions(read, tso_complete=TSO_COMPLETE, adapter_complete=ADAPTER_COMPLETE, tso_max=6, ad_max=5, bc_umi=28):
    """-> (split positions [(reason, pos)], multi_chimeric flag, matches)"""
    if len(read) < 2 * 70 + 100:
        return [], False, []
    seq = enc(read)
    tso = enc(tso_complete)
    ms = []
    internal_tso(seq, tso, tso_max, False, ms)
    internal_tso(seq, rc4(tso), tso_max, True, ms)
    ad = enc(adapter_complete)
    ats = at_scan(seq)
    for at in ats:
        adapter_scan(seq, at, ad, ad_max, bc_umi)
    prev = {A: MIN_INT, T: MIN_INT}
    for at in ats:
        if at["matches"] is None:
            continue
        start = at["matches"][0]
        old, prev[at["nuc"]] = prev[at["nuc"]], start
        if start > old + 120:
            ms.append(dict(rev=at["nuc"] == A, begin=start, kind="ADAPTER"))

    def isolated(m):
        return ("REV_ADAPTER", m["begin"] + 25) if m["rev"] else ("FWD_ADAPTER", m["begin"] - 25)

    def paired(a, b):
        name = ("REV_ADAPTER" if a["kind"] == "ADAPTER" else "REV_TSO") + "_" + ("FWD_ADAPTER" if b["kind"] == "ADAPTER" else "FWD_TSO")
        return (name, a["begin"] + (b["begin"] - a["begin"]) // 2)

    splits = []
    if len(ms) == 1:
        if ms[0]["kind"] == "ADAPTER":
            splits.append(isolated(ms[0]))
    elif ms:
        it = iter(sorted(ms, key=lambda m: m["begin"]))
        rest = len(ms)

        def nxt():
            nonlocal rest
            rest -= 1
            return next(it)

        prev_m = nxt()
        while rest > 0 and prev_m is not None:
            cur = nxt()
            if cur["begin"] - prev_m["begin"] > 160:
                if prev_m["kind"] == "ADAPTER":
                    splits.append(isolated(prev_m))
                prev_m = cur
            elif prev_m["rev"] and not cur["rev"]:
                splits.append(paired(prev_m, cur))
                prev_m = nxt() if rest > 0 else None
            else:
                prev_m = cur
            if rest == 0 and prev_m is not None and prev_m["kind"] == "ADAPTER":
                splits.append(isolated(prev_m))
    if len(splits) > 1:
        remove = [splits[i] for i in range(1, len(splits)) if splits[i][1] - splits[i - 1][1] < 100]
        ids = {id(x) for x in remove}
        splits = [s for s in splits if id(s) not in ids]
    if len(splits) > 2:
        return [], True, ms
    return splits, False, ms


def fragments(name, read, qual, splits):
    """L288-326 -> [(name, seq, qual)]"""
    out = []
    start, start_reason, idx = 0, "READSTART", 0
    for i, (reason, pos) in enumerate(splits):
        out.append((name.replace(" ", "_%ssp%d " % (TAGS[reason], i + 1), 1), read[start:pos], qual[start:pos]))
        start, start_reason, idx = pos, reason, i
    out.append((name.replace(" ", "_%ssp%d " % (TAGS[start_reason], idx + 2), 1), read[start:], qual[start:]))
    return out
