"""Host-side mirror of the reference's `assignumis` worker for one chunk of aligned reads (UmiFinderWorker ->
ReadGrouper.groupSams -> UmiClustering.cluster, FJ!umifinder/UmiFinderWorker.java, FJ!umifinder/bamreaders/ReadGrouper.java:L82-260,
FJ!umifinder/analyzers/clustering/UmiClustering.java:L97-161): region grouping (host C++), UMI pair distances (K-UMI on
the device), clustering (host C++), and the values of the tags U8 / U7 / U1 / U2.  3' barcoding by default, 5' barcoding with `five_prime=True` (the reference's `-p`,
UmiFinderMain.java:L249).  No CPU fallback.

Per read the caller supplies what the reference parses from the BAM record: the read name written by scanfastq
(FastqRecordExt.getScanDatFromReadName, FastqRecordExt.java:L395-496: X=, AE=, bcEnd=, Q=, cellBC), the strand flag
and the clustering position (NanoporeRead$ReadScanData.getGenomePosition; smi_ref_position_at_read_position).
"""
import os
import re

import numpy as np
import torch

from . import lib as _lib

_CODE = {"A": 1, "G": 2, "C": 4, "T": 8, "N": 15}
_DEC = {1: "A", 2: "G", 4: "C", 8: "T", 15: "N"}
_COMP = {1: 8, 8: 1, 2: 4, 4: 2, 15: 15}
_FIELD = {k: re.compile(r"_" + k + r"=([^_ ]+)") for k in ("AE", "bcEnd", "X", "Q")}


def _extract(sub, tag):
    """FastqRecordExt.lambda$getScanDatFromReadName$4 (L397-408): text behind the first `tag` up to the next '_'"""
    i = sub.find(tag)
    if i < 0:
        return None
    i += len(tag)
    j = sub.find("_", i)
    return sub[i:] if j < 0 else sub[i:j]


def scan_data_from_name(name, bc_edit_limit=None):
    """FastqRecordExt.getScanDatFromReadName (FastqRecordExt.java:L395-496): None when the name carries no _REV_ / _FWD_
    marker; raises where the reference throws AdapterInfoNotFoundInReadException (no AE=).  The barcode fields are only
    taken when ed <= bc_edit_limit (-b, L450-456)."""
    k = name.find("_REV_")
    rev = k >= 0
    if not rev:
        k = name.find("_FWD_")
        if k < 0:
            return None
    sub = name[k + 4:]
    ae = _extract(sub, "AE=")
    if ae is None:
        raise _lib.SmiError("adapter position (AE=) not found in read name: " + name)
    d = dict(reverse=rev, ae=int(ae), ps=None, pe=None, tso=None, bc=None, x=_extract(sub, "X="), q=None)
    for key, tag in (("ps", "PS="), ("pe", "PE="), ("tso", "T=")):
        v = _extract(sub, tag)
        if v is not None:
            d[key] = int(v)
    ed = _extract(sub, "ed=")
    if ed is not None and (bc_edit_limit is None or int(ed) <= bc_edit_limit):
        g = lambda t: _extract(sub, t)  # noqa: E731
        d["bc"] = dict(seq=g("bc="), ed=int(ed), ed_sec=None if g("ed_sec=") is None else int(g("ed_sec=")),
                       start=None if g("bcStart=") is None else int(g("bcStart=")),
                       end=None if g("bcEnd=") is None else int(g("bcEnd=")), rank=None if g("rk=") is None else int(g("rk=")))
    q = _extract(sub, "Q=")
    if q is not None:
        d["q"] = float(np.float32(q.split(" ")[0]))
    last = sub.rfind("_")
    # NumberToAndFromAscii.convertString (L492-494); a FASTQ-style name still carries ` cellBC=...` behind the id
    d["read_id"] = int(sub[last + 1:].split(" ")[0], 36) if last < len(sub) - 1 else 0
    return d


def parse_name(name):
    """fields of a scanfastq read name needed for the UMI step; None when the read carries no barcode.  The barcode is the
    bc= field (an aligner keeps only the first token of the FASTQ name, so ` cellBC=` is not in a BAM)."""
    try:
        d = scan_data_from_name(name)
    except _lib.SmiError:
        return None
    if d is None or d["bc"] is None or d["bc"]["seq"] is None or d["bc"]["end"] is None or d["x"] is None or d["q"] is None:
        return None
    return dict(cell=d["bc"]["seq"], ae=d["ae"], bc_end=d["bc"]["end"], x=d["x"].split(" ")[0], q=d["q"])


def umi_window(x, adapter_end, bc_end, five_prime=False, umi_len=12):
    """umi_len + 2 bases (14 with the shipped umis/umi_length) as 4-bit codes: the three umi_len-mers at offsets -1, 0, +1 behind the barcode
    (ClusteringEditDistanceBase L297-350; getStrandedShortSeqPosFromReadPos FastqRecordExt.java:L378).  3': on the reverse complement of X=
    (getSeqRevComp), barcode end = AE + 3 - bcEnd; 5': on X= itself (getSeq, L312-313), barcode end = bcEnd - AE + 3.  None if out of range"""
    if five_prime:
        pos = bc_end - adapter_end + 3
        if pos < 1 or pos + umi_len + 1 > len(x):
            return None
        return [_CODE.get(x[pos - 1 + k], 15) for k in range(umi_len + 2)]
    pos = adapter_end + 3 - bc_end
    if pos < 1 or pos + umi_len + 1 > len(x):
        return None
    return [_COMP[_CODE.get(x[len(x) - (pos + k)], 15)] for k in range(umi_len + 2)]


def pack_window(w):
    v = 0
    for k, c in enumerate(w):
        v |= c << (4 * k)
    return v


def assign_umis(ctx, names, positions, reverse, max_dist=500, cluster_cfg=None, n_threads=4, five_prime=False):
    """-> list (one per read) of None or dict(U8, U7, U1, U2, region, center) -- the tag VALUES the reference writes in
    ClusterOneBase.setSamflagsAndStatsForClustered; reads without barcode / position / neighbours get None"""
    n = len(names)
    info = [parse_name(nm) for nm in names]
    # every read whose name carries scan data has a clustering position, barcode or not (generateReadScanData L86-92)
    pos = [p if ("_REV_" in names[i] or "_FWD_" in names[i]) else None for i, p in enumerate(positions)]
    region, _ = _lib.region_group(pos, reverse, max_dist=max_dist, keep_data_end=False)
    return _assign_in_regions(ctx, info, region, cluster_cfg, n_threads, five_prime)


def _assign_in_regions(ctx, info, region, cluster_cfg=None, n_threads=4, five_prime=False):
    n = len(info)
    wins = [umi_window(f["x"], f["ae"], f["bc_end"], five_prime) if f is not None else None for f in info]
    groups = {}
    for i in range(n):
        if info[i] is not None and region[i] >= 0 and wins[i] is not None:
            groups.setdefault((info[i]["cell"], region[i]), []).append(i)  # canonical member order: input order
    groups = [g for g in groups.values() if len(g) > 1]  # UmiClustering.lambda$cluster$6
    out = [None] * n
    if not groups:
        return out
    order = [i for g in groups for i in g]
    sizes = [len(g) for g in groups]
    go, po, mo = ctx.umi_offsets(sizes)
    dev = torch.device("cuda", ctx.device)
    packed = np.array([pack_window(wins[i]) for i in order], dtype=np.uint64)
    d_out = torch.zeros(int(mo[-1]), dtype=torch.uint8, device=dev)
    ctx.umi_dist_device(torch.from_numpy(packed.view(np.int64)).to(dev), torch.from_numpy(go.view(np.int32)).to(dev),
                        torch.from_numpy(po.view(np.int64)).to(dev), torch.from_numpy(mo.view(np.int64)).to(dev),
                        len(sizes), int(po[-1]), d_out)
    torch.cuda.synchronize()
    qv = np.array([info[i]["q"] for i in order], dtype=np.float32)
    asg, skipped = _lib.umi_cluster_groups(d_out.cpu().numpy(), mo, go, qv, cluster_cfg, n_threads=n_threads)
    base = np.repeat(go[:-1], sizes)
    for j, i in enumerate(order):
        a = asg[j]
        if a["center"] < 0:
            if skipped is not None and skipped[j]:
                out[i] = dict(skipped=True)  # UMI_CLUSTERING_SKIPPED_HIGHCOMPLEXITY | DONT_ASSIGN_UMI (ClusterOneBase L61)
            continue
        cw = wins[order[int(base[j]) + int(a["center"])]]
        off = int(a["offset"])
        out[i] = dict(U8="".join(_DEC[c] for c in cw[off + 1:off + 13]), U7="".join(_DEC[c] for c in wins[i][1:13]),
                      U1=int(a["ed"]), U2=None if a["ed_second"] < 0 else int(a["ed_second"]), region=region[i],
                      center=order[int(base[j]) + int(a["center"])])
    return out


# ---- BAM input (BamReader.run, FJ!umifinder/bamreaders/BamReader.java:L106-158) ---------------------------------------------
def load_bam(data, n_threads=4):
    """BGZF bytes of a coordinate-sorted BAM -> (header text, [(reference, length)], inflated stream, record index)"""
    raw = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    bam, used = _lib.bgzf_inflate(raw, n_threads=n_threads)
    if used != raw.size:
        raise _lib.SmiError("truncated BGZF stream")
    text, refs, start = _lib.bam_header(bam)
    recs, end = _lib.bam_index_records(bam, start, cap=max(1, (bam.size - start) // 36))
    if end != bam.size:
        raise _lib.SmiError("truncated BAM record")
    return text, refs, bam, recs


def read_name(bam, rec):
    o, n = int(rec["name_off"]), int(rec["l_read_name"]) - 1
    return bam[o:o + n].tobytes().decode()


def clustering_position(bam, rec, scan, grouping_distance=100, five_prime=False, bc_length=16, umi_length=12):
    """NanoporeRead$ReadScanData.generateReadScanData / getGenomePosition (L86-116): reference position under read position
    polyA start - distanceFromReadEndForGrouping (3') or adapter end + cell_bc_length + umi_length + that distance (5', L90);
    None for unmapped reads and positions outside the alignment"""
    if scan is None or int(rec["flag"]) & 4:
        return None
    if not five_prime and scan["ps"] is None:
        return None  # the reference dereferences polyA_Result here: only reads scanned with --noPolyARequired lack it
    o = int(rec["cigar_off"])
    cigar = bam[o:o + 4 * int(rec["n_cigar"])].view("<u4")
    read_pos = scan["ae"] + bc_length + umi_length + grouping_distance if five_prime else scan["ps"] - grouping_distance
    return _lib.ref_position_at_read_position_raw(cigar, int(rec["pos"]) + 1, read_pos)


def assign_umis_bam(ctx, data, chunk_size=250_000, max_dist=500, bc_edit_limit=None, cluster_cfg=None, n_threads=4, native=False,
                    five_prime=False, batches=None, regions=None):
    """`assignumis` over a whole BAM -> (names, tags): tags[i] as assign_umis returns them, in BAM order.  Chunks as
    BamReader.run cuts them: `chunk_size` records or the end of a chromosome; within a chromosome the regions near the
    right edge are carried into the next chunk (ReadGrouper.groupSams keepDataEnd, smi_region_group).  batches / regions: optional lists that
    receive the record indices of every flush and {record index: run-wide region number} (genomicRegionNmber) of the grouped records."""
    _text, _refs, bam, recs = load_bam(data, n_threads=n_threads)
    n = recs.size
    names = [read_name(bam, r) for r in recs]
    if native:
        return names, _assign_umis_bam_native(ctx, bam, recs, names, chunk_size, max_dist, bc_edit_limit, n_threads, five_prime, batches, regions)
    scans = [scan_data_from_name(nm, bc_edit_limit) for nm in names]
    pos = [clustering_position(bam, recs[i], scans[i], five_prime=five_prime) for i in range(n)]
    rev = [bool(int(r["flag"]) & 16) for r in recs]
    info = []
    for d in scans:
        ok = d is not None and d["bc"] is not None and d["bc"]["seq"] and d["bc"]["end"] is not None and d["x"] and d["q"] is not None
        info.append(dict(cell=d["bc"]["seq"], ae=d["ae"], bc_end=d["bc"]["end"], x=d["x"], q=d["q"]) if ok else None)
    tags = [None] * n
    region_base = 0

    def flush(cur, keep):
        nonlocal region_base
        region, n_done = _lib.region_group([pos[i] for i in cur], [rev[i] for i in cur], max_dist=max_dist, keep_data_end=keep)
        done = cur[:n_done]
        if batches is not None:
            batches.append(list(done))  # what one OneBatchExecutor hands to the writers (sorted per batch, $BamWriters L421)
        res = _assign_in_regions(ctx, [info[i] for i in done], region[:n_done], cluster_cfg, n_threads, five_prime)
        for i, t in zip(done, res):
            if t is not None and not t.get("skipped"):
                t["center"] = done[t["center"]]
                t["region"] += region_base
            tags[i] = t
        if regions is not None:
            regions.update({i: region[k] + region_base for k, i in enumerate(done) if region[k] >= 0})
        region_base += (max(region[:n_done]) + 1) if n_done and max(region[:n_done]) >= 0 else 0
        return cur[n_done:]

    _run_chunks(recs, chunk_size, flush)
    return names, tags


def _run_chunks(recs, chunk_size, flush):
    """BamReader.run L106-147: a chunk ends after chunk_size records or with the chromosome; flush(cur, keep) returns the records
    ReadGrouper holds back for the next chunk"""
    n = recs.size
    cur, counter, chrom = [], 0, None
    for i in range(n):
        ref = int(recs[i]["ref_id"])
        if i == 0:
            chrom, cur, counter = ref, [0], 1
            continue
        counter += 1
        is_end = ref != chrom
        if is_end:
            chrom = ref
        if counter >= chunk_size or is_end:
            cur = flush(cur, keep=not is_end)
            counter = 0
        cur.append(i)
    while cur:
        cur = flush(cur, keep=False)


def _assign_umis_bam_native(ctx, bam, recs, names, chunk_size, max_dist, bc_edit_limit, n_threads, five_prime=False, batches=None, regions=None):
    """the same through smi_assignumis_chunk: one native call per chunk (name parsing, positions, grouping, K-UMI, clustering)"""
    tags = [None] * recs.size
    cig = [bam[int(r["cigar_off"]):int(r["cigar_off"]) + 4 * int(r["n_cigar"])].view("<u4") for r in recs]
    region_base = 0

    def flush(cur, keep):
        nonlocal region_base
        out, n_done = ctx.assignumis_chunk([names[i] for i in cur], recs["flag"][cur], recs["pos"][cur], [cig[i] for i in cur],
                                           keep_data_end=keep, max_dist=max_dist, bc_edit_limit=bc_edit_limit, n_threads=n_threads,
                                           five_prime=five_prime)
        if batches is not None:
            batches.append(list(cur[:n_done]))
        top = -1
        for k in range(n_done):
            t, i = out[k], cur[k]
            top = max(top, int(t["region"]))
            if regions is not None and int(t["region"]) >= 0:
                regions[i] = int(t["region"]) + region_base
            if t["flags"] & _lib.UMI_CLUSTERED:
                tags[i] = dict(U8=t["u8"].decode(), U7=t["u7"].decode(), U1=int(t["u1"]), U2=None if t["u2"] < 0 else int(t["u2"]),
                               region=int(t["region"]) + region_base, center=cur[int(t["center"])])
            elif t["flags"] & _lib.UMI_SKIPPED:
                tags[i] = dict(skipped=True)
        region_base += top + 1
        return cur[n_done:]

    _run_chunks(recs, chunk_size, flush)
    return tags


# ---- BAM tags (ReadScanResult.writeSamFlags / writeBCSamFlags, ClusterOneBase.setSamflagsAndStatsForClustered,
#      UmiFinderWorker.lambda$new$0 + $BamWriters.lambda$writeSams$2; tag names: Jar/config.xml:297-492) -------------------------
def record_tag_sets(scan, umi, u7, gene=None, gene_tag="GE"):
    """the setAttribute calls the reference makes on one record, in its order: [(tag, value)], value int, str or None (= the tag is removed).
    scan: scan_data_from_name; umi: entry of assign_umis (or None); u7: the read's own post-barcode 12-mer or None; gene: (GE, GS, XF) of
    lib.GeneTagger for this record when an --annotationFile is given.
    -> (calls, has_bc, umi_from_clustering)"""
    c = []
    if scan is None:
        return c, False, False
    if scan["pe"] is not None:                                   # polyAFound(): end != null (ReadScanResult L205-207)
        c += [("PE", scan["pe"]), ("PS", scan["ps"])]
    c.append(("AE", scan["ae"]))                                 # L209-210
    if scan["reverse"]:
        c.append(("RE", ""))                                     # L212-213
    if scan["tso"] is not None:
        c.append(("TE", scan["tso"]))                            # L215-216
    bc = scan["bc"]
    has_bc = bc is not None and bc["seq"] is not None            # barcodeFound(): barcodeseq != null
    if has_bc:                                                   # L218-237
        c.append(("BU", bc["seq"]))
        if bc["start"] is not None:
            c.append(("BV", str(bc["start"])))
        if bc["end"] is not None:
            c.append(("BE", str(bc["end"])))
        c.append(("BW", bc["ed"]))
        if bc["ed_sec"] is not None:
            c.append(("BX", "N.A." if bc["ed_sec"] == 2147483647 else str(bc["ed_sec"])))
        c.append(("SX", str(scan["read_id"])))
        if bc["rank"] is not None:
            c.append(("BH", str(bc["rank"])))
        # writeBCSamFlags(sam, flags, false, false) L254-279
        c += [("BC", bc["seq"]), ("BZ", "")]
        if bc["start"] is not None:
            c.append(("BB", str(bc["start"])))
        if bc["end"] is not None:
            c.append(("BF", str(bc["end"])))
        c.append(("B1", bc["ed"]))
        if bc["ed_sec"] is not None:
            c.append(("B2", str(bc["ed_sec"])))
        c.append(("BZ", bc["seq"]))
        if bc["rank"] is not None:
            c.append(("BH", str(bc["rank"])))
    if gene is not None and gene[2] is not None:                 # annotateGene (OneNanoporeSeqAnalyzer L98, GennameTagger.setGeneExons L108-119)
        ge, gs, xf = gene
        c.append(("XF", xf))
        c += [(gene_tag, ge), ("GS", gs)] if ge is not None and gs is not None else [(gene_tag, None), ("GS", None)]
    clustered = umi is not None and not umi.get("skipped")
    if clustered:                                                # ClusterOneBase L145-164
        c += [("U8", umi["U8"]), ("U7", umi["U7"]), ("UC", ""), ("U1", str(umi["U1"]))]
        if umi["U2"] is not None:
            c.append(("U2", str(umi["U2"])))
    elif has_bc and u7 is not None:                              # UmiFinderWorker.lambda$new$0 L248-255
        c.append(("U7", u7))
        if not (umi is not None and umi.get("skipped")):         # $BamWriters L442-447: unless DONT_ASSIGN_UMI
            c += [("U8", u7), ("UZ", "")]
    return c, has_bc, clustered


def _aux_bytes(tag, value):
    """htsjdk BinaryTagCodec.writeTag / getIntegerType (Jar/lib/htsjdk-4.1.3.jar!/htsjdk/samtools/BinaryTagCodec.class L123-180, read from the
    class file): strings as Z; an integer as c in [-128, 127], C up to 255, s up to 32767 (and down to -32768), S up to 65535, i up to
    2^31 - 1 (and down to -2^31), I above"""
    t = tag.encode()
    if isinstance(value, str):
        return t + b"Z" + value.encode() + b"\0"
    v = int(value)
    for code, fmt, lo, hi in (("c", "<b", -128, 127), ("C", "<B", 0, 255), ("s", "<h", -32768, 32767), ("S", "<H", 0, 65535),
                              ("i", "<i", -2 ** 31, 2 ** 31 - 1), ("I", "<I", 0, 2 ** 32 - 1)):
        if lo <= v <= hi:
            return t + code.encode() + np.array([v]).astype(fmt).tobytes()
    raise ValueError("integer tag out of range")


_AUX_SIZE = {ord("A"): 1, ord("c"): 1, ord("C"): 1, ord("s"): 2, ord("S"): 2, ord("i"): 4, ord("I"): 4, ord("f"): 4}


def split_aux(aux):
    """aux bytes of a BAM record -> [(tag str, raw bytes of the whole field)] in file order"""
    out, p, n = [], 0, len(aux)
    while p < n:
        ty = aux[p + 2]
        if ty in _AUX_SIZE:
            q = p + 3 + _AUX_SIZE[ty]
        elif ty in (ord("Z"), ord("H")):
            q = aux.index(b"\0", p + 3) + 1
        elif ty == ord("B"):
            q = p + 8 + _AUX_SIZE[aux[p + 3]] * int(np.frombuffer(aux[p + 4:p + 8], dtype="<u4")[0])
        else:
            raise _lib.SmiError("unknown BAM aux type")
        out.append((aux[p:p + 2].decode(), bytes(aux[p:q])))
        p = q
    return out


_INT_FMT = {ord("c"): "<i1", ord("C"): "<u1", ord("s"): "<i2", ord("S"): "<u2", ord("i"): "<i4", ord("I"): "<u4"}


def _as_htsjdk_writes(tag, raw):
    """a field read from the input as htsjdk writes it back: an integer in the smallest type that holds it (BinaryTagCodec.readSingleValue
    L316-346 boxes it, writeTag re-picks the type with getIntegerType L153-180); a hex string (H) comes back as a byte array (B:c); the rest
    unchanged"""
    ty = raw[2]
    if ty in _INT_FMT:
        return _aux_bytes(tag, int(np.frombuffer(raw[3:], dtype=_INT_FMT[ty])[0]))
    if ty == ord("H"):
        data = bytes.fromhex(raw[3:-1].decode())
        return raw[:2] + b"Bc" + np.array([len(data)], dtype="<u4").tobytes() + data
    return raw


def apply_tag_sets(fields, calls):
    """The attribute list of a record that htsjdk reads, tags and writes.  BinaryTagCodec.readTags (Jar/lib/htsjdk-4.1.3.jar!/htsjdk/samtools/
    BinaryTagCodec.class L271-305) builds the list through SAMBinaryTagAndValue.insert (L207-228), so it is ORDERED BY BINARY TAG from the
    moment it is decoded -- binary tag = (short)(second char << 8 | first char), SAMTag.makeBinaryTag L124-127 -- and a tag that occurs twice
    keeps its last value; SAMRecord.setAttribute (L1583-1602) inserts / replaces in that order and removes on null.  Every record goes through
    setAttribute here, so every written record has its tags in that order.  fields: [(tag, raw field bytes)] of the input record ->
    [(tag, raw bytes)] as written.  (Executed: tests/golden/ref_exec_auxorder.json.)"""
    key = lambda t: (ord(t[1]) << 8) | ord(t[0])  # noqa: E731
    cur = {}
    for tag, raw in fields:
        cur[tag] = _as_htsjdk_writes(tag, raw)
    for tag, value in calls:
        if value is None:                                        # setAttribute(tag, null): SAMBinaryTagAndValue.remove
            cur.pop(tag, None)
        else:
            cur[tag] = _aux_bytes(tag, value)
    return sorted(cur.items(), key=lambda kv: key(kv[0]))


def _coordinate_key(rec, name):
    """htsjdk SAMRecordCoordinateComparator.compare (SAMRecordCoordinateComparator.java:L48-105): reference index with -1 last -- two records
    without a reference are not compared by position --, alignment start, strand (forward first), name, flags, mapping quality, then the mate's
    reference index, the mate's start and the insert size as plain integers (a mate index of -1 sorts FIRST there)"""
    ref = int(rec["ref_id"])
    return (ref if ref >= 0 else 1 << 30, int(rec["pos"]) if ref >= 0 else 0, bool(int(rec["flag"]) & 16), name, int(rec["flag"]), int(rec["mapq"]),
            int(rec["next_ref_id"]), int(rec["next_pos"]), int(rec["tlen"]))


def _java_split(text, sep):
    """String.split(sep) for a literal separator: trailing empty strings are dropped, the empty string gives [""]"""
    if text == "":
        return [""]
    parts = text.split(sep)
    while parts and parts[-1] == "":
        parts.pop()
    return parts


def _count_columns(rows, five_prime):
    """rows of one written batch -> the columns of lib.GeneCounts.add"""
    cols = list(zip(*rows))
    return dict(gene=cols[0], region=cols[1], cell_bc=cols[2], umi=cols[3], has_bc_umi=cols[4], flag=cols[5], mapq=cols[6], first_cigar=cols[7],
                last_cigar=cols[8], nth_record=cols[9], five_prime=five_prime)


def tagged_record(bam, r, name, scan, umi, gene=None, five_prime=False, truncate_read_name=False, gene_tag="GE"):
    """one record of the output: (BAM record bytes incl. its block_size word, from clustering?, attribute fields) or None when the read has no
    cell barcode (it is not written).  scan: scan_data_from_name(name); umi: the record's entry of assign_umis / None; gene: (GE, GS, XF) or None"""
    u7 = None
    if scan is not None and scan["bc"] is not None and scan["bc"]["end"] is not None and scan["x"]:
        w = umi_window(scan["x"], scan["ae"], scan["bc"]["end"], five_prime)
        u7 = None if w is None else "".join(_DEC[c] for c in w[1:13])
    calls, has_bc, clustered = record_tag_sets(scan, umi, u7, gene, gene_tag)
    if not has_bc:
        return None
    o = int(r["rec_off"])
    fixed = bytearray(bam[o + 4:int(r["aux_off"])].tobytes())
    if truncate_read_name:                                  # -w: readName.split("_")[0] (L431-432)
        nm = name.split("_")[0].encode() + b"\0"
        fixed = fixed[:32] + nm + fixed[32 + int(r["l_read_name"]):]
        fixed[8] = len(nm)
    aux = bam[int(r["aux_off"]):int(r["aux_off"]) + int(r["aux_len"])].tobytes()
    fields = apply_tag_sets(split_aux(aux), calls)
    body = bytes(fixed) + b"".join(raw for _, raw in fields)
    return np.array([len(body)], dtype="<u4").tobytes() + body, clustered, fields


def gene_count_row(bam, r, fields, region, nth, gene_tag="GE"):
    """the columns of lib.GeneCounts.add for one written record, or None when it carries no U8 (updateGeneCounts is only called for records
    with the UMI attribute, UmiFinderWorker.java:L453)"""
    f = {t: raw for t, raw in fields if t in (gene_tag, "U8", "BC")}
    z = lambda t: f[t][3:-1].decode() if t in f and f[t][2:3] == b"Z" else None  # noqa: E731
    ge, u8, bc = z(gene_tag), z("U8"), z("BC")
    if u8 is None:
        return None
    cg = bam[int(r["cigar_off"]):int(r["cigar_off"]) + 4 * int(r["n_cigar"])].view("<u4")
    gene = None if ge is None else (_java_split(ge, ",") or [None])[0]
    return (gene, region, _lib.two_bit_code(bc) if bc else 0, _lib.two_bit_code(u8), 1 if bc is not None else 0, int(r["flag"]), int(r["mapq"]),
            int(cg[0]) if cg.size else 0xFFFFFFFF, int(cg[-1]) if cg.size else 0, 1 if nth else 0)


def write_tagged_bams(ctx, data, chunk_size=250_000, truncate_read_name=False, compress_level=5, n_threads=4, refflat=None, bgzf="device",
                      gene_counts=None, gene_tag="GE", **kw):
    """`assignumis` BAM in -> (bcfound BAM bytes, umifound BAM bytes, names, tags): the two BGZF streams the reference writes
    (<out>.bam: every record with a cell barcode; <out>_umifound_.bam: those whose UMI comes from clustering), header copied,
    records of a chunk in coordinate-comparator order with the tags of record_tag_sets added; refflat = text of the --annotationFile
    (GE / GS / XF through lib.GeneTagger), None = no annotation file given.  bgzf: "device" = the BGZF blocks are deflated by K-DEFLATE
    (smi_bgzf_deflate_device), "zlib" = by zlib at compress_level on n_threads host threads; the inflated streams are the same.
    gene_counts: a lib.GeneCounts that receives every written record that ends up with a U8 tag, batch by batch
    (GeneCounts.updateGeneCounts from $BamWriters.lambda$writeSams$2 L453-454) -- what <out>.genecounts.tsv / <out>.UMIdepths.tsv print."""
    _text, _refs, bam, recs = load_bam(data, n_threads=n_threads)
    gene_tags = None
    if refflat is not None:
        tagger = _lib.GeneTagger(refflat, [nm for nm, _ in _refs])
        gene_tags = tagger.tag_bam(bam, recs)
        tagger.close()
    batches, regions = [], ({} if gene_counts is not None else None)
    names, tags = assign_umis_bam(ctx, data, chunk_size=chunk_size, n_threads=n_threads, batches=batches, regions=regions, **kw)
    five_prime = bool(kw.get("five_prime", False))
    scans = [scan_data_from_name(nm, kw.get("bc_edit_limit")) for nm in names]
    header_end = int(recs[0]["rec_off"]) if recs.size else bam.size
    out_bc, out_umi = [bam[:header_end].tobytes()], [bam[:header_end].tobytes()]
    # BamReader cuts chunks; BamWriters sorts each written BATCH with the coordinate comparator (L421), not the file: batches =
    # the flushes of assign_umis_bam, in the order they complete
    order = [i for b in batches for i in sorted(b, key=lambda k: _coordinate_key(recs[k], names[k]))]
    nth = {}
    if gene_counts is not None:                                 # OneNanoporeSeqAnalyzer.call L74-80: a record of this read name was analysed before
        seen = set()
        for b in batches:
            for i in b:
                nth[i] = names[i] in seen
                seen.add(names[i])
    rows = []                                                   # in write order: UMIcounts.increment does not commute with its nth-record form
    for i in order:
        res = tagged_record(bam, recs[i], names[i], scans[i], tags[i], None if gene_tags is None else gene_tags[i], five_prime, truncate_read_name, gene_tag)
        if res is None:
            continue
        rec_bytes, clustered, fields = res
        out_bc.append(rec_bytes)
        if clustered:
            out_umi.append(rec_bytes)
        if gene_counts is not None:
            row = gene_count_row(bam, recs[i], fields, regions.get(i, -1), nth[i], gene_tag)
            if row is not None:
                rows.append(row)
    if gene_counts is not None and rows:
        gene_counts.add(**_count_columns(rows, five_prime))
    if bgzf == "device":
        z = lambda parts: ctx.bgzf_deflate_device(b"".join(parts)).tobytes()  # noqa: E731
    else:
        z = lambda parts: _lib.bgzf_deflate(b"".join(parts), level=compress_level, n_threads=n_threads).tobytes()  # noqa: E731
    return z(out_bc), z(out_umi), names, tags


def chunk_bounds(ref_ids, chunk_size):
    """BamReader.run L106-147 as positions: [(index in front of which a flush happens, keep)] -- after chunk_size records (the first chunk
    holds chunk_size - 1: the reader's counter starts at 1) or in front of the first record of another chromosome (keep = False)"""
    ref = np.asarray(ref_ids)
    out, n = [], int(ref.size)
    if n == 0:
        return out
    change = (np.flatnonzero(ref[1:] != ref[:-1]) + 1).tolist()
    step, i0, ci = max(int(chunk_size), 1), -1, 0        # i0: the last flush position (the reader's counter is 1 at the first record)
    while True:
        by_count = max(i0 + step, 1)
        while ci < len(change) and change[ci] <= i0:
            ci += 1
        nxt = change[ci] if ci < len(change) else n
        at = min(by_count, nxt)
        if at >= n:
            return out
        out.append((at, at != nxt))
        i0 = at


def write_tagged_bams_native(ctx, data, chunk_size=250_000, truncate_read_name=False, compress_level=5, n_threads=4, refflat=None, bgzf="device",
                             gene_counts=None, max_dist=500, bc_edit_limit=None, five_prime=False, cluster_cfg=None, gene_tag="GE", umi_length=0,
                             grouping_distance=None):
    """write_tagged_bams with no per-record work in Python: BGZF inflate + record index (host threads), per BamReader chunk
    smi_bam_chunk_inputs -> smi_assignumis_chunk (device), smi_gene_tag_bam, per written batch smi_bam_write_batch (host threads), BGZF by
    K-DEFLATE.  -> (bcfound BAM, umifound BAM -- numpy uint8 arrays --, info dict).  The same bytes as write_tagged_bams."""
    import time

    t0 = time.perf_counter()
    secs = dict(inflate_index=0.0, gene_tagger=0.0, chunk_inputs=0.0, umi_stage=0.0, write_batch=0.0, bgzf=0.0)
    _text, refs, bam, recs = load_bam(data, n_threads=n_threads)
    secs["inflate_index"] = time.perf_counter() - t0
    n = int(recs.size)
    gene = None
    if refflat is not None:
        t1 = time.perf_counter()
        tagger = _lib.GeneTagger(refflat, [nm for nm, _ in refs])
        gene = tagger.tag_bam_raw(bam, recs)
        tagger.close()
        secs["gene_tagger"] = time.perf_counter() - t1
    header_end = int(recs[0]["rec_off"]) if n else bam.size
    cap = header_end + (_lib.bam_write_bound(recs) if n else 0)
    buf_bc, buf_umi = np.empty(cap, dtype=np.uint8), np.empty(cap, dtype=np.uint8)     # both outputs are written in place, batch behind batch
    buf_bc[:header_end] = bam[:header_end]
    buf_umi[:header_end] = bam[:header_end]
    at_bc = at_umi = header_end
    tags = np.zeros(max(n, 1), dtype=_lib.UMI_TAG_DTYPE)
    region = np.full(max(n, 1), -1, dtype=np.int64)
    nth = _lib.bam_name_seen(bam, recs) if gene_counts is not None and n else np.zeros(max(n, 1), dtype=np.uint8)
    region_base, n_clustered, n_batches = 0, 0, 0
    cur = np.zeros(0, dtype=np.int32)

    def flush(cur, keep):
        nonlocal region_base, n_clustered, n_batches, at_bc, at_umi
        t1 = time.perf_counter()
        inp = _lib.bam_chunk_inputs(bam, recs, cur)
        t2 = time.perf_counter()
        out, n_done = ctx.assignumis_chunk_raw(inp, keep_data_end=keep, max_dist=max_dist, bc_edit_limit=bc_edit_limit, n_threads=n_threads,
                                               five_prime=five_prime, cluster_cfg=cluster_cfg, umi_length=umi_length, grouping_distance=grouping_distance)
        t3 = time.perf_counter()
        secs["chunk_inputs"] += t2 - t1
        secs["umi_stage"] += t3 - t2
        done = cur[:n_done]
        tags[done] = out[:n_done]
        reg = out["region"][:n_done].astype(np.int64)
        region[done] = np.where(reg >= 0, reg + region_base, -1)
        region_base += int(reg.max()) + 1 if n_done and reg.max() >= 0 else 0
        n_clustered += int(((out["flags"][:n_done] & _lib.UMI_CLUSTERED) != 0).sum())
        bc, umi, _order = _lib.bam_write_batch(bam, recs, done, tags, gene=gene, bc_edit_limit=bc_edit_limit, truncate_read_name=truncate_read_name,
                                               five_prime=five_prime, n_threads=n_threads, gene_counts=gene_counts, region=region, nth_record=nth,
                                               out_bc=buf_bc[at_bc:], out_umi=buf_umi[at_umi:], gene_tag=gene_tag)
        at_bc += bc.size
        at_umi += umi.size
        n_batches += 1
        secs["write_batch"] += time.perf_counter() - t3
        return cur[n_done:]

    start = 0
    for at, keep in chunk_bounds(recs["ref_id"], chunk_size) if n else []:
        cur = flush(np.concatenate([cur, np.arange(start, at, dtype=np.int32)]), keep)
        start = at
    if n:
        cur = np.concatenate([cur, np.arange(start, n, dtype=np.int32)])
        while cur.size:
            cur = flush(cur, False)
    if bgzf == "device":
        z = lambda a: ctx.bgzf_deflate_device(a)  # noqa: E731
    else:
        z = lambda a: _lib.bgzf_deflate(a, level=compress_level, n_threads=n_threads)  # noqa: E731
    t1 = time.perf_counter()
    z_bc, z_umi = z(buf_bc[:at_bc]), z(buf_umi[:at_umi])                                 # numpy uint8 arrays (bytes(...) / tofile)
    secs["bgzf"] = time.perf_counter() - t1
    return z_bc, z_umi, dict(records=n, clustered=n_clustered, batches=n_batches, tags=tags, region=region, seconds=secs,
                             wall_s=time.perf_counter() - t0, bam_bytes=int(bam.size))


def segment_cuts(ref, cur_n, g0, i0, prev_ref, step):
    """BamReader.run's cuts among the records of one segment of a streamed BAM.  ref: reference index of the segment's records (local index
    k = global index g0 + k); the first cur_n of them are already in the chunk being collected; i0: global position of the last cut (-1:
    none yet); prev_ref: reference of the record in front of the segment (None at the start of the file); step: chunk size.
    -> ([(local position in front of which the chunk is closed, keep)], i0 afterwards): closed after `step` records, or in front of the first
    record of another chromosome (keep = False: nothing is held back for the next chunk)"""
    m = int(len(ref))
    change = (np.flatnonzero(ref[1:] != ref[:-1]) + 1).tolist() if m > 1 else []
    if m and prev_ref is not None and cur_n == 0 and g0 > 0 and int(ref[0]) != prev_ref:
        change = [0] + change
    change = [c for c in change if c >= cur_n]             # (changes inside the chunk being collected were cut when they were met)
    cuts, ci = [], 0
    while True:
        by_count = max(i0 + step, 1) - g0
        while ci < len(change) and g0 + change[ci] <= i0:
            ci += 1
        nxt = change[ci] if ci < len(change) else m
        at = min(by_count, nxt)
        if at >= m:
            return cuts, i0
        cuts.append((at, at != nxt))
        i0 = g0 + at


_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bai_ref_extents(bai_path):
    """samtools' / htsjdk's BAM index -> per reference (first virtual offset, virtual offset behind its last record) or None (no records):
    the metadata pseudo-bin 37450 where the writer left one, else the least chunk start / largest chunk end over the reference's bins"""
    return bai_ref_extents_bytes(open(bai_path, "rb").read(), bai_path)


def bai_ref_extents_bytes(b, what="BAM index"):
    import struct

    if b[:4] != b"BAI\x01":
        raise _lib.SmiError(f"{what}: not a BAM index")
    n_ref, at = struct.unpack_from("<i", b, 4)[0], 8
    out = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", b, at)[0]
        at += 4
        lo, hi, meta = None, None, None
        for _b in range(n_bin):
            bin_id, n_chunk = struct.unpack_from("<Ii", b, at)
            at += 8
            for c in range(n_chunk):
                cb, ce = struct.unpack_from("<QQ", b, at)
                at += 16
                if bin_id == 37450:
                    if c == 0:
                        meta = (cb, ce)
                    continue
                lo = cb if lo is None else min(lo, cb)
                hi = ce if hi is None else max(hi, ce)
        n_intv = struct.unpack_from("<i", b, at)[0]
        at += 4 + 8 * n_intv
        out.append(meta if meta is not None and meta[1] > meta[0] else (None if lo is None else (lo, hi)))
    return out


def plan_shards(extents, world):
    """whole references dealt to the ranks in file order, balanced by compressed bytes -> per rank (first virtual offset or None = the file's
    first record, end virtual offset or None = end of file: the unmapped tail goes with the last reference).  With fewer references than
    ranks the first ranks get one each and the others nothing: (0, 0)."""
    refs = [e for e in extents if e is not None]
    m = max(1, min(world, len(refs)))
    size = [max(1, (e[1] >> 16) - (e[0] >> 16)) for e in refs]
    total, starts, acc, k = sum(size), [0], 0, 0
    for r in range(1, m):                 # rank r starts at the first reference whose preceding bytes reach r / m of the whole
        while k < len(refs) - (m - r) and (acc < total * r / m or k < r):
            acc += size[k]
            k += 1
        k = max(k, starts[-1] + 1)
        starts.append(k)
    out = []
    for r in range(world):
        if r >= m:
            out.append((0, 0))
        else:
            out.append((None if r == 0 else refs[starts[r]][0], None if r == m - 1 else refs[starts[r + 1]][0]))
    return out


def assignumis_stream(ctx, in_bam, out_prefix, segment_bytes=256 << 20, chunk_size=250_000, truncate_read_name=False, n_threads=4, refflat=None,
                      max_dist=500, bc_edit_limit=None, five_prime=False, cluster_cfg=None, bc_length=16, group=None, shard=None, no_clustering=False,
                      gene_tag="GE", umi_length=0, grouping_distance=None, random_umi_seed=0, simulate=False):
    """`assignumis -i in.bam -o out` for a BAM of any size: the file is read in segments of about segment_bytes compressed bytes (read and inflated by a thread of their own, one segment ahead), never held as a
    whole -- inflate the segment's complete BGZF blocks behind the records still pending, index, cut BamReader's chunks (the counter and the
    chromosome carry over the segment borders), per chunk smi_assignumis_chunk + smi_bam_write_batch, each written batch BGZF-deflated on the
    device and appended to <out>.bam / <out>_umifound_.bam; the records ReadGrouper holds back and the unfinished chunk go in front of the
    next segment.  Writes the four files of assignumis_files; the inflated streams and the tables equal the whole-file run's.  -> info dict

    Several GPUs ("partition by chromosome ... no collective, only a final merge of gene counts"): with torch.distributed
    initialised (one process per GPU; `group`) -- or shard=(rank, world) without it -- the references of the BAM are dealt to the ranks as whole
    chromosomes in file order, balanced by compressed bytes, from <in_bam>.bai (samtools index, as quickrun-2.1.sh:39 leaves it).  Every rank
    reads only its byte range, runs the same pipeline on it and writes <out>.bam.shard<r> / <out>_umifound_.bam.shard<r>; rank 0 then strings
    the shards together (BGZF streams concatenate) and merges the gene tables (smi_gene_counts_merge_shard).  What stays the same as in one
    process: the (cell, region) groups, the tags of every record, the order of the records, both tables.  What may differ: where the batches
    are cut (BamReader's record counter starts anew on every rank, BamReader.java:L106-158; a batch is a unit of writing only), region NUMBERS
    (equal regions, other ids: every rank numbers from rank << 40), and -- for a read with alignments on chromosomes of two ranks -- the
    "further alignment" bit of UMIcounts in the later rank (it sees the name for the first time).

    simulate (-e / -f, the accuracy simulations: UmiFinderWorker$BamWriters.java:L294, L412): the run goes through every stage and writes NO file -- its
    counts are the result; random_umi_seed (-f): every read's UMI window is a random one (smi_assignumis_config.random_umi_seed), so what still clusters is
    chance.  One process only.

    no_clustering (-s, UmiFinderMain.java:L268-269): OneBatchExecutor.call skips UmiClustering.cluster (L83) and the call-back skips the U7 fill
    of barcoded records (UmiFinderWorker$FutCallBack.onSuccess L209-214), so no record gets U7 / U8 / UC / UZ / U1 / U2; regions, scan tags,
    BC and gene tags are as always (ReadGrouper runs in the reader).  The stage still runs -- its regions and hold-back counts are needed --
    and its UMI results are dropped."""
    import time

    t_all = time.perf_counter()
    rank, world, dist = 0, 1, None
    if shard is not None:
        rank, world = int(shard[0]), int(shard[1])
    else:
        try:
            import torch.distributed as dist_mod
            if dist_mod.is_available() and dist_mod.is_initialized() and dist_mod.get_world_size(group) > 1:
                dist, rank, world = dist_mod, dist_mod.get_rank(group), dist_mod.get_world_size(group)
        except ImportError:
            pass
    gc, names_seen = _lib.GeneCounts(), _lib.NameSet()
    f_in = open(in_bam, "rb")
    v_begin = v_end = None                          # this rank's range of the file in virtual offsets (None: from the first record / to the end)
    pend = np.zeros(0, dtype=np.uint8)             # inflated bytes not consumed yet: pending records (+ a partial record)
    if world > 1:
        bai = next((p_ for p_ in (in_bam + ".bai", in_bam[:-4] + ".bai") if os.path.isfile(p_)), None)
        if bai is None:
            raise _lib.SmiError(f"assignumis over {world} ranks needs the BAM index ({in_bam}.bai: samtools index) to deal whole chromosomes to the ranks")
        extents = bai_ref_extents(bai)
        size = os.path.getsize(in_bam)
        far = max((e[1] >> 16 for e in extents if e is not None), default=0)
        if far > size:
            raise _lib.SmiError(f"{bai}: the index points behind the end of {in_bam} (block at {far}, file of {size} bytes): stale index or truncated BAM")
        head = np.fromfile(f_in, dtype=np.uint8, count=4 << 20)
        hb, _used = _lib.bgzf_inflate(head, n_threads=1)
        _t, h_refs, hlen = _lib.bam_header(hb)
        if len(h_refs) != len(extents):
            raise _lib.SmiError(f"{bai}: index of {len(extents)} references, header of {in_bam} has {len(h_refs)}: not this file's index")
        f_in.seek(0)
        v_begin, v_end = plan_shards(extents, world)[rank]
        if rank > 0 and (v_begin, v_end) != (0, 0):
            # the header comes from the file's start; it leads this rank's stream (and is not written again: rank 0 wrote it)
            pend = hb[:hlen].copy()
    suffix = f".shard{rank}" if world > 1 else ""
    if simulate and world > 1:
        raise _lib.SmiError("the accuracy simulations (-e / -f) run in one process")
    f_bc, f_umi = (open(os.devnull, "wb"), open(os.devnull, "wb")) if simulate else (open(out_prefix + ".bam" + suffix, "wb"), open(out_prefix + "_umifound_.bam" + suffix, "wb"))
    nth_pend = np.zeros(0, dtype=np.uint8)
    header, tagger, refs = None, None, None
    i0, g0, prev_ref = -1, 0, None                 # global position of the last flush, global index of pend's first record, reference of the record in front
    region_base = n_records = n_clustered = n_batches = 0
    region_base = rank << 40                       # region numbers only matter for equality: every rank numbers from its own base
    cur_n = 0                                      # how many of pend's leading records belong to the chunk being collected
    step = max(int(chunk_size), 1)
    secs = dict(read_inflate=0.0, index=0.0, umi_stage=0.0, write_batch=0.0, bgzf_write=0.0)

    stage = {}                                     # page-locked buffers reused from batch to batch: records of both outputs, their BGZF streams

    def staged(key, n):
        pb = stage.get(key)
        if pb is None or pb.array.size < n:
            if pb is not None:
                pb.close()
            pb = stage[key] = _lib.PinnedBuffer(int(n * 1.2) + (1 << 20))
        return pb.array

    def emit(bc, umi):
        import threading

        t1 = time.perf_counter()
        writers = []
        for key, fh, a in (("z_bc", f_bc, bc), ("z_umi", f_umi, umi)):
            if a.size:
                z = ctx.bgzf_deflate_device(a, out=staged(key, ctx.bgzf_device_bound(a.size)))
                # without the end-of-file block: more batches follow.  The file is written by a thread of its own while the other output
                # is deflated (a write to the page cache is a memcpy by one core)
                writers.append(threading.Thread(target=fh.write, args=(z[:-28],)))
                writers[-1].start()
        for w in writers:
            w.join()
        secs["bgzf_write"] += time.perf_counter() - t1

    # a reader thread reads and inflates the next segment while this one is processed; it leaves room in front of the inflated bytes for the
    # pending tail, which only the consumer knows
    import queue
    import threading

    room_hint = [256 << 20]
    segments = queue.Queue(maxsize=1)

    def reader():
        tail = np.zeros(0, dtype=np.uint8)
        try:
            skip = 0                                # bytes of the first inflated block in front of this rank's first record
            if v_begin is not None:
                f_in.seek(v_begin >> 16)
                skip = v_begin & 0xFFFF
            left = None if v_end is None else (v_end >> 16) - f_in.tell()   # compressed bytes of whole blocks still to read
            while True:
                want = int(segment_bytes) if left is None else min(int(segment_bytes), left)
                raw = np.fromfile(f_in, dtype=np.uint8, count=want)
                if left is not None:
                    if raw.size < want:              # the index promises bytes the file does not have (stale .bai, truncated BAM)
                        raise _lib.SmiError(f"{in_bam}: truncated BGZF stream / the BAM index does not match the file "
                                            f"({want - raw.size} bytes short of virtual offset {v_end})")
                    left -= raw.size
                last = raw.size < segment_bytes if left is None else left == 0
                comp = np.concatenate([tail, raw]) if tail.size else raw
                buf, room, used = _lib.bgzf_inflate(comp, n_threads=n_threads, room=room_hint[0]) if comp.size else (np.zeros(0, dtype=np.uint8), 0, 0)
                tail = comp[used:].copy()
                if last and tail.size:
                    raise _lib.SmiError("truncated BGZF stream")
                if skip:
                    room += skip                     # (the block at v_begin holds the end of the previous chromosome in front of it)
                    skip = 0
                if last and v_end is not None and (v_end & 0xFFFF):
                    # the block the range ends in: its first bytes are this rank's last record(s), the rest is the next rank's
                    hdr = np.fromfile(f_in, dtype=np.uint8, count=18)
                    if hdr.size < 18 or hdr[12] != 66 or hdr[13] != 67:
                        raise _lib.SmiError("BGZF block without a leading BC field at a shard border")
                    bsize = int(hdr[16]) | (int(hdr[17]) << 8)
                    blk = np.concatenate([hdr, np.fromfile(f_in, dtype=np.uint8, count=bsize + 1 - 18)])
                    tb, _u = _lib.bgzf_inflate(blk, n_threads=1)
                    buf = np.concatenate([buf, tb[:v_end & 0xFFFF]])
                segments.put((buf, room, last, None))
                if last:
                    return
        except BaseException as e:  # noqa: BLE001 -- handed to the consumer
            segments.put((None, 0, True, e))

    # more ranks than chromosomes with reads: this rank has no range.  It writes two empty shards and goes straight to the exchange
    # (no header is read, so nothing may ask for one)
    idle = world > 1 and (v_begin, v_end) == (0, 0)
    if not idle:
        threading.Thread(target=reader, daemon=True).start()
    eof = idle
    while not eof or pend.size:
        t1 = time.perf_counter()
        buf, room, eof, err = segments.get() if not eof else (np.zeros(0, dtype=np.uint8), 0, True, None)
        if err is not None:
            raise err
        if pend.size <= room:
            buf[room - pend.size:room] = pend
            bam = buf[room - pend.size:]
        else:
            bam = np.concatenate([pend, buf[room:]])
        secs["read_inflate"] += time.perf_counter() - t1
        start = 0
        if header is None:
            try:
                _text, refs, start = _lib.bam_header(bam)
            except _lib.SmiError:
                if eof:
                    raise
                pend = bam                           # the header is not complete yet: read on
                continue
            header = bam[:start].copy()
            if rank == 0:
                emit(header, header)
            if refflat is not None:
                tagger = _lib.GeneTagger(refflat, [nm for nm, _ in refs])
        t1 = time.perf_counter()
        recs, end = _lib.bam_index_records(bam, start, cap=max(1, (bam.size - start) // 36))
        if eof and end != bam.size:
            raise _lib.SmiError("truncated BAM record")
        m = int(recs.size)
        nth = np.zeros(max(m, 1), dtype=np.uint8)
        nth[:nth_pend.size] = nth_pend
        names_seen.seen(bam, recs, nth_pend.size, nth)
        gene = tagger.tag_bam_raw(bam, recs) if tagger is not None and m else None
        secs["index"] += time.perf_counter() - t1
        tags = np.zeros(max(m, 1), dtype=_lib.UMI_TAG_DTYPE)
        region = np.full(max(m, 1), -1, dtype=np.int64)
        ref = recs["ref_id"].astype(np.int64)
        cuts, i0 = segment_cuts(ref, cur_n, g0, i0, prev_ref, step)
        cur = np.arange(0, cur_n, dtype=np.int32)          # local indices of the chunk being collected (they lead pend)
        k = cur_n                                          # next local record to append

        def flush(cur, keep):
            nonlocal region_base, n_records, n_clustered, n_batches
            t2 = time.perf_counter()
            inp = _lib.bam_chunk_inputs(bam, recs, cur)
            out, n_done = ctx.assignumis_chunk_raw(inp, keep_data_end=keep, max_dist=max_dist, bc_edit_limit=bc_edit_limit, n_threads=n_threads,
                                                   five_prime=five_prime, cluster_cfg=cluster_cfg, umi_length=umi_length, grouping_distance=grouping_distance,
                                                   random_umi_seed=random_umi_seed)
            t3 = time.perf_counter()
            done = cur[:n_done]
            if no_clustering:
                out["flags"] &= _lib.UMI_HAS_BC
            tags[done] = out[:n_done]
            reg = out["region"][:n_done].astype(np.int64)
            region[done] = np.where(reg >= 0, reg + region_base, -1)
            region_base += int(reg.max()) + 1 if n_done and reg.max() >= 0 else 0
            n_clustered += int(((out["flags"][:n_done] & _lib.UMI_CLUSTERED) != 0).sum())
            bound = _lib.bam_write_bound(recs, done)
            bc, umi, _o = _lib.bam_write_batch(bam, recs, done, tags, gene=gene, bc_edit_limit=bc_edit_limit, truncate_read_name=truncate_read_name,
                                               five_prime=five_prime, n_threads=n_threads, gene_counts=gc, region=region, nth_record=nth,
                                               out_bc=staged("bc", bound), out_umi=staged("umi", bound), gene_tag=gene_tag)
            t4 = time.perf_counter()
            secs["umi_stage"] += t3 - t2
            secs["write_batch"] += t4 - t3
            emit(bc, umi)
            n_records += n_done
            n_batches += 1
            return cur[n_done:]

        for at, keep in cuts:
            cur = flush(np.concatenate([cur, np.arange(k, at, dtype=np.int32)]), keep=keep)
            k = at
        cur = np.concatenate([cur, np.arange(k, m, dtype=np.int32)])
        if eof:
            while cur.size:
                cur = flush(cur, False)
            pend = np.zeros(0, dtype=np.uint8)
            break
        # what stays for the next segment: the chunk being collected (contiguous records) and the bytes behind the last complete record
        first = int(cur[0]) if cur.size else m
        off = int(recs[first]["rec_off"]) if first < m else int(end)
        pend = bam[off:].copy()
        room_hint[0] = max(256 << 20, int(pend.size * 1.5))
        nth_pend = nth[first:m].copy()
        cur_n = int(cur.size)
        if m:
            prev_ref = int(ref[m - 1])
        g0 += first
    for fh in (f_bc, f_umi):
        if world == 1:
            fh.write(_BGZF_EOF)
        fh.close()
    f_in.close()
    for pb in stage.values():
        pb.close()
    order_dependent = 0
    if world > 1:
        # the one exchange of a sharded run: every rank's gene tables (and counts) to rank 0, which strings the output shards together
        mine = (gc.dump(), n_records, n_clustered, n_batches)
        if dist is not None:
            parts = [None] * world if rank == 0 else None
            dist.gather_object(mine, parts, dst=0, group=group)
        else:
            parts = None                          # shard=(rank, world) without a process group: the caller merges (merge_shards below)
            with open(out_prefix + f".genecounts.shard{rank}", "wb") as f:
                f.write(mine[0])
        if dist is not None and rank == 0:
            order_dependent = _string_shards(out_prefix, world, gc, [p_[0] for p_ in parts[1:]])
            n_records, n_clustered, n_batches = (sum(p_[k] for p_ in parts) for k in (1, 2, 3))
        if dist is not None:
            dist.barrier(group=group)
        if rank != 0 or dist is None:
            info = gc.info()
            gc.close()
            names_seen.close()
            if tagger is not None:
                tagger.close()
            return dict(records=n_records, clustered=n_clustered, batches=n_batches, seconds=secs, wall_s=time.perf_counter() - t_all, rank=rank, world=world, **info)
    if not simulate:
        with open(out_prefix + ".genecounts.tsv", "w") as f:
            f.write(gc.genecounts_tsv(bc_length))
        with open(out_prefix + ".UMIdepths.tsv", "w") as f:
            f.write(gc.umi_depths_tsv())
    info = gc.info()
    gc.close()
    names_seen.close()
    if tagger is not None:
        tagger.close()
    return dict(records=n_records, clustered=n_clustered, batches=n_batches, seconds=secs, wall_s=time.perf_counter() - t_all, rank=rank, world=world,
                gene_keys_order_dependent=order_dependent, **info)


def _string_shards(out_prefix, world, gc, later_dumps):
    """rank 0's end of a sharded run: the gene tables of ranks 1 .. world-1 (dumps) merged into gc in rank order, the BAM shards of both
    outputs strung together in rank order (BGZF streams concatenate) + the end-of-file block, the shard files removed -> number of
    (gene, cell, UMI) keys whose counters depend on the order of the records and were NOT added (smi_gene_counts_merge_shard)"""
    order_dependent = 0
    for dump in later_dumps:
        later = _lib.GeneCounts.load(dump)
        order_dependent += gc.merge_shard(later)
        later.close()
    for name in (".bam", "_umifound_.bam"):
        with open(out_prefix + name, "wb") as dst:
            for q in range(world):
                with open(out_prefix + name + f".shard{q}", "rb") as src:
                    while True:
                        blk = src.read(64 << 20)
                        if not blk:
                            break
                        dst.write(blk)
                os.remove(out_prefix + name + f".shard{q}")
            dst.write(_BGZF_EOF)
    return order_dependent


def merge_shards(out_prefix, world, bc_length=16):
    """what follows `world` runs of assignumis_stream(..., shard=(r, world)) that had no process group between them (one job per GPU of a
    scheduler, say): <out>.bam / <out>_umifound_.bam from the shard files, <out>.genecounts.tsv / <out>.UMIdepths.tsv from the
    <out>.genecounts.shard<r> dumps; the shard files are removed.  The files equal those of the torch.distributed run and of one process
    (assignumis_stream's docstring says what may differ) -> dict(gene_keys_order_dependent=...)"""
    dumps = []
    for q in range(world):
        for name in (".bam", "_umifound_.bam", ".genecounts"):
            if not os.path.isfile(out_prefix + name + f".shard{q}"):
                raise _lib.SmiError(f"merge_shards: {out_prefix}{name}.shard{q} is missing (rank {q} of {world} has not finished)")
        with open(out_prefix + f".genecounts.shard{q}", "rb") as f:
            dumps.append(f.read())
    gc = _lib.GeneCounts.load(dumps[0])
    order_dependent = _string_shards(out_prefix, world, gc, dumps[1:])
    with open(out_prefix + ".genecounts.tsv", "w") as f:
        f.write(gc.genecounts_tsv(bc_length))
    with open(out_prefix + ".UMIdepths.tsv", "w") as f:
        f.write(gc.umi_depths_tsv())
    info = gc.info()
    gc.close()
    for q in range(world):
        os.remove(out_prefix + f".genecounts.shard{q}")
    return dict(gene_keys_order_dependent=order_dependent, **info)


def assignumis_files(ctx, in_bam, out_prefix, bc_length=16, native=True, **kw):
    """`assignumis -i in.bam -o out`: writes <out>.bam, <out>_umifound_.bam, <out>.genecounts.tsv and <out>.UMIdepths.tsv
    (UmiFinderWorker.java:L142-143, L188-189) -> dict of what was written.  native: through write_tagged_bams_native (host threads + device,
    no per-record Python) or through the Python mirror write_tagged_bams; the files are the same.  kw: as those (refflat = text of
    --annotationFile)."""
    with open(in_bam, "rb") as f:
        data = f.read()
    gc = _lib.GeneCounts()
    if native:
        bc_bam, umi_bam, info = write_tagged_bams_native(ctx, data, gene_counts=gc, **kw)
        n_records, n_clustered = info["records"], info["clustered"]
    else:
        bc_bam, umi_bam, names, tags = write_tagged_bams(ctx, data, gene_counts=gc, **kw)
        n_records, n_clustered = len(names), sum(1 for t in tags if t is not None and not t.get("skipped"))
    texts = {".bam": bc_bam, "_umifound_.bam": umi_bam, ".genecounts.tsv": gc.genecounts_tsv(bc_length).encode(),
             ".UMIdepths.tsv": gc.umi_depths_tsv().encode()}
    for suffix, body in texts.items():
        with open(out_prefix + suffix, "wb") as f:
            f.write(bytes(body) if isinstance(body, np.ndarray) and body.size < (1 << 20) else body)
    info = gc.info()
    gc.close()
    return dict(records=n_records, clustered=n_clustered, **info, files={out_prefix + k: len(v) for k, v in texts.items()})
