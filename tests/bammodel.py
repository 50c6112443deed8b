"""Independent Python model of the two formats `assignumis` reads (SAM specification 4.1 BGZF, 4.2 BAM): a writer that
produces test files and a parser the product's index is compared with.  Test infrastructure only."""
import struct
import zlib

CIGAR_OPS = "MIDNSHP=X"
SEQ_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def bgzf_block(payload):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = c.compress(payload) + c.flush()
    bsize = 12 + 6 + len(body) + 8
    head = struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
    return head + body + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))


def bgzf_compress(data, block=0xFF00, eof=True):
    out = [bgzf_block(data[i:i + block]) for i in range(0, len(data), block)]
    if eof:
        out.append(bgzf_block(b""))
    return b"".join(out)


def bam_record(name, flag, ref_id, pos0, mapq, cigar, seq, qual=None, aux=b""):
    """cigar: [(op char, length)]; pos0 0-based; qual: bytes of phred values or None (0xFF fill)"""
    nm = name.encode() + b"\0"
    cg = b"".join(struct.pack("<I", (ln << 4) | CIGAR_OPS.index(op)) for op, ln in cigar)
    packed = bytearray((len(seq) + 1) // 2)
    for i, c in enumerate(seq):
        packed[i // 2] |= SEQ_CODE[c] << (4 if i % 2 == 0 else 0)
    q = bytes(qual) if qual is not None else b"\xff" * len(seq)
    body = struct.pack("<iiBBHHHiiii", ref_id, pos0, len(nm), mapq, 4680, len(cigar), flag, len(seq), -1, -1, 0) + nm + cg + bytes(packed) + q + aux
    return struct.pack("<I", len(body)) + body


def bam_bytes(header_text, refs, records):
    t = header_text.encode()
    out = [b"BAM\1", struct.pack("<I", len(t)), t, struct.pack("<I", len(refs))]
    for nm, ln in refs:
        b = nm.encode() + b"\0"
        out += [struct.pack("<I", len(b)), b, struct.pack("<I", ln)]
    return b"".join(out) + b"".join(records)


def bgzf_decompress(data):
    out, off = [], 0
    while off < len(data):
        xlen = struct.unpack_from("<H", data, off + 10)[0]
        bsize = None
        x = 0
        while x < xlen:
            si1, si2, slen = struct.unpack_from("<BBH", data, off + 12 + x)
            if (si1, si2) == (66, 67):
                bsize = struct.unpack_from("<H", data, off + 12 + x + 4)[0] + 1
            x += 4 + slen
        out.append(zlib.decompress(data[off + 12 + xlen:off + bsize - 8], -15))
        off += bsize
    return b"".join(out)


def parse_bam(bam):
    assert bam[:4] == b"BAM\1"
    l_text = struct.unpack_from("<I", bam, 4)[0]
    text = bam[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<I", bam, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<I", bam, p)[0]
        nm = bam[p + 4:p + 4 + ln - 1].decode()
        refs.append((nm, struct.unpack_from("<I", bam, p + 4 + ln)[0]))
        p += 8 + ln
    recs = []
    while p < len(bam):
        bs = struct.unpack_from("<I", bam, p)[0]
        ref_id, pos0, l_nm, mapq, _bin, n_cig, flag, l_seq, nref, npos, tlen = struct.unpack_from("<iiBBHHHiiii", bam, p + 4)
        q = p + 36
        name = bam[q:q + l_nm - 1].decode()
        q += l_nm
        cigar = [(CIGAR_OPS[v & 15], v >> 4) for v in struct.unpack_from("<%dI" % n_cig, bam, q)]
        q += 4 * n_cig
        seq = "".join("=ACMGRSVTWYHKDBN"[(bam[q + i // 2] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        q += (l_seq + 1) // 2
        qual = bam[q:q + l_seq]
        q += l_seq
        recs.append(dict(name=name, flag=flag, ref_id=ref_id, pos0=pos0, mapq=mapq, cigar=cigar, seq=seq, qual=qual,
                         aux=bam[q:p + 4 + bs], off=p, length=bs + 4))
        p += 4 + bs
    return text, refs, recs


# ---- a BAM index (.bai) for a BGZF stream made by bgzf_compress: what `samtools index` leaves beside passed.bam (quickrun-2.1.sh:39) ----
def bgzf_block_offsets(data, block=0xFF00):
    """compressed offset of every block of bgzf_compress(data, block)"""
    offs, at = [], 0
    for i in range(0, len(data), block):
        offs.append(at)
        at += len(bgzf_block(data[i:i + block]))
    return offs


def virtual_offset(pos, block, block_offsets):
    return (block_offsets[pos // block] << 16) | (pos % block) if pos // block < len(block_offsets) else None


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def bai_bytes(n_ref, records, meta=True):
    """records: (ref_id, pos0, end0, virtual offset of the record, virtual offset behind it) in file order -> the bytes of a .bai
    (bins with one chunk per record run, a 16 kb linear index, and -- meta -- samtools' pseudo-bin 37450 with the reference's extent)"""
    per_ref = [dict() for _ in range(n_ref)]
    ext = [None] * n_ref
    lin = [dict() for _ in range(n_ref)]
    for ref, p0, e0, vb, ve in records:
        if ref < 0:
            continue
        b = reg2bin(p0, max(e0, p0 + 1))
        ch = per_ref[ref].setdefault(b, [])
        if ch and ch[-1][1] == vb:
            ch[-1][1] = ve
        else:
            ch.append([vb, ve])
        ext[ref] = (vb, ve) if ext[ref] is None else (min(ext[ref][0], vb), max(ext[ref][1], ve))
        for w in range(p0 >> 14, (max(e0, p0 + 1) - 1 >> 14) + 1):
            lin[ref].setdefault(w, vb)
    out = [b"BAI\x01", struct.pack("<i", n_ref)]
    for ref in range(n_ref):
        bins = dict(per_ref[ref])
        n_bin = len(bins) + (1 if meta and ext[ref] is not None else 0)
        out.append(struct.pack("<i", n_bin))
        for b, chunks in sorted(bins.items()):
            out.append(struct.pack("<Ii", b, len(chunks)))
            for vb, ve in chunks:
                out.append(struct.pack("<QQ", vb, ve))
        if meta and ext[ref] is not None:
            out.append(struct.pack("<Ii", 37450, 2) + struct.pack("<QQ", ext[ref][0], ext[ref][1]) + struct.pack("<QQ", sum(1 for r in records if r[0] == ref), 0))
        n_intv = (max(lin[ref]) + 1) if lin[ref] else 0
        out.append(struct.pack("<i", n_intv))
        last = 0
        for w in range(n_intv):
            last = lin[ref].get(w, last)
            out.append(struct.pack("<Q", last))
    return b"".join(out)
