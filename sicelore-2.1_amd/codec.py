"""Host-side 2-bit codec helpers (the reference's code: A=0 G=1 C=2 T=3, first base most significant;
TB!nuc/encoding/TwoBit/NucleicAcidTwoBitPerBase.java:L78-87,L183-187,L337-342).  Pure data plumbing for
building inputs and printing outputs -- no matching logic lives here."""
import numpy as np

_ENC = np.full(256, 255, dtype=np.uint8)
for _c, _v in (("A", 0), ("G", 1), ("C", 2), ("T", 3)):
    _ENC[ord(_c)] = _v
    _ENC[ord(_c.lower())] = _v
_DEC = np.frombuffer(b"AGCT", dtype=np.uint8)


def encode_kmers(seqs, k=16):
    """list of k-mers (str) -> uint64 keys; raises on a non-ACGT character."""
    out = np.zeros(len(seqs), dtype=np.uint64)
    for i, s in enumerate(seqs):
        if len(s) != k:
            raise ValueError(f"barcode {s!r} is not {k} nt")
        codes = _ENC[np.frombuffer(s.encode(), dtype=np.uint8)]
        if (codes == 255).any():
            raise ValueError(f"barcode {s!r} has a non-ACGT base")
        v = 0
        for c in codes:
            v = (v << 2) | int(c)
        out[i] = v
    return out


def decode_kmer(key, k=16):
    key = int(key)
    return bytes(_DEC[[(key >> (2 * (k - 1 - i))) & 3 for i in range(k)]]).decode()


def codes_to_ascii(codes):
    """uint8 array of 2-bit codes (4 = N) -> ASCII bytes"""
    lut = np.frombuffer(b"AGCTN", dtype=np.uint8)
    return lut[np.asarray(codes, dtype=np.uint8)]


def ascii_to_codes(b):
    """ASCII bytes -> 2-bit codes, 4 for anything that is not ACGT"""
    c = _ENC[np.frombuffer(b, dtype=np.uint8)] if isinstance(b, (bytes, bytearray)) else _ENC[np.asarray(b)]
    return np.where(c == 255, 4, c).astype(np.uint8)


def read_barcode_file(path):
    """One barcode per line, everything from the first '-' dropped, .gz accepted
    (FJ!nanoporereadscanner/NanoporeReadScannerMain.java:L483-496)."""
    import gzip

    op = gzip.open if str(path).endswith(".gz") else open
    seqs = []
    with op(path, "rt") as fh:
        for line in fh:
            s = line.strip().split("-")[0]
            if s:
                seqs.append(s)
    return encode_kmers(seqs, len(seqs[0]) if seqs else 16)
