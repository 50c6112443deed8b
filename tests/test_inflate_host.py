"""The host's gzip decoder (smi_inflate_host.hip: smi_gz_inflate / smi_gz_inflate_into, and under smi_bgzf_inflate) against zlib's output:
every block type, code shapes that need subtables, multi-member files, and the ways a file can be broken."""
import ctypes
import gzip
import zlib

import numpy as np
import pytest


@pytest.fixture
def libmod(pkg):
    from sicelore_amd import lib as libmod

    libmod.load_library()
    return libmod


def _gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strategy)
    return c.compress(data) + c.flush()


def _fastq(rng, n):
    recs = []
    for i in range(n):
        ln = int(rng.integers(200, 1500))
        s = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, ln)])
        q = bytes((rng.integers(35, 64, ln) + 33).astype(np.uint8))
        recs.append(b"@read%d_some_name\n" % i + s + b"\n+\n" + q + b"\n")
    return b"".join(recs)


def test_round_trips_of_zlib_streams(libmod):
    rng = np.random.default_rng(5)
    skew = np.minimum(rng.geometric(0.02, 400_000), 255).astype(np.uint8).tobytes()      # code lengths up to 15: subtables
    samples = {"empty": b"", "one": b"A", "zeros": bytes(1_000_000), "fastq": _fastq(rng, 600), "random": rng.bytes(300_000), "skewed": skew,
               "text": open(__file__, "rb").read() * 30, "runs": b"".join(bytes([int(c)]) * int(k) for c, k in zip(rng.integers(0, 256, 3000), rng.integers(1, 600, 3000)))}
    n = 0
    for name, data in samples.items():
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                for mem in (1, 9):            # memLevel 1: blocks of 128 symbols, i.e. thousands of dynamic headers
                    z = _gz(data, level, strategy, mem)
                    got = libmod.gz_inflate(np.frombuffer(z, dtype=np.uint8))
                    assert got.tobytes() == data, (name, level, strategy, mem)
                    n += 1
    assert n == 8 * 4 * 5 * 2


def test_multi_member_growth_and_headers(libmod):
    rng = np.random.default_rng(6)
    parts = [b"hello ", b"world" * 1000, b"", _fastq(rng, 50), bytes(200_000)]
    z = b"".join(gzip.compress(p, compresslevel=lv) for p, lv in zip(parts, (1, 9, 6, 1, 6)))
    assert libmod.gz_inflate(np.frombuffer(z, dtype=np.uint8)).tobytes() == b"".join(parts)    # the last member's size is no hint here: the buffer grows
    # the step entry: positions stay in front of the member that does not fit
    lib = libmod.load_library()
    a = np.frombuffer(z, dtype=np.uint8)
    out = np.zeros(5006 + 10, dtype=np.uint8)
    ip, op = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert lib.smi_gz_inflate_into(a.ctypes.data, a.size, ctypes.byref(ip), out.ctypes.data, out.size, ctypes.byref(op)) == 1
    assert op.value == 5006 and out[:5006].tobytes() == b"hello " + b"world" * 1000 and z[ip.value:ip.value + 2] == b"\x1f\x8b"
    # optional header fields (FEXTRA, FNAME, FCOMMENT, FHCRC)
    body = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = body.compress(b"payload " * 99) + body.flush()
    trailer = zlib.crc32(b"payload " * 99).to_bytes(4, "little") + (8 * 99).to_bytes(4, "little")
    head = bytes([31, 139, 8, 2 | 4 | 8 | 16, 0, 0, 0, 0, 0, 255]) + (5).to_bytes(2, "little") + b"extra" + b"name.fq\0" + b"a comment\0" + b"\x12\x34"
    assert libmod.gz_inflate(np.frombuffer(head + raw + trailer, dtype=np.uint8)).tobytes() == b"payload " * 99
    # smi_gz_inflate with an exact buffer, and one byte short
    n = ctypes.c_size_t(0)
    exact = np.zeros(8 * 99, dtype=np.uint8)
    m = np.frombuffer(head + raw + trailer, dtype=np.uint8)
    assert lib.smi_gz_inflate(m.ctypes.data, m.size, exact.ctypes.data, exact.size, ctypes.byref(n)) == 0 and n.value == 8 * 99
    assert lib.smi_gz_inflate(m.ctypes.data, m.size, exact.ctypes.data, exact.size - 1, ctypes.byref(n)) != 0
    assert "too small" in lib.smi_last_error().decode()


def test_broken_files_are_refused(libmod):
    rng = np.random.default_rng(7)
    data = _fastq(rng, 80)
    z = bytearray(_gz(data, 6))

    def fails(buf, word):
        with pytest.raises(libmod.SmiError) as e:
            libmod.gz_inflate(np.frombuffer(bytes(buf), dtype=np.uint8))
        assert word in str(e.value), str(e.value)

    bad = bytearray(z)
    bad[-8] ^= 1
    fails(bad, "CRC-32")                                   # CRC
    bad = bytearray(z)
    bad[-4] ^= 1
    fails(bad, "")                                         # ISIZE (the buffer sized from it is too small, or the length check fails)
    for cut in (len(z) - 1, len(z) - 9, len(z) // 2, 19, 5):
        fails(z[:cut], "")                                 # truncated anywhere
    # bytes behind the last member end the stream, as they do for java.util.zip.GZIPInputStream (a failed read of the next header is end of
    # stream) and for gzip / zlib (padding): the text of the members in front comes back
    for pad in (b"garbage behind the member", bytes(512), b"\x1f\x8b\x09 not a header"):
        assert libmod.gz_inflate(np.frombuffer(bytes(z) + pad, dtype=np.uint8)).tobytes() == data
    fails(b"\x1f\x8b\x07" + bytes(z[3:]), "not a gzip stream")
    # every single-bit flip inside the deflate data either still decodes to something whose CRC fails, or is refused; never a crash
    for k in range(200):
        bad = bytearray(z)
        pos = 10 + int(rng.integers(0, len(z) - 18))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            got = libmod.gz_inflate(np.frombuffer(bytes(bad), dtype=np.uint8))
            assert got.tobytes() == data                   # (a flip in a stored block's padding bits changes nothing)
        except libmod.SmiError:
            pass
    # reserved block type, bad stored length, distance before the start, over-subscribed code
    def member(raw):
        return bytes([31, 139, 8, 0, 0, 0, 0, 0, 0, 255]) + raw + bytes(8)

    fails(member(b"\x07\x00"), "invalid DEFLATE")
    fails(member(b"\x01\x05\x00\x00\x00hello"), "invalid DEFLATE")
    fixed_far = zlib.compressobj(9, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)
    raw = fixed_far.compress(b"abcabcabcabc") + fixed_far.flush()
    ok = bytes([31, 139, 8, 0, 0, 0, 0, 0, 0, 255]) + raw + zlib.crc32(b"abcabcabcabc").to_bytes(4, "little") + (12).to_bytes(4, "little")
    assert libmod.gz_inflate(np.frombuffer(ok, dtype=np.uint8)).tobytes() == b"abcabcabcabc"
    # a fixed-Huffman block that starts with a match (length 3, distance 1): it reaches in front of the output
    bits = [1, 1, 0] + [0, 0, 0, 0, 0, 0, 1] + [0] * 5 + [0] * 7       # BFINAL, BTYPE = 01; symbol 257; distance code 0; end of block
    raw = bytearray((len(bits) + 7) // 8)
    for i, bit in enumerate(bits):
        raw[i >> 3] |= bit << (i & 7)
    fails(member(bytes(raw)), "invalid DEFLATE")
    with pytest.raises(zlib.error):
        zlib.decompress(bytes(raw), -15)


def test_bgzf_blocks_through_the_same_decoder(libmod):
    """BGZF (BAM) blocks: the payload of each block through the library's decoder, CRC-32 by carry-less multiplication"""
    rng = np.random.default_rng(8)
    data = _fastq(rng, 300)
    z = libmod.bgzf_deflate(data, level=6, n_threads=2)
    back, used = libmod.bgzf_inflate(z, n_threads=3)
    assert back.tobytes() == data and used == z.size
    bad = np.array(z, copy=True)
    bad[40] ^= 4
    with pytest.raises(libmod.SmiError):
        libmod.bgzf_inflate(bad, n_threads=2)
