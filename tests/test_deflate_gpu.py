"""K-DEFLATE (smi_gzip_device, the `--compress` writer of scanfastq on the device): every member inflates back to the input with an independent
inflater (zlib / gzip of the Python standard library, which also checks the CRC-32 and ISIZE of the trailer) -- the encode -> decode round
trip is the parity property of a compressor; sizes around the 64 KiB block boundary, empty input, one symbol, all 256 byte values,
incompressible input, a frequency profile whose optimal code is deeper than 15 bits, FASTQ text at the size of a pass-2 chunk."""
import gzip
import importlib
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _roundtrip(ctx, data, raw=False):
    dev = torch.device("cuda", ctx.device)
    d_in = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev) if len(data) else torch.zeros(0, dtype=torch.uint8, device=dev)
    out = ctx.gzip_device(d_in, len(data), raw_deflate=raw).cpu().numpy().tobytes()
    if raw:
        back = zlib.decompress(out, wbits=-15)
    else:
        assert out[:4] == b"\x1f\x8b\x08\x00"
        back = gzip.decompress(out)
        assert int.from_bytes(out[-8:-4], "little") == zlib.crc32(data) and int.from_bytes(out[-4:], "little") == len(data) & 0xFFFFFFFF
    assert back == data
    return out


def _fastq(n, seed):
    g = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        m = int(g.integers(300, 1800))
        seq = g.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=m).tobytes()
        qual = (33 + np.clip(g.normal(18, 7, size=m), 1, 50).astype(np.uint8)).tobytes()
        recs.append(b"@%08x-aaaa-bbbb read=%d ch=%d_FWD_PS=%d_AE=%d_bc=ACGTACGTACGTACGT_ed=1\n" % (i * 2654435761 % 2 ** 32, i, i % 512, m - 60, m - 20) + seq + b"\n+\n" + qual + b"\n")
    return b"".join(recs)


@pytest.mark.parametrize("n", [0, 1, 2, 255, 256, 257, 4095, 65535, 65536, 65537, 131072, 200001])
def test_sizes_around_the_block_boundaries(gpu_ctx, n):
    g = np.random.default_rng(n)
    data = g.choice(np.frombuffer(b"ACGTN\n!5?I", dtype=np.uint8), size=n, p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .02]).tobytes()
    out = _roundtrip(gpu_ctx, data)
    _roundtrip(gpu_ctx, data, raw=True)
    if n >= 4095:
        assert len(out) < 0.45 * n        # four letters dominate: close to 2.5 bits per byte


def test_one_symbol_all_symbols_and_incompressible_input(gpu_ctx):
    _roundtrip(gpu_ctx, b"A" * 70_000)
    _roundtrip(gpu_ctx, bytes(range(256)) * 300)
    rnd = np.random.default_rng(1).integers(0, 256, size=300_000, dtype=np.uint8).tobytes()
    out = _roundtrip(gpu_ctx, rnd)
    assert len(out) < len(rnd) * 1.01 + 1024       # a Huffman code over uniform bytes costs 8 bits each: no blow-up


def test_code_deeper_than_fifteen_bits_is_flattened(gpu_ctx):
    # Fibonacci frequencies: the minimum-redundancy code of 24 such symbols is 23 bits deep; deflate allows 15
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    assert sum(fib) < 65536 + 60_000
    blk = b"".join(bytes([65 + k]) * f for k, f in enumerate(fib))[:65536]
    g = np.random.default_rng(3)
    data = bytes(g.permutation(np.frombuffer(blk, dtype=np.uint8)))
    _roundtrip(gpu_ctx, data)
    _roundtrip(gpu_ctx, data + b"tail of another block" * 100)


def test_fastq_chunk_and_concatenated_members(gpu_ctx):
    a, b = _fastq(6000, 11), _fastq(50, 12)
    za, zb = _roundtrip(gpu_ctx, a), _roundtrip(gpu_ctx, b)
    assert gzip.decompress(za + zb) == a + b                   # members of one .gz file (one per chunk)
    libmod = importlib.import_module("sicelore_amd.lib")       # and through the library's own host decoder (two-literal table entries)
    assert libmod.gz_inflate(np.frombuffer(za + zb, dtype=np.uint8)).tobytes() == a + b
    ref = len(zlib.compress(a, 6))
    assert len(za) < 1.25 * ref                                # literals only: within a quarter of zlib level 6 on FASTQ text
    # an unaligned input pointer (a view into a larger device buffer)
    dev = torch.device("cuda", gpu_ctx.device)
    big = torch.from_numpy(np.frombuffer(b"xyz" + a, dtype=np.uint8).copy()).to(dev)
    out = gpu_ctx.gzip_device(big[3:], len(a)).cpu().numpy().tobytes()
    assert gzip.decompress(out) == a


def test_text_worker_with_compress_returns_the_members_of_its_text(pkg, synth, gpu_ctx):
    """smi_scanfastq_pass2_chunk with cfg.compress: `passed` / `failed` inflate to exactly the text the same call returns without it; the
    statistics of the chunk are those of the packed worker; the packed worker refuses the option"""
    wl = synth.make_whitelist(20_000, seed=77)
    used = synth.pick_used(wl, 150, seed=78)
    reads = synth.gen_reads(700, used, seed=79, n_rate=0.002)
    chim = synth.make_chimeras(reads, 900, seed=80)
    text = "".join(f"@read{i} runid=x ch={i % 9}\n{c[0]}\n+\n{c[1]}\n" for i, c in enumerate(chim)).encode()
    keys = np.sort(used.numpy().astype(np.uint64))
    gpu_ctx.set_barcode_set(keys, mode=0)
    kw = dict(rank_keys=keys, rank_values=(np.arange(keys.size) % 40 + 1).astype(np.int32), first_read_id=36 ** 3, want_results=True)
    p, f, info = gpu_ctx.scanfastq_pass2_chunk(text, **kw)
    zp, zf, zinfo = gpu_ctx.scanfastq_pass2_chunk(text, compress=True, **kw)
    assert gzip.decompress(bytes(zp)) == bytes(p) and gzip.decompress(bytes(zf)) == bytes(f)
    assert (zinfo["passed_text_bytes"], zinfo["failed_text_bytes"]) == (len(p), len(f)) == (info["passed_text_bytes"], info["failed_text_bytes"])
    assert len(zp) < 0.6 * len(p) and zinfo["n_passed"] == info["n_passed"]
    pk = gpu_ctx.scanfastq_pass2_chunk(text, packed=True, n_threads=2, **kw)[2]
    assert (info["stats"] == pk["stats"]).all() and (zinfo["stats"] == pk["stats"]).all() and int(pk["stats"][3]) == info["n_records_out"] > 900
    with pytest.raises(pkg.SmiError):
        gpu_ctx.scanfastq_pass2_chunk(text, packed=True, compress=True)
    # an empty stream is still a member (a chunk whose records all passed)
    allp = "".join(f"@r{i}\n{c[0]}\n+\n{c[1]}\n" for i, c in enumerate(chim[:1])).encode()
    zp1, zf1, i1 = gpu_ctx.scanfastq_pass2_chunk(allp, compress=True)
    assert gzip.decompress(bytes(zp1) + bytes(zf1)) is not None and (i1["passed_text_bytes"] == 0 or i1["failed_text_bytes"] == 0)


def test_bgzf_blocks_made_on_the_device(pkg, gpu_ctx):
    """smi_bgzf_deflate_device (the BGZF writer under assignumis' output BAMs with K-DEFLATE inside): the stream inflates back through Python's
    gzip (a BGZF file is a multi-member gzip file), through the library's own BGZF reader, block by block through the BSIZE fields; blocks
    stay below 64 KiB whatever the content; the empty stream is the 28-byte end-of-file block"""
    from sicelore_amd import lib as libmod

    g = np.random.default_rng(4)
    bam_like = b"".join(int(g.integers(200, 400)).to_bytes(4, "little") + bytes(g.integers(0, 40, size=int(g.integers(200, 400)), dtype=np.uint8)) +
                        b"read%06d_FWD_PS=12_AE=99_bc=ACGTACGTACGTACGT\0" % i for i in range(3000))
    rnd = g.integers(0, 256, size=200_000, dtype=np.uint8).tobytes()
    eof = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    for data in (bam_like, rnd, b"x", b"A" * 61_440, b"B" * 61_441, b""):
        z = gpu_ctx.bgzf_deflate_device(data).tobytes()
        assert z.endswith(eof)
        assert gzip.decompress(z) == data
        back, used = libmod.bgzf_inflate(np.frombuffer(z, dtype=np.uint8), n_threads=2)
        assert back.tobytes() == data and used == len(z)
        at, total = 0, 0
        while at < len(z):
            assert z[at:at + 4] == b"\x1f\x8b\x08\x04" and z[at + 12:at + 16] == b"BC\x02\x00"
            bsize = int.from_bytes(z[at + 16:at + 18], "little") + 1
            assert bsize <= 65536
            isize = int.from_bytes(z[at + bsize - 4:at + bsize], "little")
            assert isize <= 61_440 and zlib.crc32(data[total:total + isize]) == int.from_bytes(z[at + bsize - 8:at + bsize - 4], "little")
            total += isize
            at += bsize
        assert at == len(z) and total == len(data)
    assert gpu_ctx.bgzf_deflate_device(b"").tobytes() == eof
    assert len(gpu_ctx.bgzf_deflate_device(bam_like)) < 0.8 * len(bam_like)
