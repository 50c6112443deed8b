"""K-INFLATE (smi_gz_inflate_device, the *.fastq.gz inputs of scanfastq inflated on the device): the text of every file equals what zlib makes of
it -- all compression levels and strategies (stored, fixed and dynamic blocks; runs; long distances), gzip headers with a file name, several
members per file, many files per call, K-DEFLATE's own members, empty input; corrupted files and a capacity that is too small come back with a
status and no text."""
import gzip
import io
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strategy)
    return c.compress(data) + c.flush()


def _fastq(n, seed):
    g = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        m = int(g.integers(200, 1500))
        seq = g.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=m, p=[.3, .2, .2, .3]).tobytes()
        qual = (33 + np.clip(g.normal(20, 8, size=m), 1, 60).astype(np.uint8)).tobytes()
        recs.append(b"@%08x-aaaa-bbbb-cccc-%012x runid=abcdef read=%d ch=%d\n" % (i * 2654435761 % 2 ** 32, i, i, i % 512) + seq + b"\n+\n" + qual + b"\n")
    return b"".join(recs)


def _check(ctx, files, texts, caps=None):
    out, offs, lens, status, n_mem = ctx.gz_inflate_device(files, caps)
    host = out.cpu().numpy()
    for i, t in enumerate(texts):
        assert status[i] == 0, (i, int(status[i]))
        assert lens[i] == len(t), (i, int(lens[i]), len(t))
        got = host[offs[i]:offs[i] + lens[i]].tobytes()
        if got != t:
            k = next(j for j in range(len(t)) if got[j] != t[j])
            raise AssertionError((i, "first difference at byte", k, got[max(0, k - 20):k + 20], t[max(0, k - 20):k + 20]))
    return n_mem


def test_levels_and_strategies(gpu_ctx):
    g = np.random.default_rng(1)
    text = _fastq(400, 2)
    runs = b"".join(bytes([int(g.integers(65, 70))]) * int(g.integers(1, 700)) for _ in range(3000))
    rnd = g.integers(0, 256, size=300_000, dtype=np.uint8).tobytes()
    words = b" ".join(g.choice([b"polyA", b"adapter", b"barcode", b"UMI", b"TSO", b"read", b"nanopore", b"x"], size=120_000).tolist())
    files, texts = [], []
    for data in (text, runs, rnd, words, b"A", b"", b"ACGT" * 70_000):
        for level in (0, 1, 4, 6, 9):
            files.append(_gz(data, level))
            texts.append(data)
        for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            files.append(_gz(data, 6, strategy))
            texts.append(data)
        files.append(_gz(data, 9, mem=9))
        texts.append(data)
    _check(gpu_ctx, files, texts)


def test_headers_members_and_own_output(gpu_ctx):
    a, b, c = _fastq(300, 5), _fastq(10, 6), _fastq(1, 7)
    buf = io.BytesIO()
    with gzip.GzipFile(filename="reads_0001.fastq", mode="wb", fileobj=buf, mtime=1234567) as f:   # FNAME in the header
        f.write(a)
    named = buf.getvalue()
    assert named[3] & 8
    multi = _gz(a, 6) + _gz(b, 1) + _gz(b"", 6) + _gz(c, 9)
    dev = torch.device("cuda", gpu_ctx.device)
    own = b"".join(gpu_ctx.gzip_device(torch.from_numpy(np.frombuffer(t, dtype=np.uint8).copy()).to(dev), len(t)).cpu().numpy().tobytes() for t in (a, b))
    n_mem = _check(gpu_ctx, [named, multi, own], [a, a + b + c, a + b], caps=[len(a), len(a + b + c), len(a + b)])
    assert list(n_mem) == [1, 4, 2]


def test_many_files_in_one_call(gpu_ctx):
    texts = [_fastq(int(n), 100 + i) for i, n in enumerate(np.random.default_rng(9).integers(1, 120, size=150))]
    files = [_gz(t, int(lv)) for t, lv in zip(texts, np.random.default_rng(10).integers(1, 10, size=len(texts)))]
    _check(gpu_ctx, files, texts)


def test_bad_input_is_reported_not_inflated(gpu_ctx):
    text = _fastq(200, 11)
    good = _gz(text, 6)
    flipped = bytearray(good)
    flipped[len(good) // 2] ^= 0x10                      # a bit inside the deflate stream
    wrong_crc = bytearray(good)
    wrong_crc[-6] ^= 1
    truncated = good[:len(good) // 2]
    not_gzip = b"@read\nACGT\n+\nIIII\n" * 10
    out, offs, lens, status, _ = gpu_ctx.gz_inflate_device([good, bytes(flipped), bytes(wrong_crc), truncated, not_gzip, good],
                                                         [len(text)] * 5 + [len(text) - 1])
    assert status[0] == 0 and out[offs[0]:offs[0] + lens[0]].cpu().numpy().tobytes() == text
    assert all(int(s) != 0 for s in status[1:]), list(status)
