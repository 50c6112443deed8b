"""Seeded synthetic inputs: whitelist, used-barcode list with log-normal cell sizes and
Nanopore-error-profile barcode regions.  The reference has no generator and its test data are absent
(/root/reference/.MISSING_LARGE_BLOBS), so every measured number names the seed and version below.

Written with torch ops only so that the same code builds 10^7 windows on the GPU in milliseconds and small
cases on the CPU for the oracle comparison.  Generator, not product: nothing here assigns barcodes.
"""
import torch

GENERATOR_VERSION = "synth-1"

# Jar/config.xml:111-118 (3' adapter, complete form) and :155 (TSO)
ADAPTER_3P_COMPLETE = "CTACACGACGCTCTTCCGATCT"
ADAPTER_3P_SHORT = "CTTCCGATCT"
TSO = "AACGCAGAGTACATGG"

_CODE = {"A": 0, "G": 1, "C": 2, "T": 3}


def _codes(s, device):
    return torch.tensor([_CODE[c] for c in s], dtype=torch.int64, device=device)


def _rc(codes):
    """reverse complement along the last dim (complement = 3 - code)"""
    return 3 - torch.flip(codes, dims=[-1])


def _gen(seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def make_whitelist(n, seed=1, device="cpu"):
    """n distinct pseudo-random 16-mers as sorted int64 keys (stand-in for 3M-february-2018.txt.gz)."""
    g = _gen(seed, device)
    keys = torch.empty(0, dtype=torch.int64, device=device)
    while keys.numel() < n:
        m = int((n - keys.numel()) * 1.05) + 1024
        new = torch.randint(0, 1 << 32, (m,), generator=g, device=device, dtype=torch.int64)
        keys = torch.unique(torch.cat([keys, new]))
    if keys.numel() > n:
        # drop a seeded random subset so the result is not biased towards small keys
        perm = torch.randperm(keys.numel(), generator=g, device=device)[:n]
        keys = torch.sort(keys[perm]).values
    return keys


def pick_used(whitelist, n_cells, seed=2):
    g = _gen(seed, whitelist.device)
    idx = torch.randperm(whitelist.numel(), generator=g, device=whitelist.device)[:n_cells]
    return whitelist[idx]


def keys_to_codes(keys, k=16):
    shifts = torch.arange(k - 1, -1, -1, device=keys.device, dtype=torch.int64) * 2
    return (keys.unsqueeze(-1) >> shifts) & 3


def gen_bc_region(n, used_keys, seed=3, device=None, err=0.063, frac=(0.4, 0.3, 0.3), five_prime=False, jitter=0.1,
                  n_rate=0.0, width=128):
    """Simulate the barcode end of n reads through a sub/ins/del channel.

    Returns a dict:
      codes  uint8 [n, width]   the read region in stranded orientation (2-bit codes, 4 = N)
      ae     int32 [n]          adapter end AE (1-based, as Parser.assignBarcode reads it), with +-1 jitter
      truth  int64 [n]          the barcode that was put in
      umi    int64 [n]          the 12-nt UMI that was put in (2-bit packed)
    """
    device = device or used_keys.device
    g = _gen(seed, device)
    n_cells = used_keys.numel()
    # log-normal cell sizes (gives the knee SURVEY 8d asks for)
    w = torch.exp(torch.randn(n_cells, generator=g, device=device))
    cell = torch.multinomial(w, n, replacement=True, generator=g)
    truth = used_keys.to(device)[cell]
    bc = keys_to_codes(truth, 16)
    umi = torch.randint(0, 4, (n, 12), generator=g, device=device, dtype=torch.int64)
    umi_key = (umi << (torch.arange(11, -1, -1, device=device, dtype=torch.int64) * 2)).sum(-1)
    if not five_prime:
        cdna = torch.randint(0, 4, (n, 20), generator=g, device=device, dtype=torch.int64)
        polya = torch.zeros((n, 12), dtype=torch.int64, device=device)
        ad = _rc(_codes(ADAPTER_3P_COMPLETE, device)).expand(n, -1)
        src = torch.cat([cdna, polya, _rc(umi), _rc(bc), ad], dim=1)
        anchor = 20 + 12 + 12 + 16  # source index of the first adapter base
    else:
        pre = torch.randint(0, 4, (n, 6), generator=g, device=device, dtype=torch.int64)
        ad = _codes(ADAPTER_3P_COMPLETE, device).expand(n, -1)
        cdna = torch.randint(0, 4, (n, 30), generator=g, device=device, dtype=torch.int64)
        src = torch.cat([pre, ad, bc, umi, cdna], dim=1)
        anchor = 6 + 22  # source index of the first barcode base
    S = src.shape[1]
    p_sub, p_ins, p_del = (err * f for f in frac)
    u = torch.rand((n, S), generator=g, device=device)
    is_del = u < p_del
    is_sub = (u >= p_del) & (u < p_del + p_sub)
    sub_base = (src + torch.randint(1, 4, (n, S), generator=g, device=device, dtype=torch.int64)) & 3
    base = torch.where(is_sub, sub_base, src)
    has_ins = torch.rand((n, S), generator=g, device=device) < p_ins
    ins_base = torch.randint(0, 4, (n, S), generator=g, device=device, dtype=torch.int64)
    kept = (~is_del).to(torch.int64)
    cnt = kept + has_ins.to(torch.int64)
    start = torch.cumsum(cnt, dim=1) - cnt
    out = torch.randint(0, 4, (n, width + 2), generator=g, device=device, dtype=torch.int64)
    dump = width + 1  # writes of absent elements land in a scratch column
    pos_base = torch.where(is_del, torch.full_like(start, dump), start).clamp_(max=dump)
    out.scatter_(1, pos_base, base)
    pos_ins = torch.where(has_ins, start + kept, torch.full_like(start, dump)).clamp_(max=dump)
    out.scatter_(1, pos_ins, ins_base)
    out = out[:, :width]
    a0 = start[:, anchor]  # 0-based output index of the first base emitted from the anchor onwards
    ae = a0 + 1 if not five_prime else a0  # 3': AE = first adapter(rc) base; 5': AE = last adapter base
    j = torch.rand((n,), generator=g, device=device)
    ae = ae + (j < jitter / 2).to(torch.int64) - ((j >= jitter / 2) & (j < jitter)).to(torch.int64)
    if n_rate > 0:
        is_n = torch.rand((n, width), generator=g, device=device) < n_rate
        out = torch.where(is_n, torch.full_like(out, 4), out)
    return {"codes": out.to(torch.uint8), "ae": ae.to(torch.int32), "truth": truth, "umi": umi_key}


def pack_windows(codes, ae, five_prime=False):
    """The packing half of Parser.lambda$assignBarcode$4 done with torch ops (used to feed the matcher directly
    when reads are synthesised on the device): -> int64 [n, 2] = smi_bc_window records."""
    n, width = codes.shape
    W = 25 if five_prime else 24
    first = (ae.to(torch.int64) - 1) if five_prime else (ae.to(torch.int64) - 22)  # 1-based
    valid = (ae > 0) & (first >= 1) & (first + W - 1 <= width)
    idx = (first - 1).clamp(0, width - W).unsqueeze(1) + torch.arange(W, device=codes.device)
    win = torch.gather(codes.to(torch.int64), 1, idx)
    is_n = win > 3
    c = torch.where(is_n, torch.zeros_like(win), win)
    shifts = torch.arange(W - 1, -1, -1, device=codes.device, dtype=torch.int64) * 2
    bases = (c << shifts).sum(1)
    nmask = (is_n.to(torch.int64) << torch.arange(W, device=codes.device, dtype=torch.int64)).sum(1)
    bases = torch.where(valid, bases, torch.zeros_like(bases))
    nmask = torch.where(valid, nmask, torch.zeros_like(nmask))
    word1 = nmask | (valid.to(torch.int64) << 32)
    return torch.stack([bases, word1], dim=1).contiguous()


# ---------------------------------------------------------------------------------------------------------------
# whole reads (both ends) for the scan stage
# ---------------------------------------------------------------------------------------------------------------
END_BASES = 224  # bases kept per read end on the device (= SMI_END_BASES: 175 scanned + room for the barcode windows)
TSO_COMPLETE = "AAGCAGTGGTATCAACGCAGAGTACATGGG"


def _channel(src, forced_del, g, err, frac, width):
    """sub/ins/del channel; returns (out [n, width] left-aligned, total emitted length [n])"""
    n, S = src.shape
    device = src.device
    p_sub, p_ins, p_del = (err * f for f in frac)
    u = torch.rand((n, S), generator=g, device=device)
    is_del = (u < p_del) | forced_del
    is_sub = (u >= p_del) & (u < p_del + p_sub)
    sub_base = (src + torch.randint(1, 4, (n, S), generator=g, device=device, dtype=torch.int64)) & 3
    base = torch.where(is_sub, sub_base, src)
    has_ins = (torch.rand((n, S), generator=g, device=device) < p_ins) & ~forced_del
    ins_base = torch.randint(0, 4, (n, S), generator=g, device=device, dtype=torch.int64)
    kept = (~is_del).to(torch.int64)
    cnt = kept + has_ins.to(torch.int64)
    start = torch.cumsum(cnt, dim=1) - cnt
    total = cnt.sum(1)
    out = torch.randint(0, 4, (n, width + 1), generator=g, device=device, dtype=torch.int64)
    dump = width
    out.scatter_(1, torch.where(is_del, torch.full_like(start, dump), start).clamp_(max=dump), base)
    out.scatter_(1, torch.where(has_ins, start + kept, torch.full_like(start, dump)).clamp_(max=dump), ins_base)
    return out[:, :width], total.clamp(max=width)


def gen_reads(n, used_keys, seed=5, device=None, err=0.063, frac=(0.4, 0.3, 0.3), n_rate=0.0, max_mid=1500,
              q_mean=12.0, adapter_complete=ADAPTER_3P_COMPLETE, tso_complete=None, umi_len=12):
    """n synthetic 3' reads, kept as their two END_BASES-long ends + the length of the (unmaterialised) middle.

    read (transcript sense) = TSO + cDNA + polyA(20..60) + rc(UMI12) + rc(BC16) + rc(CTACACGACGCTCTTCCGATCT);
    half of the reads are reverse-complemented.  adapter_complete / tso_complete / umi_len: other config.xml sequences and UMI lengths (the
    defaults leave every seeded batch as it was).  Returns a dict of tensors:
      head, tail  uint8 [n, END_BASES]  raw read's first / last bases (2-bit codes, 4 = N)
      qhead, qtail uint8 [n, END_BASES] Phred+33 of those bases;  qmid uint8 [n] quality of every middle base
      mid_len int64 [n]; length = 2*END_BASES + mid_len; reverse bool [n]; truth int64 [n]; umi int64 [n]
    """
    device = device or used_keys.device
    g = _gen(seed, device)
    E = END_BASES
    n_cells = used_keys.numel()
    w = torch.exp(torch.randn(n_cells, generator=g, device=device))
    cell = torch.multinomial(w, n, replacement=True, generator=g)
    truth = used_keys.to(device)[cell]
    bc = keys_to_codes(truth, 16)
    umi = torch.randint(0, 4, (n, umi_len), generator=g, device=device, dtype=torch.int64)
    umi_key = (umi << (torch.arange(umi_len - 1, -1, -1, device=device, dtype=torch.int64) * 2)).sum(-1)
    # barcode end (transcript sense, right-aligned at the read's 3' end)
    pad = torch.randint(0, 4, (n, 160), generator=g, device=device, dtype=torch.int64)
    polya = torch.zeros((n, 60), dtype=torch.int64, device=device)
    ad = _rc(_codes(adapter_complete, device)).expand(n, -1)
    src = torch.cat([pad, polya, _rc(umi), _rc(bc), ad], dim=1)
    pa_len = torch.randint(20, 61, (n,), generator=g, device=device)
    forced = torch.zeros(src.shape, dtype=torch.bool, device=device)
    forced[:, 160:220] = torch.arange(60, device=device).unsqueeze(0) < (60 - pa_len).unsqueeze(1)
    W = 320
    out3, tot3 = _channel(src, forced, g, err, frac, W)
    idx = (tot3 - E).clamp(min=0).unsqueeze(1) + torch.arange(E, device=device)
    end3 = torch.gather(out3, 1, idx.clamp(max=W - 1))
    # TSO end (left-aligned at the 5' end)
    tso = _codes(tso_complete or TSO_COMPLETE, device).expand(n, -1)
    cdna = torch.randint(0, 4, (n, 260 - tso.shape[1]), generator=g, device=device, dtype=torch.int64)
    out5, _ = _channel(torch.cat([tso, cdna], dim=1), torch.zeros((n, 260), dtype=torch.bool, device=device), g, err,
                       frac, W)
    end5 = out5[:, :E]
    if n_rate > 0:
        end3 = torch.where(torch.rand((n, E), generator=g, device=device) < n_rate, torch.full_like(end3, 4), end3)
        end5 = torch.where(torch.rand((n, E), generator=g, device=device) < n_rate, torch.full_like(end5, 4), end5)
    reverse = torch.rand((n,), generator=g, device=device) < 0.5

    def rc4(x):  # reverse complement keeping N (4)
        y = torch.flip(x, dims=[1])
        return torch.where(y > 3, y, 3 - y)

    head = torch.where(reverse.unsqueeze(1), rc4(end3), end5)
    tail = torch.where(reverse.unsqueeze(1), rc4(end5), end3)
    q = lambda: (torch.randn((n, E), generator=g, device=device) * 3.0 + q_mean).round().clamp(2, 40).to(torch.uint8) + 33  # noqa: E731
    mid_len = torch.randint(0, max_mid + 1, (n,), generator=g, device=device)
    qmid = (torch.randn((n,), generator=g, device=device) * 2.0 + q_mean).round().clamp(2, 40).to(torch.uint8) + 33
    return {"head": head.to(torch.uint8), "tail": tail.to(torch.uint8), "qhead": q(), "qtail": q(), "qmid": qmid,
            "mid_len": mid_len, "reverse": reverse, "truth": truth, "umi": umi_key}


def materialize(reads, i, seed=0):
    """ASCII (read, qual) of read i for the oracle; the middle is seeded random sequence"""
    import numpy as np

    lut = np.frombuffer(b"AGCTN", dtype=np.uint8)
    m = int(reads["mid_len"][i])
    rng = np.random.default_rng(seed * 1_000_003 + i)
    mid = lut[rng.integers(0, 4, m)]
    seq = np.concatenate([lut[reads["head"][i].cpu().numpy()], mid, lut[reads["tail"][i].cpu().numpy()]])
    qual = np.concatenate([reads["qhead"][i].cpu().numpy(), np.full(m, int(reads["qmid"][i]), dtype=np.uint8),
                           reads["qtail"][i].cpu().numpy()])
    return bytes(seq).decode(), bytes(qual).decode()


_LUT24 = None


def pack_ends(head, tail):
    """-> int32 [28, 2n]: both scan-orientation ends of every read as four IUPAC bit-planes of END_BASES bits
    (plane c, word w of end e at row 7*c + w, column e; e = 2*read for the head, 2*read+1 for the reverse
    complement of the tail).  Bit p of a plane = bit c of the 4-bit code (A=1 G=2 C=4 T=8 N=15) of base p."""
    device = head.device
    n, E = head.shape
    lut = torch.tensor([1, 2, 4, 8, 15], dtype=torch.int64, device=device)
    fwd = lut[head.long()]
    t = torch.flip(tail.long(), dims=[1])
    t = torch.where(t > 3, t, 3 - t)
    rev = lut[t]
    ends = torch.stack([fwd, rev], dim=1).reshape(2 * n, E)  # [2n, E]
    nw = (E + 31) // 32
    padw = nw * 32 - E
    if padw:
        ends = torch.cat([ends, torch.zeros((2 * n, padw), dtype=torch.int64, device=device)], dim=1)
    sh = torch.arange(32, device=device, dtype=torch.int64)
    rows = []
    for c in range(4):
        bits = ((ends >> c) & 1).reshape(2 * n, nw, 32)
        words = (bits << sh).sum(-1)  # [2n, nw] values < 2^32
        rows.append(words.t())
    planes = torch.cat(rows, dim=0)  # [4*nw, 2n]
    planes = torch.where(planes >= (1 << 31), planes - (1 << 32), planes)
    return planes.to(torch.int32).contiguous()


def make_chimeras(reads, n, seed=9, parts=(1, 2, 2, 2, 3, 4)):
    """n ASCII (read, qual, n_parts) made by concatenating 1..4 materialised reads of `reads` (ligation chimeras: the
    junction carries the 3' adapter of one molecule next to the TSO / adapter of the next); host-side, tests only"""
    import numpy as np

    rng = np.random.default_rng(seed)
    n_src = reads["head"].shape[0]
    out = []
    for _ in range(n):
        k = int(parts[rng.integers(0, len(parts))])
        seqs, quals = zip(*(materialize(reads, int(rng.integers(0, n_src))) for _ in range(k)))
        out.append(("".join(seqs), "".join(quals), k))
    return out


def materialize_device(reads, seed=0, chunk=262144):
    """device-side twin of materialize() for whole batches: (ascii uint8 [total], offsets int64 [n + 1]); the middle
    of every read is random sequence (not the same bytes materialize() draws)"""
    dev = reads["head"].device
    n = reads["head"].shape[0]
    E = END_BASES
    lens = 2 * E + reads["mid_len"].to(torch.int64)
    offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offs[1:] = torch.cumsum(lens, 0)
    total = int(offs[-1])
    lut = torch.tensor(list(b"AGCTN"), dtype=torch.uint8, device=dev)
    g = _gen(seed + 77, dev)
    buf = lut[torch.randint(0, 4, (total,), generator=g, device=dev)]
    ar = torch.arange(E, device=dev)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        buf[(offs[a:b, None] + ar).reshape(-1)] = lut[reads["head"][a:b].long()].reshape(-1)
        buf[(offs[a + 1:b + 1, None] - E + ar).reshape(-1)] = lut[reads["tail"][a:b].long()].reshape(-1)
    return buf, offs


# 5' barcoding (Jar/config.xml:124-141): adapter + BC16 + UMI + TSO at the 5' end, polyA + 3' adapter at the other end
ADAPTER_5P_3PRIME = "AAGCAGTGGTATCAACGCAGAGTAC"
TSO_5P = "TTTCTTATATGGG"


def gen_reads_5p(n, used_keys, seed=6, device=None, err=0.063, frac=(0.4, 0.3, 0.3), n_rate=0.0, max_mid=1500, q_mean=12.0,
                 umi_len=12):
    """n synthetic 5'-protocol reads in the layout of gen_reads():
    read (transcript sense) = CTACACGACGCTCTTCCGATCT + BC16 + UMI + TTTCTTATATGGG + cDNA + polyA(20..60) + rc(3' adapter);
    half of the reads are reverse-complemented."""
    device = device or used_keys.device
    g = _gen(seed, device)
    E = END_BASES
    n_cells = used_keys.numel()
    w = torch.exp(torch.randn(n_cells, generator=g, device=device))
    cell = torch.multinomial(w, n, replacement=True, generator=g)
    truth = used_keys.to(device)[cell]
    bc = keys_to_codes(truth, 16)
    umi = torch.randint(0, 4, (n, umi_len), generator=g, device=device, dtype=torch.int64)
    umi_key = (umi << (torch.arange(umi_len - 1, -1, -1, device=device, dtype=torch.int64) * 2)).sum(-1)
    W = 320
    ad = _codes(ADAPTER_3P_COMPLETE, device).expand(n, -1)
    tso = _codes(TSO_5P, device).expand(n, -1)
    cdna = torch.randint(0, 4, (n, 230), generator=g, device=device, dtype=torch.int64)
    src5 = torch.cat([ad, bc, umi, tso, cdna], dim=1)
    out5, _ = _channel(src5, torch.zeros(src5.shape, dtype=torch.bool, device=device), g, err, frac, W)
    end5 = out5[:, :E]
    pad = torch.randint(0, 4, (n, 180), generator=g, device=device, dtype=torch.int64)
    polya = torch.zeros((n, 60), dtype=torch.int64, device=device)
    ad3 = _rc(_codes(ADAPTER_5P_3PRIME, device)).expand(n, -1)
    src3 = torch.cat([pad, polya, ad3], dim=1)
    pa_len = torch.randint(20, 61, (n,), generator=g, device=device)
    forced = torch.zeros(src3.shape, dtype=torch.bool, device=device)
    forced[:, 180:240] = torch.arange(60, device=device).unsqueeze(0) < (60 - pa_len).unsqueeze(1)
    out3, tot3 = _channel(src3, forced, g, err, frac, W)
    idx = (tot3 - E).clamp(min=0).unsqueeze(1) + torch.arange(E, device=device)
    end3 = torch.gather(out3, 1, idx.clamp(max=W - 1))
    if n_rate > 0:
        end3 = torch.where(torch.rand((n, E), generator=g, device=device) < n_rate, torch.full_like(end3, 4), end3)
        end5 = torch.where(torch.rand((n, E), generator=g, device=device) < n_rate, torch.full_like(end5, 4), end5)
    reverse = torch.rand((n,), generator=g, device=device) < 0.5

    def rc4(x):
        y = torch.flip(x, dims=[1])
        return torch.where(y > 3, y, 3 - y)

    head = torch.where(reverse.unsqueeze(1), rc4(end3), end5)
    tail = torch.where(reverse.unsqueeze(1), rc4(end5), end3)
    q = lambda: (torch.randn((n, E), generator=g, device=device) * 3.0 + q_mean).round().clamp(2, 40).to(torch.uint8) + 33  # noqa: E731
    mid_len = torch.randint(0, max_mid + 1, (n,), generator=g, device=device)
    qmid = (torch.randn((n,), generator=g, device=device) * 2.0 + q_mean).round().clamp(2, 40).to(torch.uint8) + 33
    return {"head": head.to(torch.uint8), "tail": tail.to(torch.uint8), "qhead": q(), "qtail": q(), "qmid": qmid,
            "mid_len": mid_len, "reverse": reverse, "truth": truth, "umi": umi_key}


def fastq_text_device(rd, chimera_frac=0.0, seed=11):
    """FASTQ text of a read batch, built on the device: records "@rNNNNNNNN\n" bases "\n+\n" qualities "\n" (qualities 'I').
    chimera_frac > 0: that fraction of the record boundaries is dropped, i.e. two neighbouring molecules become ONE record (a
    ligation chimera: 3' adapter of one next to the TSO of the next), as tools/microbench.py does for K-CHIM.
    -> (text uint8 tensor, contiguous bases, offsets)"""
    buf, offs = materialize_device(rd)
    dev = buf.device
    if chimera_frac > 0:
        n0 = offs.numel() - 1
        keep = torch.ones(n0 + 1, dtype=torch.bool, device=dev)
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        keep[1:n0][torch.rand(n0 - 1, device=dev, generator=g) < chimera_frac] = False
        offs = offs[keep].contiguous()
    n = offs.numel() - 1
    lens = offs[1:] - offs[:-1]
    rec_len = 2 * lens + 15  # 11 + len + 3 + len + 1
    rec_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rec_off[1:] = torch.cumsum(rec_len, 0)
    total = int(rec_off[-1])
    text = torch.full((total,), ord("I"), dtype=torch.uint8, device=dev)
    idx = torch.arange(n, device=dev)
    text[rec_off[:-1]] = ord("@")
    text[rec_off[:-1] + 1] = ord("r")
    for k in range(8):
        text[rec_off[:-1] + 2 + k] = (48 + (idx // 10 ** (7 - k)) % 10).to(torch.uint8)
    text[rec_off[:-1] + 10] = 10
    text[rec_off[:-1] + 11 + lens] = 10
    text[rec_off[:-1] + 12 + lens] = ord("+")
    text[rec_off[:-1] + 13 + lens] = 10
    text[rec_off[1:] - 1] = 10
    pos = torch.arange(int(offs[-1]), device=dev) - torch.repeat_interleave(offs[:-1], lens) + \
        torch.repeat_interleave(rec_off[:-1] + 11, lens)
    text[pos] = buf
    return text, buf, offs


def bam_from_rows(rows, ref_names=("chr1",), read_len=1200, seed=0):
    """test / bench input: an uncompressed BAM stream (SAM specification 4.2) of mapped records -- rows: (0-based position, read name, FLAG
    [, reference index]) (on the first reference when no index is given), one `read_len`M operation, random bases and qualities, the aux fields minimap2 writes (NM ms AS nn tp cm s1 s2
    de rl, in its order).  -> bytes"""
    import struct

    import numpy as np

    rng = np.random.default_rng(seed)
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{r}\tLN:250000000\n" for r in ref_names)
    out = [b"BAM\1", struct.pack("<i", len(text)), text.encode(), struct.pack("<i", len(ref_names))]
    for r in ref_names:
        out.append(struct.pack("<i", len(r) + 1) + r.encode() + b"\0" + struct.pack("<i", 250000000))
    half = (read_len + 1) // 2
    codes = np.array([1, 2, 4, 8], dtype=np.uint8)                                   # A C G T as 4-bit codes, two per byte
    pairs = codes[rng.integers(0, 4, (len(rows), half))] << 4 | codes[rng.integers(0, 4, (len(rows), half))]
    quals = rng.integers(2, 50, (len(rows), read_len), dtype=np.uint8)
    cigar = struct.pack("<I", (read_len << 4) | 0)
    for k, row in enumerate(rows):
        pos, name, flag = row[:3]
        ref = row[3] if len(row) > 3 else 0
        seq, qual = pairs[k].tobytes(), quals[k].tobytes()
        nm = name.encode() + b"\0"
        aux = (b"NMi" + struct.pack("<i", int(rng.integers(0, 90))) + b"msi" + struct.pack("<i", 900) + b"ASi" + struct.pack("<i", 880) + b"nni\0\0\0\0" +
               b"tpAP" + b"cmi" + struct.pack("<i", 120) + b"s1i" + struct.pack("<i", 700) + b"s2i\0\0\0\0" + b"def" + struct.pack("<f", 0.05) + b"rli\0\0\0\0")
        b, e = 0, pos + read_len - 1                                          # reg2bin (SAM specification 5.3)
        for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
            if pos >> shift == e >> shift:
                b = base + (pos >> shift)
                break
        body = struct.pack("<iiBBHHHiiii", int(ref), int(pos), len(nm), 60, b & 0xFFFF, 1, int(flag), read_len, -1, -1, 0) + nm + cigar + seq + qual + aux
        out.append(struct.pack("<i", len(body)) + body)
    return b"".join(out)
