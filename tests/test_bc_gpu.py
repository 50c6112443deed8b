"""GPU parity tests (through the C ABI): HIP barcode assignment == oracle, bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FIELDS = ("bc", "ed", "ed_sec", "offset", "ins_minus_del")


def _compare(pkg, got, st, exp):
    exp_found = np.where(st < 0, st, exp["found"])  # -1: window outside the read, -2: tree bin (see sicelore_mi.h)
    bad = np.nonzero(got["found"] != exp_found)[0]
    assert bad.size == 0, f"found differs at {bad[:10]}: got {got['found'][bad[:10]]} exp {exp_found[bad[:10]]}"
    sel = exp_found == 1
    for f in FIELDS:
        g = got[f][sel].astype(np.int64)
        e = exp[f][sel].astype(np.int64) & (0xFFFFFFFF if f == "bc" else -1)
        bad = np.nonzero(g != e)[0]
        assert bad.size == 0, f"{f} differs at {bad[:10]}"
    nm = np.nonzero((got["n_matches"] != exp["n_matches"]) & (st >= 0))[0]
    assert nm.size == 0, f"n_matches differs at {nm[:10]}"
    return int(sel.sum())


def _run_device(pkg, ctx, win, max_ed, five_prime):
    n = win.shape[0]
    d_win = win.cuda()
    d_out = torch.zeros((max(n, 1), 4), dtype=torch.int32, device="cuda")
    ctx.bc_match_device(d_win, d_out, n, max_ed=max_ed, five_prime=five_prime)
    torch.cuda.synchronize()
    return d_out.cpu().numpy().view(pkg.BC_RESULT_DTYPE).reshape(-1)[:n]


@pytest.mark.parametrize("five_prime", [False, True])
@pytest.mark.parametrize("max_ed", [1, 0])
def test_used_list_mode(pkg, synth, sor, gpu_ctx, five_prime, max_ed):
    wl = synth.make_whitelist(100_000, seed=21)
    used = synth.pick_used(wl, 5000, seed=22)
    reg = synth.gen_bc_region(100_000, used, seed=23 + five_prime, five_prime=five_prime)
    win = synth.pack_windows(reg["codes"], reg["ae"], five_prime)
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    got = _run_device(pkg, gpu_ctx, win, max_ed, five_prime)
    st, exp = sor.assign_batch(sor.BarcodeSet(used.numpy()), reg["codes"].numpy(), reg["ae"].numpy(), max_ed=max_ed,
                               five_prime=five_prime, n_threads=8)
    n_found = _compare(pkg, got, st, exp)
    assert n_found > (30_000 if max_ed else 10_000)
    # host-buffer entry point gives the same answer
    got2 = gpu_ctx.bc_match(win.numpy().view(pkg.BC_WINDOW_DTYPE).reshape(-1), max_ed=max_ed, five_prime=five_prime)
    assert (got2.view(np.uint8) == got.view(np.uint8)).all()


@pytest.mark.parametrize("five_prime", [False, True])
def test_whitelist_mode_3p6M(pkg, synth, sor, gpu_ctx, five_prime):
    """BASELINE.json configs[1] shape: ed <= 1 against the 3.6 M-entry whitelist (search set = whole list)"""
    wl = synth.make_whitelist(3_600_000, seed=1)
    used = synth.pick_used(wl, 5000, seed=2)
    reg = synth.gen_bc_region(200_000, used, seed=31 + five_prime, five_prime=five_prime, n_rate=0.001)
    win = synth.pack_windows(reg["codes"], reg["ae"], five_prime)
    gpu_ctx.set_barcode_set(wl.numpy().astype(np.uint64), mode=1)
    got = _run_device(pkg, gpu_ctx, win, 1, five_prime)
    st, exp = sor.assign_batch(sor.BarcodeSet(wl.numpy()), reg["codes"].numpy(), reg["ae"].numpy(), max_ed=1,
                               five_prime=five_prime, n_threads=8)
    n_found = _compare(pkg, got, st, exp)
    assert n_found > 60_000
    # sanity, not parity: against the whole 3.6 M list ~0.5 spurious ed<=1 neighbours per read are expected
    # (620 probes x 3.6e6 / 2^32), so accuracy is far below the two-pass mode's -- the reference's reason for pass 1
    sel = got["found"] == 1
    acc = (got["bc"][sel] == (reg["truth"].numpy()[sel] & 0xFFFFFFFF)).mean()
    assert acc > 0.8


def test_n_rich_and_edge_windows(pkg, synth, sor, gpu_ctx):
    wl = synth.make_whitelist(50_000, seed=41)
    used = synth.pick_used(wl, 300, seed=42)
    gpu_ctx.set_barcode_set(wl.numpy().astype(np.uint64), mode=1)
    bset = sor.BarcodeSet(wl.numpy())
    for five_prime in (False, True):
        reg = synth.gen_bc_region(30_000, used, seed=43, five_prime=five_prime, n_rate=0.05)
        codes, ae = reg["codes"].clone(), reg["ae"].clone()
        # windows that do not fit the read: the reference throws -> found = -1
        ae[:200] = torch.arange(-50, 150, dtype=torch.int32)
        ae[200:300] = codes.shape[1] - torch.arange(0, 100, dtype=torch.int32)
        # homopolymers: all five windows identical (same HashSet bucket)
        codes[300:310] = 0
        codes[310:320] = 3
        win = synth.pack_windows(codes, ae, five_prime)
        got = _run_device(pkg, gpu_ctx, win, 1, five_prime)
        st, exp = sor.assign_batch(bset, codes.numpy(), ae.numpy(), max_ed=1, five_prime=five_prime, n_threads=8)
        assert (st < 0).sum() > 50
        _compare(pkg, got, st, exp)


def test_adversarial_sets(pkg, synth, sor, gpu_ctx):
    """dense neighbourhoods: every window has many barcodes at distance <= 1 at several offsets, so first-hit
    order, the insert-at-14 wrap and the HashSet bucket order all decide results"""
    rng = np.random.default_rng(5)
    n = 20_000
    wl = synth.make_whitelist(2000, seed=51)
    used = synth.pick_used(wl, 200, seed=52)
    reg = synth.gen_bc_region(n, used, seed=53, err=0.03)
    codes, ae = reg["codes"].numpy(), reg["ae"].numpy()
    # barcode set = all 16-mers seen at offsets -2..2 of the first 400 reads (rc), plus random neighbours of them
    keys = []
    for i in range(400):
        for o in (-2, -1, 0, 1, 2):
            a = int(ae[i]) - 16 + o - 1
            if a < 0:
                continue
            w = codes[i, a:a + 16]
            k = 0
            for c in w[::-1]:
                k = (k << 2) | (3 - int(c))
            keys.append(k)
    keys = np.array(keys, dtype=np.uint64)
    nb = keys.copy()
    pos = rng.integers(0, 16, nb.size)
    nb ^= (rng.integers(1, 4, nb.size).astype(np.uint64) << (2 * pos).astype(np.uint64))
    allk = np.unique(np.concatenate([keys, nb, wl.numpy().astype(np.uint64)]))
    gpu_ctx.set_barcode_set(allk, mode=0)
    win = synth.pack_windows(reg["codes"], reg["ae"])
    got = _run_device(pkg, gpu_ctx, win, 1, False)
    st, exp = sor.assign_batch(sor.BarcodeSet(allk.astype(np.int64)), codes, ae, max_ed=1, n_threads=8)
    _compare(pkg, got, st, exp)
    assert (exp["n_matches"] >= 3).sum() > 100


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 257])
def test_ragged_batch_sizes(pkg, synth, sor, gpu_ctx, n):
    wl = synth.make_whitelist(10_000, seed=61)
    used = synth.pick_used(wl, 100, seed=62)
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    reg = synth.gen_bc_region(max(n, 1), used, seed=63)
    codes, ae = reg["codes"][:n], reg["ae"][:n]
    win = synth.pack_windows(codes, ae) if n else torch.zeros((0, 2), dtype=torch.int64)
    got = _run_device(pkg, gpu_ctx, win, 1, False)
    assert got.shape[0] == n
    if n:
        st, exp = sor.assign_batch(sor.BarcodeSet(used.numpy()), codes.numpy(), ae.numpy(), max_ed=1)
        _compare(pkg, got, st, exp)


def test_extract_windows_kernel(pkg, synth, gpu_ctx):
    """K-WIN == the torch packing used by the generator, incl. N and out-of-read cases"""
    wl = synth.make_whitelist(10_000, seed=71)
    used = synth.pick_used(wl, 100, seed=72)
    for five_prime in (False, True):
        reg = synth.gen_bc_region(5000, used, seed=73, five_prime=five_prime, n_rate=0.02)
        codes, ae = reg["codes"], reg["ae"].clone()
        ae[:100] = torch.arange(-20, 80, dtype=torch.int32)
        ae[100:200] = codes.shape[1] - torch.arange(0, 100, dtype=torch.int32)
        lut = torch.tensor(list(b"AGCTN"), dtype=torch.uint8)
        reads = lut[codes.long()].reshape(-1).cuda()
        offsets = (torch.arange(0, 5001, dtype=torch.int64) * codes.shape[1]).cuda()
        d_win = torch.zeros((5000, 2), dtype=torch.int64, device="cuda")
        gpu_ctx.extract_windows_device(reads, offsets, ae.cuda(), d_win, 5000, five_prime=five_prime)
        torch.cuda.synchronize()
        exp = synth.pack_windows(codes, ae, five_prime)
        assert (d_win.cpu() == exp).all()


def test_full_size_properties(pkg, synth, gpu_ctx):
    """BASELINE.json configs[1] at full size (10 M reads, 3.6 M whitelist): size-independent properties --
    determinism, every assigned barcode is a whitelist member within the stated distance, and permutation
    invariance (results travel with their reads)."""
    n = 10_000_000
    wl = synth.make_whitelist(3_600_000, seed=1, device="cuda")
    used = synth.pick_used(wl, 5000, seed=2)
    gpu_ctx.set_barcode_set_device(wl.to(torch.int32), mode=1)
    reg = synth.gen_bc_region(n, used, seed=3, device="cuda")
    win = synth.pack_windows(reg["codes"], reg["ae"])
    out1 = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    out2 = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    gpu_ctx.bc_match_device(win, out1, n, max_ed=1)
    gpu_ctx.bc_match_device(win, out2, n, max_ed=1)
    torch.cuda.synchronize()
    assert bool((out1 == out2).all())
    perm = torch.randperm(n, device="cuda")
    out3 = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    gpu_ctx.bc_match_device(win[perm].contiguous(), out3, n, max_ed=1)
    torch.cuda.synchronize()
    assert bool((out3 == out1[perm]).all())
    found = (out1[:, 2] & 0xFF) == 1
    bc = out1[:, 0].to(torch.int64) & 0xFFFFFFFF
    assert float(found.float().mean()) > 0.3
    # membership of every assigned barcode
    srt = torch.sort(wl).values
    idx = torch.searchsorted(srt, bc[found]).clamp(max=srt.numel() - 1)
    assert bool((srt[idx] == bc[found]).all())
    acc = float((bc[found] == reg["truth"][found]).float().mean())
    assert acc > 0.8


@pytest.mark.parametrize("five_prime", [False, True])
def test_ed2_used_list_mode(pkg, synth, sor, gpu_ctx, five_prime):
    """BASELINE.json configs[2] shape: ed <= 2 against the used list (pass 2 of the two-pass mode)"""
    wl = synth.make_whitelist(100_000, seed=301)
    used = synth.pick_used(wl, 3000, seed=302)
    n = 20_000
    reg = synth.gen_bc_region(n, used, seed=303 + five_prime, five_prime=five_prime, n_rate=0.002, err=0.08)
    win = synth.pack_windows(reg["codes"], reg["ae"], five_prime)
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    got = _run_device(pkg, gpu_ctx, win, 2, five_prime)
    st, exp = sor.assign_batch(sor.BarcodeSet(used.numpy()), reg["codes"].numpy(), reg["ae"].numpy(), max_ed=2,
                               five_prime=five_prime, n_threads=8)
    n_found = _compare(pkg, got, st, exp)
    assert n_found > 0.6 * n
    assert (exp["ed"][exp["found"] == 1] == 2).sum() > 200


@pytest.mark.parametrize("five_prime", [False, True])
@pytest.mark.parametrize("n_wl", [100_000, 400_000, 3_600_000])
def test_ed2_long_lists_equal_oracle(pkg, synth, sor, gpu_ctx, n_wl, five_prime):
    """K-BC2 against the oracle where the search set is the WHOLE list (-g semantics) and longer than 65,536 keys: the path through the
    neighbourhood table behind P.nb (no item filter, no two-step bitmap), and -- with SMI_BC2_DENSE_ENUM on lists above 300 k keys -- the
    kernel that enumerates instead; 4,000 reads per case, windows with N bases among them"""
    import os

    wl = synth.make_whitelist(n_wl, seed=331 + n_wl % 97)
    used = synth.pick_used(wl, 3000, seed=332)
    n = 4000
    reg = synth.gen_bc_region(n, used, seed=333 + five_prime, five_prime=five_prime, n_rate=0.002, err=0.08)
    win = synth.pack_windows(reg["codes"], reg["ae"], five_prime)
    gpu_ctx.set_barcode_set(wl.numpy().astype(np.uint64), mode=1)
    st, exp = sor.assign_batch(sor.BarcodeSet(wl.numpy()), reg["codes"].numpy(), reg["ae"].numpy(), max_ed=2, five_prime=five_prime,
                               n_threads=16)
    got = _run_device(pkg, gpu_ctx, win, 2, five_prime)
    n_found = _compare(pkg, got, st, exp)
    # (against the whole 3.6 M list most windows have several barcodes within two edits: few reads are assigned, which is the point)
    assert n_found > 100 and (exp["n_matches"] >= 2).sum() > 100
    if n_wl > 300_000:
        os.environ["SMI_BC2_DENSE_ENUM"] = "1"
        try:
            got2 = _run_device(pkg, gpu_ctx, win, 2, five_prime)
        finally:
            del os.environ["SMI_BC2_DENSE_ENUM"]
        assert got2.tobytes() == got.tobytes()


def test_ed2_dense_and_degenerate(pkg, synth, sor, gpu_ctx):
    """ed <= 2 where the dedup set decides: dense neighbourhoods (many ed-1/ed-2 barcodes per window), homopolymer
    and low-complexity windows (equal children at different positions), all-T keys (hash sentinel) and N bases"""
    rng = np.random.default_rng(11)
    wl = synth.make_whitelist(3000, seed=311)
    used = synth.pick_used(wl, 300, seed=312)
    n = 6000
    reg = synth.gen_bc_region(n, used, seed=313, err=0.05, n_rate=0.01)
    codes, ae = reg["codes"].clone(), reg["ae"].clone()
    # low-complexity reads
    codes[:200] = 3                                            # all T -> key 0xFFFFFFFF in 5' / all A after revcomp
    codes[200:400] = 0                                         # all A -> all T after revcomp (3')
    codes[400:600] = torch.tensor([0, 3], dtype=torch.uint8).repeat(64)[None, :]            # ATAT...
    codes[600:800] = torch.tensor([2, 2, 1, 1], dtype=torch.uint8).repeat(32)[None, :]      # CCGG...
    for i in range(800, 1200):                                                               # runs with one break
        codes[i] = int(rng.integers(4))
        codes[i, rng.integers(30, 90, size=3)] = torch.tensor(rng.integers(0, 4, size=3), dtype=torch.uint8)
    keys = []
    cn = codes.numpy()
    for i in list(range(0, 1200, 7)) + list(range(1200, 2200, 5)):
        for o in (-2, -1, 0, 1, 2):
            a = int(ae[i]) - 16 + o - 1
            if a < 0 or a + 16 > cn.shape[1]:
                continue
            k = 0
            for c in cn[i, a:a + 16][::-1]:
                k = (k << 2) | (3 - min(int(c), 3))
            keys.append(k)
    keys = np.array(keys, dtype=np.uint64)
    nb1 = keys ^ (rng.integers(1, 4, keys.size).astype(np.uint64) << (2 * rng.integers(0, 16, keys.size)).astype(np.uint64))
    nb2 = nb1 ^ (rng.integers(1, 4, keys.size).astype(np.uint64) << (2 * rng.integers(0, 16, keys.size)).astype(np.uint64))
    allk = np.unique(np.concatenate([keys[::3], nb1, nb2, used.numpy().astype(np.uint64),
                                     np.array([0xFFFFFFFF, 0, 0x33333333, 0xCCCCCCCC], dtype=np.uint64)]))
    for five_prime in (False, True):
        gpu_ctx.set_barcode_set(allk, mode=0)
        win = synth.pack_windows(codes, ae, five_prime)
        got = _run_device(pkg, gpu_ctx, win, 2, five_prime)
        st, exp = sor.assign_batch(sor.BarcodeSet(allk.astype(np.int64)), cn, ae.numpy(), max_ed=2,
                                   five_prime=five_prime, n_threads=8)
        _compare(pkg, got, st, exp)
        assert (exp["n_matches"] >= 4).sum() > 100


def test_pass2_counters_per_barcode_and_ed(pkg, synth, gpu_ctx):
    """K-CNT: counts[3 * ordinal(bc) + ed] over the results of a batch == numpy over the same results; accumulates"""
    import torch

    dev = torch.device("cuda", gpu_ctx.device)
    wl = synth.make_whitelist(30_000, seed=71, device=dev)
    used = synth.pick_used(wl, 400, seed=72)
    keys = np.sort(used.cpu().numpy().astype(np.uint64))
    gpu_ctx.set_barcode_set(keys, mode=0)
    n = 50_000
    rd = synth.gen_reads(n, used, seed=73, device=dev)
    ends = synth.pack_ends(rd["head"], rd["tail"])
    lens = (2 * synth.END_BASES + rd["mid_len"]).to(torch.int32)
    scan = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    win = torch.zeros((n, 2), dtype=torch.int64, device=dev)
    gpu_ctx.scan_device(ends, lens, n, gpu_ctx.scan_config(2), scan, win)
    counts = torch.zeros((keys.size, 3), dtype=torch.int32, device=dev)
    exp = np.zeros((keys.size, 3), dtype=np.int64)
    for ed in (2, 1):
        res = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        gpu_ctx.bc_match_device(win, res, n, max_ed=ed)
        gpu_ctx.bc_counts_device(res, n, counts)
        r = res.cpu().numpy().view(pkg.BC_RESULT_DTYPE).reshape(-1)
        ok = r["found"] == 1
        np.add.at(exp, (np.searchsorted(keys, r["bc"][ok].astype(np.uint64)), r["ed"][ok].astype(np.int64)), 1)
    got = counts.cpu().numpy()
    assert (got == exp).all() and exp[:, 0].sum() > 10_000 and exp[:, 1].sum() > 5_000 and exp[:, 2].sum() > 100
    from sicelore_amd import lib as libmod

    tsv = libmod.assigned_tsv(keys, got.astype(np.uint32), max_ed=2).splitlines()
    assert tsv[0] == "Barcode\tn Reads with ED<=2 match\tED=0\tED=1\tED=2" and len(tsv) == 1 + int((exp.sum(1) > 0).sum())
    tot = [int(l.split("\t")[1].replace(",", "")) for l in tsv[1:]]
    assert tot == sorted(tot, reverse=True) and sum(tot) == exp.sum()


@pytest.mark.parametrize("case", ["ed1_whitelist", "ed1_used_list", "ed2_used_list", "ed1_adversarial"])
def test_filters_change_nothing(pkg, synth, gpu_ctx, monkeypatch, case):
    """K-BC1's offset filter and K-BC2's item filter (the inverse one-step neighbourhood of the barcode set, smi_bc.hip) against the same
    kernels without them: byte-identical result records.  The switches are read when a set is loaded."""
    import os
    five_prime = False
    if case == "ed1_whitelist":
        wl = synth.make_whitelist(1_000_000, seed=61)
        keys, max_ed, mode = wl.numpy().astype(np.uint64), 1, 1
        reg = synth.gen_bc_region(150_000, synth.pick_used(wl, 3000, seed=62), seed=63, five_prime=five_prime, n_rate=0.002)
    elif case == "ed1_used_list":
        wl = synth.make_whitelist(200_000, seed=64)
        used = synth.pick_used(wl, 4000, seed=65)
        keys, max_ed, mode = used.numpy().astype(np.uint64), 1, 0
        reg = synth.gen_bc_region(150_000, used, seed=66, five_prime=True)
        five_prime = True
    elif case == "ed2_used_list":
        wl = synth.make_whitelist(200_000, seed=67)
        used = synth.pick_used(wl, 4000, seed=68)
        keys, max_ed, mode = used.numpy().astype(np.uint64), 2, 0
        reg = synth.gen_bc_region(30_000, used, seed=69, five_prime=five_prime, n_rate=0.002)
    else:
        # barcodes that are one and two steps away from each other, homopolymers and short repeats: where different mutations of a
        # window coincide and where a child of a barcode has a second barcode in reach
        rng = np.random.default_rng(70)
        base = rng.integers(0, 1 << 32, 300, dtype=np.uint64)
        near = [int(b) ^ (int(rng.integers(1, 4)) << (2 * int(rng.integers(0, 16)))) for b in base]
        near2 = [n ^ (int(rng.integers(1, 4)) << (2 * int(rng.integers(0, 16)))) for n in near]
        shifted = [((int(b) << 2) | int(rng.integers(0, 4))) & 0xFFFFFFFF for b in base] + [int(b) >> 2 for b in base]
        homo = [0, 0xFFFFFFFF, 0x55555555, 0xAAAAAAAA, 0x33333333, 0xCCCCCCCC, 0x0F0F0F0F]
        keys = np.unique(np.array(list(map(int, base)) + near + near2 + shifted + homo, dtype=np.uint64))
        used = torch.from_numpy(keys.astype(np.int64))
        max_ed, mode = 1, 0
        reg = synth.gen_bc_region(60_000, used, seed=71, five_prime=five_prime, err=0.08)
    win = synth.pack_windows(reg["codes"], reg["ae"], five_prime)
    gpu_ctx.set_barcode_set(keys, mode=mode)
    with_filter = _run_device(pkg, gpu_ctx, win, max_ed, five_prime)
    got2 = None
    if case == "ed1_adversarial":  # the same set through K-BC2 as well
        got2 = _run_device(pkg, gpu_ctx, win, 2, five_prime)
    monkeypatch.setenv("SMI_BC1_NO_NB5", "1")    # K-BC1's table path with the plain 512 MiB filter bitmap instead of the one laid out by the shared core (nb5)
    gpu_ctx.set_barcode_set(keys, mode=mode)
    assert (with_filter.view(np.uint8) == _run_device(pkg, gpu_ctx, win, max_ed, five_prime).view(np.uint8)).all()
    monkeypatch.delenv("SMI_BC1_NO_NB5")
    if max_ed == 2 or got2 is not None:
        monkeypatch.setenv("SMI_BC2_ONE_FILTER", "1")   # K-BC2's item filter in its prefix-major layout only (round 6: the children of positions <= 6 read a suffix-major copy)
        gpu_ctx.set_barcode_set(keys, mode=mode)
        assert (_run_device(pkg, gpu_ctx, win, 2, five_prime).view(np.uint8) == (with_filter if max_ed == 2 else got2).view(np.uint8)).all()
        monkeypatch.delenv("SMI_BC2_ONE_FILTER")
    monkeypatch.setenv("SMI_BC1_NO_TABLE", "1")  # K-BC1: offset filter, mutants of the flagged offsets enumerated (k_bc_match_ed1f)
    monkeypatch.setenv("SMI_BC2_NO_TABLE", "1")  # K-BC2: level 2 by enumeration of the items the filter lets through (read at launch)
    gpu_ctx.set_barcode_set(keys, mode=mode)
    no_table = _run_device(pkg, gpu_ctx, win, max_ed, five_prime)
    assert (with_filter.view(np.uint8) == no_table.view(np.uint8)).all()
    if got2 is not None:
        assert (got2.view(np.uint8) == _run_device(pkg, gpu_ctx, win, 2, five_prime).view(np.uint8)).all()
    monkeypatch.setenv("SMI_BC1_NO_FILTER", "1")
    monkeypatch.setenv("SMI_BC2_NO_FILTER", "1")
    monkeypatch.setenv("SMI_BC2_NO_OFFSET_FILTER", "1")
    gpu_ctx.set_barcode_set(keys, mode=mode)
    without = _run_device(pkg, gpu_ctx, win, max_ed, five_prime)
    assert (with_filter.view(np.uint8) == without.view(np.uint8)).all()
    if got2 is not None:
        assert (got2.view(np.uint8) == _run_device(pkg, gpu_ctx, win, 2, five_prime).view(np.uint8)).all()
    assert int((with_filter["found"] == 1).sum()) > 1000
    monkeypatch.delenv("SMI_BC1_NO_FILTER")
    monkeypatch.delenv("SMI_BC1_NO_TABLE")
    monkeypatch.delenv("SMI_BC2_NO_TABLE")
    # K-BC2 without its filters but WITH the table (what a list of 65,537 .. 300,000 barcodes runs): every created item is looked up
    if max_ed == 2 or got2 is not None:
        gpu_ctx.set_barcode_set(keys, mode=mode)
        assert (_run_device(pkg, gpu_ctx, win, 2, five_prime).view(np.uint8) == (with_filter if max_ed == 2 else got2).view(np.uint8)).all()
    monkeypatch.delenv("SMI_BC2_NO_FILTER")
    monkeypatch.delenv("SMI_BC2_NO_OFFSET_FILTER")
    assert "SMI_BC1_NO_FILTER" not in os.environ
    gpu_ctx.set_barcode_set(keys, mode=mode)  # leave the context with its filters on


def test_membership_only_set_answers_like_the_full_one(pkg, synth, gpu_ctx):
    """SMI_SET_MEMBERSHIP (what pass 1 loads: the pyramid, none of the matchers' neighbourhood structures): a matcher call on it goes through the pyramid
    kernels and gives the full set's results; the pass-1 histogram is the same"""
    from sicelore_amd import lib as libmod

    wl = synth.make_whitelist(300_000, seed=91)
    used = synth.pick_used(wl, 2000, seed=92)
    reg = synth.gen_bc_region(40_000, used, seed=93, n_rate=0.002)
    win = synth.pack_windows(reg["codes"], reg["ae"], False)
    keys = wl.numpy().astype(np.uint64)
    gpu_ctx.set_barcode_set(keys, mode=libmod.SET_WHITELIST)
    full = {ed: _run_device(pkg, gpu_ctx, win, ed, False) for ed in (0, 1)}
    gpu_ctx.set_barcode_set(keys, mode=libmod.SET_MEMBERSHIP)
    for ed in (0, 1):
        assert (_run_device(pkg, gpu_ctx, win, ed, False).view(np.uint8) == full[ed].view(np.uint8)).all()
    assert int((full[1]["found"] == 1).sum()) > 10_000
    gpu_ctx.set_barcode_set(keys, mode=libmod.SET_WHITELIST)


@pytest.mark.parametrize("max_ed", [0, 1, 2])
def test_empty_and_single_key_sets(pkg, synth, sor, gpu_ctx, max_ed):
    """the two smallest barcode sets: none (every structure empty, no filter or table built) and one barcode"""
    wl = synth.make_whitelist(1000, seed=81)
    used = synth.pick_used(wl, 1, seed=82)
    reg = synth.gen_bc_region(3000, used, seed=83)
    win = synth.pack_windows(reg["codes"], reg["ae"], False)
    gpu_ctx.set_barcode_set(np.zeros(0, dtype=np.uint64), mode=0)
    got = _run_device(pkg, gpu_ctx, win, max_ed, False)
    assert (got["found"][got["found"] >= 0] == 0).all() and (got["n_matches"] == 0).all()
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    got = _run_device(pkg, gpu_ctx, win, max_ed, False)
    st, exp = sor.assign_batch(sor.BarcodeSet(used.numpy()), reg["codes"].numpy(), reg["ae"].numpy(), max_ed=max_ed, n_threads=8)
    assert _compare(pkg, got, st, exp) > (500 if max_ed else 100)


@pytest.mark.parametrize("n_keys", [3_600_000, 5_000, 1])
def test_nb5_by_transposition_equals_nb5_by_atomics(pkg, synth, gpu_ctx, monkeypatch, n_keys):
    """round 6: the 2.5 GiB offset filter `nb5` is DERIVED from the 512 MiB bitmap `nb` by a streaming transposition (k_nb5_from_nb) instead of five
    scattered atomics per neighbourhood member (round 5's k_set_nb, still there behind SMI_BC1_NB5_ATOMIC): both builds leave the same bits -- same
    count, same order-sensitive digest -- and nb5 holds five bits (one per offset) for every bit of nb; the table holds one entry per (barcode, step)"""
    wl = synth.make_whitelist(max(n_keys, 16), seed=71)[:n_keys]
    keys = wl.numpy().astype(np.uint64)
    monkeypatch.setenv("SMI_BC1_NB5_TRANSPOSE", "1")   # (short lists take the atomics by themselves: the transposition costs 3.7 ms whatever the list holds)
    gpu_ctx.set_barcode_set(keys, mode=1)
    a = gpu_ctx.set_stats(digests=True)
    monkeypatch.delenv("SMI_BC1_NB5_TRANSPOSE")
    monkeypatch.setenv("SMI_BC1_NB5_ATOMIC", "1")
    gpu_ctx.set_barcode_set(keys, mode=1)
    b = gpu_ctx.set_stats(digests=True)
    monkeypatch.delenv("SMI_BC1_NB5_ATOMIC")
    gpu_ctx.set_barcode_set(keys, mode=1)              # the default for this list's size
    assert gpu_ctx.set_stats(digests=True)["nb5_digest"] == a["nb5_digest"]
    assert a["keys"] == b["keys"] == n_keys
    assert a["nb_bits"] == b["nb_bits"] > 100 * n_keys
    assert a["nb5_bits"] == b["nb5_bits"] == 5 * a["nb_bits"]
    assert a["nb5_digest"] == b["nb5_digest"] != 0
    assert a["nt_entries"] == b["nt_entries"] and a["nt_slots"] >= a["nt_entries"] >= a["nb_bits"]
    assert a["hbm_bytes"] == b["hbm_bytes"] > (1 << 29) and a["build_ms"] > 0
    if n_keys == 3_600_000:
        print(f"set build: transposed {a['build_ms']:.1f} ms, atomics {b['build_ms']:.1f} ms, {a['hbm_bytes'] / 1e9:.2f} GB")
