"""GPU parity: K-UMI pair-distance matrices == oracle, byte for byte."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_groups(seed, sizes):
    rng = np.random.default_rng(seed)
    ws = []
    for n in sizes:
        n_umi = max(1, n // 3)
        umis = rng.choice([1, 2, 4, 8], size=(n_umi, 14)).astype(np.uint8)
        for _ in range(n):
            w = umis[rng.integers(n_umi)].copy()
            for _ in range(rng.integers(0, 3)):
                p = rng.integers(14)
                op = rng.integers(3)
                if op == 0:
                    w[p] = rng.choice([1, 2, 4, 8, 15])
                elif op == 1:
                    w[p + 1:] = w[p:-1]
                    w[p] = rng.choice([1, 2, 4, 8])
                else:
                    w[p:-1] = w[p + 1:]
                    w[-1] = rng.choice([1, 2, 4, 8])
            ws.append(w)
    return np.array(ws, dtype=np.uint8)


def _pack(ws):
    out = np.zeros(ws.shape[0], dtype=np.uint64)
    for k in range(14):
        out |= ws[:, k].astype(np.uint64) << np.uint64(4 * k)
    return out


# (groups above 64 reads are computed tile by tile, 8 x 8 blocks with 8-byte stores where a block is whole: sizes on both sides of every edge)
@pytest.mark.parametrize("sizes", [[2, 3, 1, 7, 64, 5, 130, 2, 2, 33], [400], [1] * 50 + [2] * 300,
                                   [65, 3, 66, 71, 72, 73, 1, 127, 128, 129, 64, 63, 191, 192, 193], [257, 2, 321]])
def test_umi_matrices_match_oracle(pkg, sor, gpu_ctx, sizes):
    ws = _make_groups(len(sizes), sizes)
    go, po, mo = gpu_ctx.umi_offsets(sizes)
    d_w = torch.from_numpy(_pack(ws).view(np.int64)).cuda()
    d_go = torch.from_numpy(go.view(np.int32)).cuda()
    d_po = torch.from_numpy(po.view(np.int64)).cuda()
    d_mo = torch.from_numpy(mo.view(np.int64)).cuda()
    d_out = torch.full((int(mo[-1]),), 255, dtype=torch.uint8, device="cuda")
    gpu_ctx.umi_dist_device(d_w, d_go, d_po, d_mo, len(sizes), int(po[-1]), d_out)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    for g, n in enumerate(sizes):
        exp = sor.umi_matrix(ws[go[g]:go[g + 1]])
        got = out[int(mo[g]):int(mo[g + 1])].reshape(n, n)
        assert (got == exp).all(), g


def test_umi_distances_then_clustering_end_to_end(pkg, sor, gpu_ctx):
    """K-UMI matrices (device) -> host clustering (product) == oracle matrices -> oracle clustering, per group"""
    from sicelore_amd import lib as libmod

    sizes = [2, 3, 9, 40, 100, 101, 150, 7, 64, 1, 250]
    ws = _make_groups(77, sizes)
    go, po, mo = gpu_ctx.umi_offsets(sizes)
    d_out = torch.zeros((int(mo[-1]),), dtype=torch.uint8, device="cuda")
    gpu_ctx.umi_dist_device(torch.from_numpy(_pack(ws).view(np.int64)).cuda(), torch.from_numpy(go.view(np.int32)).cuda(),
                            torch.from_numpy(po.view(np.int64)).cuda(), torch.from_numpy(mo.view(np.int64)).cuda(),
                            len(sizes), int(po[-1]), d_out)
    torch.cuda.synchronize()
    qv = np.random.default_rng(5).uniform(8, 25, int(go[-1])).astype(np.float32)
    got, got_sk = libmod.umi_cluster_groups(d_out.cpu().numpy(), mo, go, qv, n_threads=4)
    n_clustered = 0
    for g, n in enumerate(sizes):
        a, b = int(go[g]), int(go[g + 1])
        exp, exp_sk = sor.umi_cluster_group(sor.umi_matrix(ws[a:b]).reshape(-1), n, qv[a:b])
        assert (got[a:b] == exp.astype(got.dtype)).all() and (got_sk[a:b] == exp_sk).all(), g
        n_clustered += int((exp["center"] >= 0).sum())
    assert n_clustered > 400


def _name_with_window(i, w, q, bc="ACGTACGTACGTACGT"):
    """a pass-2 read name whose UMI window (3': reverse complement of X= from AE + 3 - bcEnd on) is the 14 codes of w"""
    comp = {1: "T", 2: "C", 4: "G", 8: "A", 15: "N"}     # the letter whose complement has code c
    x = ["A"] * 43
    for k in range(14):
        x[24 - k] = comp[int(w[k])]
    return f"r{i}_FWD_PS=700_PE=730_AE=743_bc={bc}_ed=0_ed_sec=3_bcStart=742_bcEnd=727_X={''.join(x)}_Q={q}_{i:x} cellBC={bc}"


@pytest.mark.parametrize("n,own_above", [(101, 100), (1000, 100), (3300, 4000), (20000, 100)])
def test_assignumis_chunk_large_groups_equal_oracle(pkg, sor, gpu_ctx, n, own_above):
    """one (cell, region) group of n reads through smi_assignumis_chunk against the oracle's clusterer on the same matrix: just above the
    switch to ClusterOne_MyClustering (101), well inside it (1000, 20000: a 400 MB matrix), and -- with the switch moved up -- the
    hierarchical path with more than 3000 neighbour reads, where the reference changes from complete link to single link"""
    from sicelore_amd import lib as libmod

    rng = np.random.default_rng(n)
    n_umi = max(3, n // 12)
    ws = _make_groups(n, [n])                              # n reads around n // 3 UMIs with up to two edits each
    if n >= 3000:                                          # fewer, deeper molecules: clusters of very different depth (the fold filter)
        base = _make_groups(n + 1, [n_umi])
        pick = np.minimum(rng.zipf(1.4, n) - 1, n_umi - 1)
        ws = base[pick].copy()
        noisy = rng.random(n) < 0.3
        ws[noisy, rng.integers(0, 14, int(noisy.sum()))] = rng.choice([1, 2, 4, 8], int(noisy.sum()))
    qs = [f"{v:.1f}".rstrip("0").rstrip(".") for v in rng.uniform(8, 25, n)]
    names = [_name_with_window(i, ws[i], qs[i]) for i in range(n)]
    flags = np.zeros(n, dtype=np.uint16)
    pos0 = (100_000 + rng.integers(0, 50, n)).astype(np.int32)
    order = np.argsort(pos0, kind="stable")               # a coordinate-sorted chunk
    names, ws, qs, pos0 = [names[i] for i in order], ws[order], [qs[i] for i in order], pos0[order]
    cigars = [np.array([1000 << 4], dtype=np.uint32)] * n
    ccfg = libmod.umi_cluster_config(own_clusterer_above=own_above)
    tags, n_done = gpu_ctx.assignumis_chunk(names, flags, pos0, cigars, n_threads=8, cluster_cfg=ccfg)
    assert n_done == n and (tags["region"] == tags["region"][0]).all() and tags["region"][0] >= 0
    # the matrix: the oracle's for the sizes it finishes in seconds, K-UMI's otherwise (checked against the oracle on sampled pairs)
    if n <= 1000:
        mat = sor.umi_matrix(ws).reshape(-1)
    else:
        mat = gpu_ctx.umi_dist_batch(_pack(ws), np.array([0, n], dtype=np.uint32))
        for a, b in rng.integers(0, n, (1500, 2)):
            a, b = int(min(a, b)), int(max(a, b))                                            # ([v][i] holds the transposed copy)
            assert int(mat[a * n + b]) == int(sor.umi_pair(ws[a], ws[b]))
    qv = np.array([float(q) for q in qs], dtype=np.float32)
    exp, exp_sk = sor.umi_cluster_group(mat, n, qv, sor.umi_cluster_params(own_above=own_above))
    dec = {1: "A", 2: "G", 4: "C", 8: "T", 15: "N"}
    n_clustered = 0
    for j in range(n):
        t = tags[j]
        if exp["center"][j] < 0:
            assert not (t["flags"] & libmod.UMI_CLUSTERED), j
            assert bool(t["flags"] & libmod.UMI_SKIPPED) == bool(exp_sk[j]), j
            continue
        n_clustered += 1
        c = int(exp["center"][j])
        assert t["flags"] & libmod.UMI_CLUSTERED and t["center"] == c and t["u1"] == exp["ed"][j] and t["u2"] == exp["ed_second"][j], j
        off = int(exp["offset"][j])
        assert t["u8"].decode() == "".join(dec[int(ws[c][k + 1 + off])] for k in range(12)), j
        assert t["u7"].decode() == "".join(dec[int(ws[j][k + 1])] for k in range(12))
    assert n_clustered > 0.5 * n
    if n == 3300:   # more than 3000 reads have a neighbour within ed 2: the reference's switch to single-link clustering was taken
        m2 = (np.asarray(mat).reshape(n, n) & 15) <= 2
        np.fill_diagonal(m2, False)
        assert int(m2.any(axis=1).sum()) > 3000


# ---- umis/umi_length other than 12 (config.xml:264; smi_ctx_set_knobs, round 6) ---------------------------------------------------------------
@pytest.mark.parametrize("ul", [8, 9, 10, 11])
def test_umi_matrices_of_other_umi_lengths_match_oracle(pkg, sor, gpu_ctx, ul):
    """K-UMI<UL>: flat kernel (small groups) and tiled kernel (groups above 64 reads) against the oracle's matrices of umi_length-mers; the
    windows carry garbage in the nibbles above their umi_length + 2 bases (nothing may read them)"""
    from sicelore_amd import lib as libmod

    sizes = [2, 3, 1, 7, 64, 5, 130, 2, 33, 65, 257]
    ws = _make_groups(100 + ul, sizes)[:, :ul + 2]
    packed = _pack(np.concatenate([ws, np.zeros((ws.shape[0], 12 - ul), np.uint8)], axis=1))
    packed |= np.random.default_rng(ul).integers(0, 1 << 62, packed.size, dtype=np.uint64) & ~np.uint64((1 << (4 * (ul + 2))) - 1)   # garbage above the window
    go, po, mo = gpu_ctx.umi_offsets(sizes)
    d_out = torch.full((int(mo[-1]),), 255, dtype=torch.uint8, device="cuda")
    gpu_ctx.set_knobs(libmod.run_knobs(umi_length=ul))
    try:
        gpu_ctx.umi_dist_device(torch.from_numpy(packed.view(np.int64)).cuda(), torch.from_numpy(go.view(np.int32)).cuda(), torch.from_numpy(po.view(np.int64)).cuda(),
                                torch.from_numpy(mo.view(np.int64)).cuda(), len(sizes), int(po[-1]), d_out)
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_knobs(None)
    out = d_out.cpu().numpy()
    seen = set()
    for g, n in enumerate(sizes):
        exp = sor.umi_matrix(ws[go[g]:go[g + 1]], ul)
        got = out[int(mo[g]):int(mo[g + 1])].reshape(n, n)
        assert (got == exp).all(), (ul, g)
        seen |= set((exp & 15).reshape(-1).tolist())
    assert seen == {0, 1, 2, 3, 4, 5}
    # and the same windows read as 12-mers give other matrices (the length is not decoration)
    d12 = torch.full((int(mo[-1]),), 255, dtype=torch.uint8, device="cuda")
    gpu_ctx.umi_dist_device(torch.from_numpy(packed.view(np.int64)).cuda(), torch.from_numpy(go.view(np.int32)).cuda(), torch.from_numpy(po.view(np.int64)).cuda(),
                            torch.from_numpy(mo.view(np.int64)).cuda(), len(sizes), int(po[-1]), d12)
    torch.cuda.synchronize()
    assert (d12.cpu().numpy() != out).any()


def _name_with_window_len(i, w, q, ul, five, bc="ACGTACGTACGTACGT"):
    """a pass-2 read name whose UMI window holds the ul + 2 codes of w: 3' on the reverse complement of X= from AE + 3 - bcEnd on, 5' on X= from
    bcEnd - AE + 3 on"""
    dec = {1: "A", 2: "G", 4: "C", 8: "T", 15: "N"}
    comp = {1: "T", 2: "C", 4: "G", 8: "A", 15: "N"}
    if five:
        x = ["A"] * 42
        for k in range(ul + 2):
            x[18 + k] = dec[int(w[k])]                      # bcEnd - AE + 3 = 19 (1-based) with AE = 50, bcEnd = 66
        return f"r{i}_FWD_AE=50_bc={bc}_ed=0_bcStart=51_bcEnd=66_X={''.join(x)}_Q={q}_{i:x} cellBC={bc}"
    x = ["A"] * 43
    for k in range(ul + 2):
        x[24 - k] = comp[int(w[k])]
    return f"r{i}_FWD_PS=700_PE=730_AE=743_bc={bc}_ed=0_ed_sec=3_bcStart=742_bcEnd=727_X={''.join(x)}_Q={q}_{i:x} cellBC={bc}"


@pytest.mark.parametrize("ul,five,host", [(10, False, False), (10, True, False), (10, False, True), (8, False, False), (11, True, False)])
def test_assignumis_chunk_with_other_umi_lengths_equals_oracle(pkg, sor, gpu_ctx, ul, five, host, monkeypatch):
    """the whole UMI stage (K-UPARSE windows of umi_length + 2 bases, K-UMI<UL>, K-UCLUST / the own clusterer, K-UTAG's umi_length characters) for
    groups of 2 .. 300 reads of one cell against the oracle's matrices and clusterer; 5': the clustering position is the reference position under
    AE + 16 + umi_length + 100 (NanoporeRead$ReadScanData.java:L90)"""
    from sicelore_amd import lib as libmod

    if host:
        monkeypatch.setenv("SMI_AU_HOST", "1")
    rng = np.random.default_rng(1000 + ul + five)
    sizes = [3, 4, 9, 40, 100, 130, 300, 7]              # (a region needs more than two reads: ReadGrouper.java:L171-184)
    ws_all, names, pos0 = [], [], []
    for g, n in enumerate(sizes):
        ws = _make_groups(50 + g + ul, [n])[:, :ul + 2]
        qs = [f"{v:.1f}".rstrip("0").rstrip(".") for v in rng.uniform(8, 25, n)]
        offs = np.zeros(n, dtype=np.int64)                    # one position per group (ReadGrouper's off-centre rules are not the subject here)
        for i in range(n):
            names.append(_name_with_window_len(len(names), ws[i], qs[i], ul, five))
            pos0.append(100_000 + 5_000 * g + int(offs[i]))
        ws_all.append((ws, np.array([float(q) for q in qs], dtype=np.float32)))
    n = len(names)
    cigars = [np.array([1000 << 4], dtype=np.uint32)] * n
    tags, n_done = gpu_ctx.assignumis_chunk(names, np.zeros(n, np.uint16), np.array(pos0, dtype=np.int32), cigars, n_threads=4, five_prime=five, umi_length=ul)
    assert n_done == n
    dec = {1: "A", 2: "G", 4: "C", 8: "T", 15: "N"}
    ws_flat = np.concatenate([w for w, _ in ws_all])
    qv_flat = np.concatenate([q for _, q in ws_all])
    n_clustered = n_grouped = 0
    for j in range(n):
        assert tags["u7"][j].decode() == "".join(dec[int(c)] for c in ws_flat[j][1:1 + ul]), j
    # the groups as ReadGrouper cut them (one cell: a group = a region; the read that opens a new chain can stay without a region), members in input order
    for reg in sorted(set(int(r) for r in tags["region"] if r >= 0)):
        mem = np.nonzero(tags["region"] == reg)[0]
        m = mem.size
        if m < 2:
            continue
        n_grouped += m
        ws, qv = ws_flat[mem], qv_flat[mem]
        exp, exp_sk = sor.umi_cluster_group(sor.umi_matrix(ws, ul).reshape(-1), m, qv)
        for j in range(m):
            t = tags[mem[j]]
            if exp["center"][j] < 0:
                assert not (t["flags"] & libmod.UMI_CLUSTERED), (reg, j)
                continue
            n_clustered += 1
            c, off = int(exp["center"][j]), int(exp["offset"][j])
            assert t["flags"] & libmod.UMI_CLUSTERED and t["center"] == mem[c] and t["u1"] == exp["ed"][j] and t["u2"] == exp["ed_second"][j], (reg, j)
            assert t["u8"].decode() == "".join(dec[int(ws[c][k + 1 + off])] for k in range(ul)), (reg, j)
    assert n_grouped > 0.95 * n and n_clustered > 300


@pytest.mark.parametrize("sizes", [[2, 3, 1, 7, 64, 5, 130, 2, 2, 33], [400], [65, 3, 66, 71, 127, 128, 129, 64, 63, 191, 192, 193], [257, 2, 321]])
def test_padded_matrices_hold_the_dense_ones(pkg, sor, gpu_ctx, sizes):
    """smi_umi_dist_device_padded (the layout of the chunk worker's own matrices: rows of a group above 64 reads rounded up to whole 64-byte lines, groups
    on line boundaries): cell [i][v] at mat_off[g] + i * row + v equals the dense matrix's and the oracle's; nothing is written into the padding"""
    from sicelore_amd import lib as libmod

    ws = _make_groups(len(sizes) + 7, sizes)
    go, po, mo = gpu_ctx.umi_offsets(sizes, padded=True)
    L = libmod.load_library()
    for n in sizes:
        row, nbytes = int(L.smi_umi_padded_row(n)), int(L.smi_umi_padded_bytes(n))
        assert row == (n if n <= 64 else (n + 63) // 64 * 64) and nbytes == (row * n + 63) // 64 * 64
    assert all(int(x) % 64 == 0 for x in mo)
    d_out = torch.full((int(mo[-1]),), 255, dtype=torch.uint8, device="cuda")
    gpu_ctx.umi_dist_device(torch.from_numpy(_pack(ws).view(np.int64)).cuda(), torch.from_numpy(go.view(np.int32)).cuda(), torch.from_numpy(po.view(np.int64)).cuda(),
                            torch.from_numpy(mo.view(np.int64)).cuda(), len(sizes), int(po[-1]), d_out, padded=True)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    for g, n in enumerate(sizes):
        row = n if n <= 64 else (n + 63) // 64 * 64
        blk = out[int(mo[g]):int(mo[g]) + row * n].reshape(n, row)
        assert (blk[:, :n] == sor.umi_matrix(ws[go[g]:go[g + 1]])).all(), g
        assert (blk[:, n:] == 255).all() and (out[int(mo[g]) + row * n:int(mo[g + 1])] == 255).all(), g
