"""The UMI stage of assignumis on the device (smi_umi_stage.hip) against the host path it replaces (which tests/test_pipeline_gpu.py,
tests/test_umi_gpu.py and tests/test_ref_exec.py hold to the oracle and to the reference's bytecode): K-UCLUST against the oracle's
clusterer and against the reference-executed groups, whole chunks through smi_assignumis_chunk in both forms."""
import importlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _device_cluster(pkg, ctx, mats, qvs, sizes, cfg=None):
    from sicelore_amd import lib as libmod

    sizes = np.asarray(sizes, dtype=np.int64)
    go = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    mo = np.concatenate([[0], np.cumsum(sizes * sizes)]).astype(np.uint64)
    dev = torch.device("cuda", ctx.device)
    d_dist = torch.from_numpy(np.concatenate(mats).astype(np.uint8)).to(dev)
    d_qv = torch.from_numpy(np.concatenate(qvs).astype(np.float32)).to(dev)
    d_go = torch.from_numpy(go.view(np.int32)).to(dev)
    d_mo = torch.from_numpy(mo.view(np.int64)).to(dev)
    n = int(go[-1])
    d_out = torch.zeros(n, dtype=torch.int64, device=dev)
    d_sk = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.umi_cluster_groups_device(d_dist, d_mo, d_go, len(sizes), d_qv, d_out, d_sk, cfg=cfg)
    torch.cuda.synchronize()
    return d_out.cpu().numpy().view(libmod.UMI_ASSIGNMENT_DTYPE), d_sk.cpu().numpy().astype(bool), go


def test_k_uclust_equals_oracle_on_random_groups(pkg, sor, gpu_ctx):
    """3,000 groups of 2 .. 100 reads (ties everywhere: integer distances 0 .. 5), shipped knobs and a tighter fold filter"""
    import test_cluster as tc

    rng = np.random.default_rng(31)
    mats, qvs, sizes = [], [], []
    for trial in range(3000):
        n = int(rng.choice([2, 3, 4, 5, 6, 8, 12, 20, 33, 63, 64, 65, 99, 100])) if trial % 3 else int(rng.integers(2, 101))
        if n > 40 and trial % 7:
            n = int(rng.integers(2, 30))          # the expensive Python matrix builder: few large groups
        m, q = tc.make_group(rng, n, max(1, n // int(rng.integers(2, 6))), err=float(rng.choice([0.03, 0.08, 0.15, 0.3])))
        mats.append(m)
        qvs.append(q)
        sizes.append(n)
    from sicelore_amd import lib as libmod

    for kw, okw in ((dict(), dict()), (dict(fold_depth_below_max=3), dict(fold=3)), (dict(complete_link_ed=1), dict(complete_ed=1))):
        got, got_sk, go = _device_cluster(pkg, gpu_ctx, mats, qvs, sizes, cfg=libmod.umi_cluster_config(**kw))
        n_clustered = n_skipped = 0
        for g, n in enumerate(sizes):
            exp, exp_sk = sor.umi_cluster_group(mats[g], n, qvs[g], sor.umi_cluster_params(**okw))
            a, b = int(go[g]), int(go[g + 1])
            assert (got[a:b] == exp.astype(got.dtype)).all(), (g, n, got[a:b], exp)
            assert (got_sk[a:b] == exp_sk).all(), (g, n)
            n_clustered += int((exp["center"] >= 0).sum())
            n_skipped += int(exp_sk.sum())
        assert n_clustered > 10_000
        if kw.get("fold_depth_below_max"):
            assert n_skipped > 50


def test_k_uclust_equals_reference_bytecode(pkg, sor, gpu_ctx):
    """the groups of tests/golden/ref_exec_cluster.json (ClusterOneHierarchical.call executed from the reference's class files) through
    the device clusterer: equal to the reference where its answer does not depend on a hash order, equal to the host path everywhere"""
    from sicelore_amd import lib as libmod
    from test_ref_exec import _cluster_tags

    assignumis = importlib.import_module("sicelore_amd.assignumis")
    with open(os.path.join(GOLD, "ref_exec_cluster.json")) as f:
        sec = json.load(f)["sections"][0]

    def device(m, n, q):
        got, sk, _ = _device_cluster(pkg, gpu_ctx, [m], [q], [n])
        return got, sk

    host = lambda m, n, q: libmod.umi_cluster_groups(m, [0, n * n], [0, n], q)  # noqa: E731
    n_ref = 0
    for c in sec["cases"]:
        got = _cluster_tags(sor, assignumis.scan_data_from_name, c["names"], device)
        assert got == _cluster_tags(sor, assignumis.scan_data_from_name, c["names"], host), c["names"]
        if c["hash_orders_agree"]:
            assert got == c["set_attribute"], c["names"]
            n_ref += 1
    assert n_ref >= 25


def _chunk(pkg, synth, ctx, n_mol, copies, genes, seed, five=False):
    """a BamReader chunk as tools/microbench.py builds it: names from a real pass 2, every molecule read `copies` times"""
    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    rng = np.random.default_rng(seed)
    wl = synth.make_whitelist(50_000, seed=seed)
    used = synth.pick_used(wl, 40, seed=seed + 1)
    ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    if five:
        mol = synth.gen_reads_5p(n_mol, used, seed=seed + 2, err=0.0)
    else:
        mol = synth.gen_reads(n_mol, used, seed=seed + 2, err=0.0, q_mean=20.0)
    seqs, quals = zip(*(synth.materialize(mol, i) for i in range(n_mol)))
    text = "".join(f"@m{i} ch=1\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(zip(seqs, quals))).encode()
    rs = scanfastq.ReadScanner(ctx, max_ed=1, split_chimeras=False, five_prime=five, dont_search_polya=five)
    recs = [r for r in rs.pass2_chunk(text) if "_FAILED" not in r["name"] and "bc=" in r["name"]]
    gene = rng.integers(0, genes, len(recs))
    rows = []
    for m, r in enumerate(recs):
        q = r["name"].split(" ")[0]
        head, x_rest = q.split("_X=")
        x, rest = x_rest.split("_", 1)
        for c in range(copies if m % 11 else 4 * copies):       # some deep molecules
            xs = list(x)
            if rng.random() < 0.3:
                xs[int(rng.integers(0, len(xs)))] = "ACGT"[int(rng.integers(0, 4))]
            nm = f"{head.replace('m', 'r%d_' % c, 1)}_X={''.join(xs)}_{rest}"
            rows.append((int(gene[m]) * 5_000 + int(rng.integers(-100, 100)), nm, 16 if gene[m] & 1 else 0, r["length"]))
    # a few records the parser must pass over: no scan data at all, unmapped, a name whose barcode was cut off by -b
    rows += [(int(rng.integers(0, genes)) * 5_000, f"plain{i}", 0, 500) for i in range(20)]
    rows.sort(key=lambda t: t[0])
    names = [t[1] for t in rows]
    flags = np.array([t[2] for t in rows], dtype=np.uint16)
    flags[::97] |= 4
    pos0 = np.array([max(t[0], 0) + 1_000_000 for t in rows], dtype=np.int32)
    cigars = [np.array([(30 << 4) | 4, (t[3] - 60) << 4, (20 << 4) | 2, (30 << 4) | 0], dtype=np.uint32) for t in rows]   # 30S (L-60)M 20D 30M
    return names, flags, pos0, cigars


def _both_paths(ctx, names, flags, pos0, cigars, **kw):
    os.environ["SMI_AU_HOST"] = "1"
    try:
        host = ctx.assignumis_chunk(names, flags, pos0, cigars, **kw)
    finally:
        del os.environ["SMI_AU_HOST"]
    dev = ctx.assignumis_chunk(names, flags, pos0, cigars, **kw)
    assert dev[1] == host[1]
    assert dev[0].tobytes() == host[0].tobytes()
    return dev


@pytest.mark.parametrize("five", [False, True])
def test_chunk_device_equals_host_path(pkg, synth, gpu_ctx, five):
    from sicelore_amd import lib as libmod

    names, flags, pos0, cigars = _chunk(pkg, synth, gpu_ctx, 1500, 5, 150, 41 + five, five)
    assert len(names) > 5000
    tags, n_done = _both_paths(gpu_ctx, names, flags, pos0, cigars, five_prime=five, n_threads=4)
    assert n_done == len(names) and int((tags["flags"] & libmod.UMI_CLUSTERED != 0).sum()) > 0.5 * len(names)
    tags2, n_done2 = _both_paths(gpu_ctx, names, flags, pos0, cigars, five_prime=five, keep_data_end=True, n_threads=4)
    assert n_done2 < len(names)
    _both_paths(gpu_ctx, names, flags, pos0, cigars, five_prime=five, bc_edit_limit=0)
    _both_paths(gpu_ctx, names[:1], flags[:1], pos0[:1], cigars[:1], five_prime=five)


def test_chunk_with_unusual_names_takes_the_host_path(pkg, synth, gpu_ctx):
    """barcodes that are not 16 letters of ACGT, odd number formats: same tags as the host path (which then IS the path); a name without
    AE= fails as before"""
    names, flags, pos0, cigars = _chunk(pkg, synth, gpu_ctx, 300, 4, 20, 61)
    odd = list(names)
    k = next(i for i, nm in enumerate(odd) if "_bc=" in nm)
    odd[k] = odd[k].replace("_bc=", "_bc=N", 1)                      # 17 characters
    _both_paths(gpu_ctx, odd, flags, pos0, cigars)
    odd = list(names)
    odd[k] = odd[k].replace("_Q=", "_Q=1e1", 1)
    _both_paths(gpu_ctx, odd, flags, pos0, cigars)
    bad = list(names)
    bad[k] = bad[k].replace("_AE=", "_AF=", 1)
    with pytest.raises(pkg.SmiError, match="AE="):
        gpu_ctx.assignumis_chunk(bad, flags, pos0, cigars)
