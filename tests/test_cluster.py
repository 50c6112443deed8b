"""UMI clustering of a (cell, region) group: oracle == independent Python model (== product, added below)."""
import numpy as np
import pytest

import pymodel_cluster as pc


def graft_pkg():
    import __graft_entry__ as graft

    return graft.load_package()


PRODUCT_KEYS = {"complete_ed": "complete_link_ed", "single_ed": "single_link_ed", "single_switch": "single_link_switch",
                "fold": "fold_depth_below_max", "own_above": "own_clusterer_above"}


def lev(a, b):
    d = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        p = d[:]
        d[0] = i
        for j, cb in enumerate(b, 1):
            d[j] = min(p[j] + 1, d[j - 1] + 1, p[j - 1] + (ca != cb))
    return d[-1]


def make_group(rng, n, n_umis, err=0.08):
    """n reads drawn from n_umis molecules with sequencing errors; packed matrix like K-UMI's (positions random but
    consistent: [v][i] is the transposed copy)"""
    umis = ["".join(rng.choice(list("ACGT"), 12)) for _ in range(n_umis)]
    reads = []
    for _ in range(n):
        u = list(umis[rng.integers(0, n_umis)])
        for k in range(12):
            if rng.random() < err:
                u[k] = "ACGT"[rng.integers(0, 4)]
        reads.append("".join(u))
    mat = np.zeros((n, n), dtype=np.uint8)
    for i in range(n):
        for j in range(i, n):
            ed = min(lev(reads[i], reads[j]), 5) if i != j else 0
            p1, p2 = (1, 1) if i == j else (int(rng.integers(0, 3)), int(rng.integers(0, 3)))
            mat[i, j] = ed | p1 << 4 | p2 << 6
            mat[j, i] = ed | p2 << 4 | p1 << 6
    return mat.reshape(-1), rng.uniform(8, 20, n).astype(np.float32)


def check(sor, mat, n, qv, **kw):
    graft_pkg()
    out, sk = sor.umi_cluster_group(mat, n, qv, sor.umi_cluster_params(**kw))
    m_out, m_sk = pc.cluster_group([int(x) for x in mat], n, [float(x) for x in qv], **kw)
    for i in range(n):
        got = None if out["center"][i] < 0 else dict(center=int(out["center"][i]), offset=int(out["offset"][i]),
                                                      ed=int(out["ed"][i]), pos2=int(out["pos2"][i]),
                                                      ed_second=None if out["ed_second"][i] < 0 else int(out["ed_second"][i]))
        assert got == m_out[i], (i, got, m_out[i])
    assert list(sk) == m_sk
    # product (host C++ in libsicelore_mi.so) on the same group
    from sicelore_amd import lib as libmod

    cfg = libmod.umi_cluster_config(**{PRODUCT_KEYS[k]: v for k, v in kw.items()})
    p_out, p_sk = libmod.umi_cluster_groups(mat, [0, n * n], [0, n], qv, cfg)
    for f in ("center", "offset", "ed", "ed_second", "pos2"):
        assert (p_out[f] == out[f]).all(), f
    assert (p_sk == sk).all()
    return out, sk


def test_fastutil_iteration_order_known_values():
    # hand-computed from mix(k) = (k * 0x9E3779B9) ^ (>>> 16), table 32, top-down iteration, 0 first
    assert pc.fastutil_order([0, 1, 2, 3]) == [0] + sorted([1, 2, 3], key=lambda k: -(pc.mix(k) & 31))
    big = pc.fastutil_order(range(1, 60))  # forces two rehashes (24 -> 64, 48 -> 128)
    assert sorted(big) == list(range(1, 60))


def test_hierarchical_small_cases(sor):
    rng = np.random.default_rng(3)
    # two tight molecules + one loner
    n = 7
    ed = np.full((n, n), 5, dtype=np.uint8)
    for grp in ([0, 2, 5], [1, 3, 4]):
        for a in grp:
            for b in grp:
                ed[a, b] = 0 if a == b else 1
    np.fill_diagonal(ed, 0)
    mat = (ed | (1 << 4) | (1 << 6)).reshape(-1)
    out, sk = check(sor, mat, n, rng.uniform(8, 20, n).astype(np.float32))
    assert set(out["center"][[0, 2, 5]]) <= {0, 2, 5} and len(set(out["center"][[0, 2, 5]])) == 1
    assert out["center"][6] == -1 and not sk.any()
    assert (out["ed_second"][:6] == 5).all()


def test_random_groups_oracle_equals_model(sor):
    rng = np.random.default_rng(11)
    n_clustered = 0
    for trial in range(60):
        n = int(rng.integers(2, 40))
        mat, qv = make_group(rng, n, max(1, n // int(rng.integers(2, 6))), err=float(rng.choice([0.03, 0.08, 0.15])))
        out, _ = check(sor, mat, n, qv)
        n_clustered += int((out["center"] >= 0).sum())
    assert n_clustered > 300


def test_fold_depth_filter(sor):
    rng = np.random.default_rng(12)
    n = 90
    ed = np.full((n, n), 5, dtype=np.uint8)
    ed[:80, :80] = 1  # one molecule with 80 reads
    ed[80:82, 80:82] = 1  # a 2-read molecule: 2 * 50 = 100 > 80 -> kept
    ed[84:85, 84:85] = 0
    np.fill_diagonal(ed, 0)
    mat = (ed | (1 << 4) | (1 << 6)).reshape(-1)
    out, sk = check(sor, mat, n, rng.uniform(8, 20, n).astype(np.float32), fold=30)  # 2 * 30 = 60 <= 80 -> skipped
    assert sk[80] and sk[81] and out["center"][80] == -1
    out, sk = check(sor, mat, n, rng.uniform(8, 20, n).astype(np.float32))
    assert not sk.any() and out["center"][80] in (80, 81)


def test_own_clusterer_path(sor):
    rng = np.random.default_rng(13)
    n_clustered = n_groups = 0
    for trial in range(40):
        n = int(rng.integers(6, 60))
        mat, qv = make_group(rng, n, max(1, n // int(rng.integers(2, 8))), err=float(rng.choice([0.05, 0.12, 0.2])))
        out, _ = check(sor, mat, n, qv, own_above=5)
        n_clustered += int((out["center"] >= 0).sum())
        n_groups += 1
    assert n_clustered > 300
    # above the real threshold
    mat, qv = make_group(rng, 130, 20, err=0.08)
    check(sor, mat, 130, qv)


def test_batched_groups_threads(sor):
    from sicelore_amd import lib as libmod

    graft_pkg()
    rng = np.random.default_rng(21)
    mats, qvs, sizes = [], [], []
    for _ in range(50):
        n = int(rng.integers(1, 50))
        m, q = make_group(rng, n, max(1, n // 3))
        mats.append(m)
        qvs.append(q)
        sizes.append(n)
    sizes = np.array(sizes)
    group_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    mat_off = np.concatenate([[0], np.cumsum(sizes.astype(np.uint64) ** 2)]).astype(np.uint64)
    dist, qv = np.concatenate(mats), np.concatenate(qvs)
    a1, s1 = libmod.umi_cluster_groups(dist, mat_off, group_off, qv, n_threads=1)
    a4, s4 = libmod.umi_cluster_groups(dist, mat_off, group_off, qv, n_threads=4)
    assert (a1 == a4).all() and (s1 == s4).all()
    for g in range(len(sizes)):
        o, _ = sor.umi_cluster_group(mats[g], int(sizes[g]), qvs[g])
        assert (a1[group_off[g]:group_off[g + 1]] == o.astype(a1.dtype)).all()
