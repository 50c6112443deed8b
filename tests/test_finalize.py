"""Pass-1 finalize (host step) and its multi-rank form: product (libsicelore_mi host code) == oracle == Python model;
world_size-2 gloo run gives every rank the single-process answer."""
import os
import random

import numpy as np
import pytest

import pymodel as pm
import __graft_entry__ as graft


class JHashMap:
    """java.util.HashMap<Long, V> with real buckets (insertion-ordered bins, resize split)"""

    def __init__(self):
        self.tab = [[] for _ in range(16)]
        self.size = 0

    @staticmethod
    def _h(k):
        h = (k ^ (k >> 32)) & 0xFFFFFFFF
        return h ^ (h >> 16)

    def put(self, k, v):
        h = self._h(k)
        b = self.tab[h & (len(self.tab) - 1)]
        for e in b:
            if e[1] == k:
                e[2] = v
                return
        assert len(b) < 8, "tree bins not modelled"
        b.append([h, k, v])
        self.size += 1
        if self.size > len(self.tab) * 3 // 4:
            oc = len(self.tab)
            new = [[] for _ in range(2 * oc)]
            for j, bb in enumerate(self.tab):
                for e in bb:
                    (new[j] if (e[0] & oc) == 0 else new[j + oc]).append(e)
            self.tab = new

    def items(self):
        for b in self.tab:
            for e in b:
                yield e[1], e[2]


def model_finalize(keys, counts, record_count, merge_ed, fold=10, below=500, details=None):
    f32 = np.float32
    cutoff = f32(f32(2.0) * f32(record_count)) / f32(5000000.0)
    cnt = {int(k): int(c) for k, c in zip(keys, counts) if f32(c) > cutoff and c > 1}
    if not cnt:
        return [], [], []
    kset = set(cnt)
    coll = {}
    for k in sorted(cnt):
        ms, _ = pm.bc_match(kset, k, 16, merge_ed, True, True, None, 0, False)
        if ms:
            coll[k] = ms
    order = sorted(coll, key=lambda k: (-cnt[k], k))  # canonical tie: ascending key
    to_merge = JHashMap()
    for k in order:
        cut = cnt[k] // fold
        to_merge.put(k, {m["matching_bc"] for m in coll[k] if m["ed"] <= merge_ed and cnt[m["matching_bc"]] < cut})
    alive = dict(cnt)
    for k, s in to_merge.items():
        if k in alive:
            for x in s:
                alive.pop(x, None)
    mn = max(alive.values()) // below
    fin = sorted(((k, c) for k, c in alive.items() if c >= mn), key=lambda kc: (-kc[1], kc[0]))
    if details is not None:
        details.update(cnt=cnt, coll=coll)
    return [k for k, _ in fin], [c for _, c in fin], list(range(1, len(fin) + 1))


def make_case(seed, n_cells=120, deep=False):
    rng = random.Random(seed)
    keys, counts = [], []
    base = [rng.getrandbits(32) for _ in range(n_cells)]
    for b in base:
        c = int(rng.lognormvariate(6, 1.5)) + 2
        keys.append(b)
        counts.append(c)
        # error-derived neighbours at ed 1 / 2 with far fewer reads, sometimes chained (A -> B -> C)
        cur, cc = b, c
        for _ in range(rng.choice([0, 1, 1, 2, 3])):
            nb = cur ^ (rng.randrange(1, 4) << (2 * rng.randrange(16)))
            if rng.random() < 0.4:
                nb ^= (rng.randrange(1, 4) << (2 * rng.randrange(16)))
            if rng.random() < 0.3:  # an indel-type neighbour: drop a base, append one
                p = rng.randrange(15)
                hi = cur >> (2 * (16 - p)) << (2 * (16 - p))
                lo = (cur & ((1 << (2 * (15 - p))) - 1)) << 2
                nb = (hi | lo | rng.randrange(4)) & 0xFFFFFFFF
            cc = max(1, cc // rng.choice([3, 8, 15, 40, 200]))
            keys.append(nb)
            counts.append(cc)
            if deep:
                cur = nb
    # equal counts and sub-threshold entries
    for _ in range(20):
        keys.append(rng.getrandbits(32))
        counts.append(rng.choice([1, 2, 2, 3, 7, 7]))
    # de-duplicate keys (keep the first)
    seen, k2, c2 = set(), [], []
    for k, c in zip(keys, counts):
        if k not in seen:
            seen.add(k)
            k2.append(k)
            c2.append(c)
    return np.array(k2, dtype=np.uint64), np.array(c2, dtype=np.uint32)


@pytest.mark.parametrize("merge_ed", [1, 2])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_finalize_product_oracle_model_agree(pkg, sor, merge_ed, seed):
    from sicelore_amd import lib as libmod

    keys, counts = make_case(seed, n_cells=60 if merge_ed == 2 else 150, deep=seed == 3)
    rec = 40
    pk, pc, pr = libmod.finalize_used_list(keys, counts, rec, merge_ed)
    ok, oc, orank = sor.finalize_used_list(keys.astype(np.int64), counts, rec, merge_ed)
    assert pk.tolist() == ok.astype(np.uint64).tolist() and pc.tolist() == oc.tolist() and pr.tolist() == orank.tolist()
    mk, mc, mr = model_finalize(keys, counts, rec, merge_ed)
    assert pk.tolist() == mk and pc.tolist() == mc and pr.tolist() == mr
    # something was merged away and something was kept
    assert 10 < pk.size < keys.size


def test_finalize_edge_cases(pkg, sor):
    from sicelore_amd import lib as libmod

    k, c, r = libmod.finalize_used_list(np.zeros(0, np.uint64), np.zeros(0, np.uint32), 1, 1)
    assert k.size == 0
    # all below the count > 1 rule
    k, c, r = libmod.finalize_used_list(np.array([5, 9], np.uint64), np.array([1, 1], np.uint32), 1, 1)
    assert k.size == 0
    # the chunk-count cutoff: 2 * recordCount / 5e6 (UsedCellBCListGenerator.java:L391)
    keys = np.array([100, 200, 300], np.uint64)
    cnts = np.array([5, 50, 500], np.uint32)
    k, c, r = libmod.finalize_used_list(keys, cnts, 20_000_000, 1)  # cutoff 8.0 -> 5 is dropped
    assert k.tolist() == [300, 200] and r.tolist() == [1, 2]
    # low-depth cut: max / 500
    keys = np.array([0x11111111, 0x22222222, 0x33333333], np.uint64)
    cnts = np.array([100_000, 150, 250], np.uint32)
    k, c, r = libmod.finalize_used_list(keys, cnts, 1, 1)
    assert sorted(k.tolist()) == sorted([0x11111111, 0x33333333])


def _dist_worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    graft.load_package()
    import importlib

    dmod = importlib.import_module(graft.PKG_NAME + ".distributed")
    keys, counts = make_case(7, n_cells=200)
    order = np.argsort(keys)
    skeys, scounts = keys[order], counts[order].astype(np.int64)
    # split every counter between the ranks (reads are sharded, so each rank sees part of every cell)
    rng = np.random.default_rng(100)
    part0 = rng.binomial(scounts, 0.5)
    mine = part0 if rank == 0 else scounts - part0
    lo, hi = dmod.shard_range(1001, rank, world)
    hist = torch.from_numpy(mine.astype(np.int32))
    k, c, r = dmod.pass1_finalize(hist, skeys, record_count=hi - lo, merge_ed=1)
    np.save(os.path.join(tmp, f"k{rank}.npy"), k)
    np.save(os.path.join(tmp, f"c{rank}.npy"), c)
    np.save(os.path.join(tmp, f"r{rank}.npy"), r)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_histogram_allreduce_and_finalize(pkg, tmp_path):
    """world_size 2 over gloo: all-reduce of the pass-1 histogram + finalize + broadcast == single-process result"""
    import torch.multiprocessing as mp
    from sicelore_amd import lib as libmod

    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_dist_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    keys, counts = make_case(7, n_cells=200)
    ek, ec, er = libmod.finalize_used_list(keys, counts, 1001, 1)
    for rank in (0, 1):
        k = np.load(tmp_path / f"k{rank}.npy")
        c = np.load(tmp_path / f"c{rank}.npy")
        r = np.load(tmp_path / f"r{rank}.npy")
        assert k.tolist() == ek.tolist() and c.tolist() == ec.tolist() and r.tolist() == er.tolist()
    assert ek.size > 50


def _model_assigned_tsv(keys, counts, max_ed):
    """ParseStatsHtmlPrinter.writeAssignedTSV (L294-327) in Python; equal counts in ascending key order (canonical)"""
    rows = [(int(c.sum()), int(k), c) for k, c in zip(keys, counts) if c.sum()]
    rows.sort(key=lambda t: (-t[0], t[1]))
    out = ["Barcode\tn Reads with ED<=%d match" % max_ed + "".join("\tED=%d" % e for e in range(max_ed + 1))]
    for tot, k, c in rows:
        bc = "".join("AGCT"[(k >> (2 * (15 - j))) & 3] for j in range(16))
        out.append(bc + "\t" + f"{tot:,}" + "".join("\t" + (f"{int(c[e]):,}" if c[e] else "0") for e in range(max_ed + 1)))
    return "\n".join(out) + "\n"


def test_assigned_tsv_equals_model(pkg):
    from sicelore_amd import lib as libmod

    rng = np.random.default_rng(3)
    keys = np.sort(rng.choice(2 ** 32, 400, replace=False).astype(np.uint64))
    counts = np.zeros((400, 3), dtype=np.uint32)
    hot = rng.choice(400, 150, replace=False)
    counts[hot, 0] = rng.integers(0, 3_000_000, 150)
    counts[hot, 1] = rng.integers(0, 2000, 150)
    counts[hot[:40], 2] = rng.integers(0, 30, 40)
    counts[hot[5]] = counts[hot[6]]  # equal totals: ascending key
    for max_ed in (0, 1, 2):
        c = counts.copy()
        c[:, max_ed + 1:] = 0
        assert libmod.assigned_tsv(keys, c, max_ed) == _model_assigned_tsv(keys, c, max_ed)
    assert libmod.assigned_tsv(keys[:0], counts[:0], 1) == "Barcode\tn Reads with ED<=1 match\tED=0\tED=1\n"
    assert "\t1,234,567\t" in libmod.assigned_tsv(keys[:1], np.array([[1234567, 0, 0]], dtype=np.uint32), 1)


def _counts_worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    graft.load_package()
    import importlib

    dmod = importlib.import_module(graft.PKG_NAME + ".distributed")
    rng = np.random.default_rng(11)
    keys = np.sort(rng.choice(2 ** 32, 300, replace=False).astype(np.uint64))
    total = rng.integers(0, 5000, (300, 3)).astype(np.int64)
    part0 = np.random.default_rng(12).binomial(total, 0.4)
    mine = part0 if rank == 0 else total - part0
    text = dmod.assigned_counts_tsv(torch.from_numpy(mine.astype(np.int32)), keys, max_ed=2)
    open(os.path.join(tmp, f"tsv{rank}.txt"), "w").write(text)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_assigned_counters(pkg, tmp_path):
    """world_size 2 over gloo: the pass-2 counters summed over the ranks give the single-process BarcodesAssigned.tsv"""
    import torch.multiprocessing as mp

    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_counts_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(11)
    keys = np.sort(rng.choice(2 ** 32, 300, replace=False).astype(np.uint64))
    total = rng.integers(0, 5000, (300, 3)).astype(np.uint32)
    exp = _model_assigned_tsv(keys, total, 2)
    assert open(tmp_path / "tsv0.txt").read() == exp == open(tmp_path / "tsv1.txt").read()


def _bc_str(k):
    return "".join("AGCT"[(k >> (2 * (15 - i))) & 3] for i in range(16))


def model_barcode_list_tsv(keys, counts, record_count, merge_ed, no_whitelist=False):
    """ParseStatsHtmlPrinter.writesedBarcodesListTSV (L235-285) on top of model_finalize"""
    det = {}
    fk, fc, _ = model_finalize(keys, counts, record_count, merge_ed, details=det)
    if not det:
        return "Barcode\tn Reads with full match\t\n"
    cnt, coll = det["cnt"], det["coll"]
    eds = sorted({m["ed"] for ms in coll.values() for m in ms})                      # TreeMap over the distances that occur
    shown = [(k, c) for k, c in zip(fk, fc) if not (no_whitelist and ("TTTTT" in _bc_str(k) or "AAAAA" in _bc_str(k)))]
    used = {k: c for k, c in shown}
    out = ["Barcode\tn Reads with full match\t" + "\t".join(f"BCs colliding at ED {e}" for e in eds) + "\n"]
    for k, c in shown:
        row = f"{_bc_str(k)}\t{c}"
        for e in eds:
            row += "\t" + ",".join(f"{_bc_str(m['matching_bc'])}({used[m['matching_bc']]} x)" if m["matching_bc"] in used
                                    else f"{_bc_str(m['matching_bc'])}({cnt[m['matching_bc']]} m)" for m in coll.get(k, []) if m["ed"] == e)
        out.append(row + "\n")
    return "".join(out)


@pytest.mark.parametrize("merge_ed", [1, 2])
def test_barcode_list_tsv_equals_model(pkg, merge_ed):
    from sicelore_amd import lib as libmod

    for seed in (3, 4, 5):
        keys, counts = make_case(seed, n_cells=90, deep=seed == 5)
        for no_wl in (False, True):
            got = libmod.barcode_list_tsv(keys, counts, 40, merge_ed=merge_ed, no_whitelist=no_wl)
            exp = model_barcode_list_tsv(keys, counts, 40, merge_ed, no_whitelist=no_wl)
            assert got == exp
            rows = got.splitlines()
            assert rows[0].startswith("Barcode\tn Reads with full match\t") and len(rows) > 40
    assert " x)" in got or " m)" in got
