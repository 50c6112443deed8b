"""End to end: FASTQ text -> scanfastq pass 2 on the device (K-FQ, K-PACKR, K-CHIM, K-PACK, K-SCAN, K-BC1, name writer)
== the oracle run record by record; and pass 1 -> used list -> pass 2 with ranks."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
COMP = bytes.maketrans(b"ACGTN", b"TGCAN")


def _fastq(seqs, quals):
    return "".join(f"@read{i} runid=x ch={i % 9}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(zip(seqs, quals))).encode()


def _oracle_pass2(sor, bset, seqs, quals, max_ed, rank_of):
    out = []
    rid = 0
    for i, (s, q) in enumerate(zip(seqs, quals)):
        name = f"read{i} runid=x ch={i % 9}"
        rc, splits, multi, _, raw = sor.chimera_split(s)
        assert rc == 0
        cuts = [0] + [p for _, p in splits] + [len(s)]
        for k in range(len(cuts) - 1):
            fs, fq = s[cuts[k]:cuts[k + 1]], q[cuts[k]:cuts[k + 1]]
            fname = sor.chimera_fragment_name(name, raw, k) if splits else name
            if multi:
                out.append(fname.split(" ")[0] + "_FAILED ")
                rid += 1
                continue
            rc, sc = sor.scan_read_3p(fs, fq, "CTTCCGATCT")
            assert rc == 0
            a = None
            if sc["adapter_found"]:
                stranded = fs.encode().translate(COMP)[::-1] if sc["reverse"] else fs.encode()
                rc2, a_ = sor.assign_barcode(bset, stranded, int(sc["adapter_end"]), max_ed=max_ed)
                if rc2 == 1:
                    a = a_
            rk = rank_of.get(int(a["bc"]) & 0xFFFFFFFF, 0) if a is not None else 0
            out.append(sor.format_read_name(fname, fs, fq, sc, a, rank=rk, read_id=rid))
            rid += 1
    return out


def test_pass2_from_fastq_text_equals_oracle(pkg, synth, sor, gpu_ctx):
    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    wl = synth.make_whitelist(20_000, seed=251)
    used = synth.pick_used(wl, 200, seed=252)
    reads = synth.gen_reads(300, used, seed=253, n_rate=0.002)
    chim = synth.make_chimeras(reads, 400, seed=254)
    seqs = [c[0] for c in chim] + ["ACGT" * 30]  # one too-short read
    quals = [c[1] for c in chim] + ["5" * 120]
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    rank_of = {int(k): i + 1 for i, k in enumerate(used.numpy())}
    rs = scanfastq.ReadScanner(gpu_ctx, max_ed=1)
    got = rs.pass2_chunk(_fastq(seqs, quals), rank_of=rank_of)
    exp = _oracle_pass2(sor, sor.BarcodeSet(used.numpy()), seqs, quals, 1, rank_of)
    assert [g["name"] for g in got] == exp
    assert len(got) > len(seqs) and sum(g["passed"] for g in got) > 300
    assert sum("_FAILED " in g["name"] for g in got) > 10 and sum(" cellBC=" in g["name"] for g in got) > 250


def test_two_pass_flow(pkg, synth, sor, gpu_ctx):
    """pass 1 on the whitelist -> finalize -> pass 2 on the used list: most reads get their planted barcode and a rank"""
    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    from sicelore_amd import lib as libmod

    wl = synth.make_whitelist(100_000, seed=261)
    used = synth.pick_used(wl, 40, seed=262)
    reads = synth.gen_reads(4000, used, seed=263, q_mean=16.0)
    seqs, quals = zip(*(synth.materialize(reads, i) for i in range(4000)))
    text = _fastq(seqs, quals)
    keys = np.sort(wl.numpy().astype(np.uint64))
    gpu_ctx.set_barcode_set(keys, mode=1)
    rs = scanfastq.ReadScanner(gpu_ctx, max_ed=1)
    hist = torch.zeros(keys.size, dtype=torch.int32, device="cuda")
    assert rs.pass1_chunk(text, hist) == 4000
    h = hist.cpu().numpy()
    nz = np.nonzero(h)[0]
    k, c, r = libmod.finalize_used_list(keys[nz], h[nz].astype(np.uint32), 5000, 1, 10, 500)  # record_count as for a full run
    assert 20 <= k.size <= 45 and set(k.tolist()) <= set(used.numpy().astype(np.uint64).tolist())
    gpu_ctx.set_barcode_set(k, mode=0)
    out = rs.pass2_chunk(text, rank_of={int(kk): int(rr) for kk, rr in zip(k, r)})
    named = [o for o in out if " cellBC=" in o["name"]]
    assert len(named) > 2500 and all("_rk=" in o["name"] for o in named)
    truth = reads["truth"].numpy()
    ok = sum(sor.encode(o["name"].split(" cellBC=")[1]) == int(truth[o["source"]]) for o in named)
    assert ok > 0.98 * len(named)


def test_assignumis_flow_recovers_planted_umis(pkg, synth, sor, gpu_ctx):
    """scanfastq names -> region grouping -> K-UMI -> clustering: reads of one molecule end with one UMI; every stage is
    also run through the oracle on the same inputs"""
    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    assignumis = importlib.import_module("sicelore_amd.assignumis")
    from sicelore_amd import lib as libmod

    rng = np.random.default_rng(7)
    wl = synth.make_whitelist(50_000, seed=271)
    used = synth.pick_used(wl, 6, seed=272)
    n_mol, copies = 60, 6
    mol = synth.gen_reads(n_mol, used, seed=273, err=0.0, q_mean=20.0)          # error-free molecules ...
    seqs, quals, mol_of = [], [], []
    for m in range(n_mol):
        s, q = synth.materialize(mol, m)
        for _ in range(copies):                                                  # ... read several times with errors
            t = list(s)
            for p in rng.integers(0, len(t), max(1, len(t) // 40)):
                t[p] = "ACGT"[rng.integers(0, 4)]
            seqs.append("".join(t))
            quals.append(q)
            mol_of.append(m)
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    rs = scanfastq.ReadScanner(gpu_ctx, max_ed=1, split_chimeras=False)
    recs = rs.pass2_chunk(_fastq(seqs, quals))
    names = [r["name"] for r in recs]
    gene_of_mol = rng.integers(0, 8, n_mol)
    positions = [int(100_000 + 20_000 * gene_of_mol[m] + rng.integers(-100, 100)) for m in mol_of]
    strand = [bool(gene_of_mol[m] & 1) for m in mol_of]
    tags = assignumis.assign_umis(gpu_ctx, names, positions, strand)
    # planted UMI of a molecule (transcript sense); the tested sequence is the reverse complement of X=
    n_tagged = n_right = 0
    by_mol = {}
    for i, t in enumerate(tags):
        if t is None:
            continue
        n_tagged += 1
        by_mol.setdefault(mol_of[i], set()).add(t["U8"])
        n_right += t["U8"] == sor.decode(int(mol["umi"][mol_of[i]]), 12)
    assert n_tagged > 0.6 * len(names) and n_right > 0.9 * n_tagged
    assert sum(len(v) == 1 for v in by_mol.values()) > 0.85 * len(by_mol)
    # the same flow through the oracle
    info = [assignumis.parse_name(nm) for nm in names]
    pos = [p if info[i] is not None else None for i, p in enumerate(positions)]
    region, _ = sor.region_group(pos, strand)
    assert region == libmod.region_group(pos, strand)[0]
    groups = {}
    for i, f in enumerate(info):
        if f is None or region[i] < 0:
            continue
        w = sor.umi_window_3p(f["x"], f["ae"], f["bc_end"])
        mine = assignumis.umi_window(f["x"], f["ae"], f["bc_end"])
        assert (w is None) == (mine is None)
        if w is None:
            continue
        assert list(w) == mine
        groups.setdefault((f["cell"], region[i]), []).append((i, w))
    for g in groups.values():
        if len(g) < 2:
            continue
        idx = [i for i, _ in g]
        ws = np.array([w for _, w in g], dtype=np.uint8)
        asg, _ = sor.umi_cluster_group(sor.umi_matrix(ws).reshape(-1), len(g), np.array([info[i]["q"] for i in idx], np.float32))
        for j, i in enumerate(idx):
            if asg["center"][j] < 0:
                assert tags[i] is None
            else:
                assert tags[i]["center"] == idx[int(asg["center"][j])] and tags[i]["U1"] == asg["ed"][j]
