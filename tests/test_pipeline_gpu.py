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
    pos = [p if ("_REV_" in names[i] or "_FWD_" in names[i]) else None for i, p in enumerate(positions)]  # generateReadScanData
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


def test_assignumis_from_bam_chunks_equal_oracle(pkg, synth, sor, gpu_ctx):
    """BGZF BAM -> host BAM index -> BamReader.run chunking (chromosome ends, chunk size, carried tail) -> region grouping
    -> K-UMI -> clustering == the same flow with the oracle's region grouping / windows / distances / clustering"""
    import bammodel

    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    assignumis = importlib.import_module("sicelore_amd.assignumis")

    rng = np.random.default_rng(17)
    wl = synth.make_whitelist(50_000, seed=281)
    used = synth.pick_used(wl, 5, seed=282)
    n_mol, copies = 90, 5
    mol = synth.gen_reads(n_mol, used, seed=283, err=0.0, q_mean=20.0)
    seqs, quals, mol_of = [], [], []
    for m in range(n_mol):
        s, q = synth.materialize(mol, m)
        for _ in range(copies):
            t = list(s)
            for p in rng.integers(0, len(t), max(1, len(t) // 40)):
                t[p] = "ACGT"[rng.integers(0, 4)]
            seqs.append("".join(t))
            quals.append(q)
            mol_of.append(m)
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    rs = scanfastq.ReadScanner(gpu_ctx, max_ed=1, split_chimeras=False)
    recs = rs.pass2_chunk(_fastq(seqs, quals))
    # alignments: two chromosomes, genes 2.5 kb apart (closer than the 3 x 500 guard, so chunk cuts carry reads over)
    gene = rng.integers(0, 12, n_mol)
    rows = []
    for i, r in enumerate(recs):
        qname = r["name"].split(" ")[0]  # an aligner keeps the first token of the FASTQ name
        d = assignumis.scan_data_from_name(qname) if "_FAILED" not in qname else None
        m = mol_of[r["source"]]
        chrom = int(gene[m] >= 6)
        want = 20_000 + 2_500 * int(gene[m] % 6) + int(rng.integers(-120, 120))
        L = r["length"]
        if d is None:
            rows.append((chrom, want, qname, 4 if i % 2 else 0, [("M", L)] if i % 2 == 0 else [], L))
            continue
        rp = d["ps"] - 100
        lead = int(rng.integers(0, 20))  # soft clip in front: read position rp lies at reference pos0 + rp - lead
        cigar = ([("S", lead)] if lead else []) + [("M", L - lead)]
        rows.append((chrom, want - (rp - lead) + 1 - 1, qname, 16 if gene[m] & 1 else 0, cigar, L))
    rows.sort(key=lambda t: (t[0], t[1]))
    brecs = [bammodel.bam_record(nm, fl, ch if not fl & 4 else -1, p0, 30, cg, "A" * L) for ch, p0, nm, fl, cg, L in rows]
    data = bammodel.bgzf_compress(bammodel.bam_bytes("@HD\tVN:1.6\tSO:coordinate\n", [("chr1", 10 ** 6), ("chr2", 10 ** 6)], brecs), block=4096)

    names, tags = assignumis.assign_umis_bam(gpu_ctx, data, chunk_size=60)
    assert names == [t[2] for t in rows]
    # the native chunk worker (smi_assignumis_chunk: name parsing, positions, grouping, K-UMI, clustering in one call)
    names_n, tags_n = assignumis.assign_umis_bam(gpu_ctx, data, chunk_size=60, native=True)
    assert names_n == names and tags_n == tags
    assert assignumis.assign_umis_bam(gpu_ctx, data, chunk_size=10_000, native=True)[1] == assignumis.assign_umis_bam(gpu_ctx, data, chunk_size=10_000)[1]

    # ---- the oracle flow over the same chunks -------------------------------------------------------------------------------
    scans = [assignumis.scan_data_from_name(nm) if "_FAILED" not in nm else None for nm in names]
    pos = []
    for (ch, p0, nm, fl, cg, L), d in zip(rows, scans):
        pos.append(None if d is None or fl & 4 else sor.ref_position_at_read_position(cg, p0 + 1, d["ps"] - 100))
    rev = [bool(t[3] & 16) for t in rows]
    exp = [None] * len(rows)

    def flush(cur, keep):
        region, n_done = sor.region_group([pos[i] for i in cur], [rev[i] for i in cur], keep_data_end=keep)
        done = cur[:n_done]
        groups = {}
        for k, i in enumerate(done):
            d = scans[i]
            if d is None or d["bc"] is None or region[k] < 0:
                continue
            w = sor.umi_window_3p(d["x"], d["ae"], d["bc"]["end"])
            if w is not None:
                groups.setdefault((d["bc"]["seq"], region[k]), []).append((i, w))
        for g in groups.values():
            if len(g) < 2:
                continue
            idx = [i for i, _ in g]
            asg, _ = sor.umi_cluster_group(sor.umi_matrix(np.array([w for _, w in g], dtype=np.uint8)).reshape(-1), len(g),
                                           np.array([scans[i]["q"] for i in idx], np.float32))
            for j, i in enumerate(idx):
                if asg["center"][j] >= 0:
                    exp[i] = (idx[int(asg["center"][j])], int(asg["ed"][j]))
        return cur[n_done:]

    eff_ref = [-1 if t[3] & 4 else t[0] for t in rows]  # SAMRecord.getReferenceName() of an unmapped record is "*"
    cur, counter, chrom, n_flush = [0], 1, eff_ref[0], 0
    for i in range(1, len(rows)):
        counter += 1
        is_end = eff_ref[i] != chrom
        chrom = eff_ref[i]
        if counter >= 60 or is_end:
            cur = flush(cur, keep=not is_end)
            counter, n_flush = 0, n_flush + 1
        cur.append(i)
    while cur:
        cur = flush(cur, keep=False)
    assert n_flush >= 6
    got = [None if t is None else (t["center"], t["U1"]) for t in tags]
    assert got == exp
    n_tagged = sum(t is not None for t in tags)
    assert n_tagged > 0.5 * len(rows)
    # reads of one molecule end with one UMI
    by_mol = {}
    for (ch, p0, nm, fl, cg, L), t in zip(rows, tags):
        if t is not None:
            by_mol.setdefault(nm.split("_")[0], set())
    assert len(by_mol) > 200


def test_tagged_bam_output(pkg, synth, sor, gpu_ctx):
    """BAM in -> the two output BAMs: records with a barcode only, fixed fields untouched, tag values = scan data of the name
    + the UMI of the clustering (or the read's own 12-mer with UZ), umifound = the clustered subset"""
    import bammodel
    from test_bam import _parse_aux

    import os

    scanfastq = importlib.import_module("sicelore_amd.scanfastq")
    assignumis = importlib.import_module("sicelore_amd.assignumis")
    libmod = importlib.import_module("sicelore_amd.lib")
    rng = np.random.default_rng(23)
    wl = synth.make_whitelist(50_000, seed=291)
    used = synth.pick_used(wl, 4, seed=292)
    n_mol, copies = 40, 4
    mol = synth.gen_reads(n_mol, used, seed=293, err=0.0, q_mean=20.0)
    seqs, quals, mol_of = [], [], []
    for m in range(n_mol):
        s, q = synth.materialize(mol, m)
        for _ in range(copies):
            t = list(s)
            for p in rng.integers(0, len(t), max(1, len(t) // 40)):
                t[p] = "ACGT"[rng.integers(0, 4)]
            seqs.append("".join(t))
            quals.append(q)
            mol_of.append(m)
    gpu_ctx.set_barcode_set(used.numpy().astype(np.uint64), mode=0)
    recs = scanfastq.ReadScanner(gpu_ctx, max_ed=1, split_chimeras=False).pass2_chunk(_fastq(seqs, quals))
    gene = rng.integers(0, 5, n_mol)
    rows = []
    for r in recs:
        qname = r["name"].split(" ")[0]
        d = assignumis.scan_data_from_name(qname) if "_FAILED" not in qname else None
        m = mol_of[r["source"]]
        want = 30_000 + 5_000 * int(gene[m]) + int(rng.integers(-100, 100))
        if d is None:
            rows.append((want, qname, 0, [("M", r["length"])], r["length"]))
        else:
            rows.append((want - (d["ps"] - 100), qname, 16 if gene[m] & 1 else 0, [("M", r["length"])], r["length"]))
    rows.sort(key=lambda t: t[0])
    aux_in = b"NMC\x02" + b"tpAP"
    brecs = [bammodel.bam_record(nm, fl, 0, p0, 30, cg, "C" * L, aux=aux_in) for p0, nm, fl, cg, L in rows]
    header = bammodel.bam_bytes("@HD\tVN:1.6\tSO:coordinate\n", [("chr1", 10 ** 6)], [])
    data = bammodel.bgzf_compress(header + b"".join(brecs), block=8192)

    z_bc, z_umi, names, tags = assignumis.write_tagged_bams(gpu_ctx, data)
    raw_bc, raw_umi = bammodel.bgzf_decompress(z_bc), bammodel.bgzf_decompress(z_umi)
    assert raw_bc.startswith(header) and raw_umi.startswith(header)
    _, _, out_bc = bammodel.parse_bam(raw_bc)
    _, _, out_umi = bammodel.parse_bam(raw_umi)
    by_name = {nm: i for i, nm in enumerate(names)}
    n_clustered = n_uz = 0
    assert [o["pos0"] for o in out_bc] == sorted(o["pos0"] for o in out_bc)
    for o in out_bc:
        i = by_name[o["name"]]
        d = assignumis.scan_data_from_name(o["name"])
        assert d["bc"] is not None and (o["flag"], o["cigar"], o["seq"]) == (rows[i][2], rows[i][3], "C" * rows[i][4])
        a = {}
        for t, ty, v in _parse_aux(o["aux"]):
            assert t not in a
            a[t] = (ty, v)
        assert a["NM"] == ("c", 2) and a["tp"] == ("A", b"P")   # the input's NM:C:2 comes back in the smallest type, as htsjdk writes it
        assert a["BC"] == ("Z", d["bc"]["seq"]) == a["BU"] == a["BZ"] and a["AE"][1] == d["ae"] and a["PS"][1] == d["ps"] and a["PE"][1] == d["pe"]
        assert a["BB"] == ("Z", str(d["bc"]["start"])) == a["BV"] and a["BF"] == ("Z", str(d["bc"]["end"])) == a["BE"]
        assert a["B1"][1] == d["bc"]["ed"] == a["BW"][1] and a["B2"] == ("Z", str(d["bc"]["ed_sec"])) and a["SX"] == ("Z", str(d["read_id"]))
        assert ("RE" in a) == d["reverse"] and ("TE" in a) == (d["tso"] is not None)
        t = tags[i]
        w = assignumis.umi_window(d["x"], d["ae"], d["bc"]["end"])
        own = None if w is None else "".join("AGCT"[{1: 0, 2: 1, 4: 2, 8: 3}[c]] if c != 15 else "N" for c in w[1:13])
        if t is not None and not t.get("skipped"):
            n_clustered += 1
            assert a["U8"] == ("Z", t["U8"]) and a["U7"] == ("Z", t["U7"]) == ("Z", own) and a["UC"] == ("Z", "") and a["U1"] == ("Z", str(t["U1"]))
            assert ("U2" in a) == (t["U2"] is not None) and "UZ" not in a
        elif own is not None:
            n_uz += 1
            assert a["U8"] == ("Z", own) == a["U7"] and a["UZ"] == ("Z", "") and "UC" not in a
    assert n_clustered > 60 and n_uz >= 1
    assert [o["name"] for o in out_umi] == [o["name"] for o in out_bc if tags[by_name[o["name"]]] is not None and not tags[by_name[o["name"]]].get("skipped")]
    assert sum(1 for nm in names if "_bc=" in nm) == len(out_bc)
    # -w: read names cut at the first '_' (BamWriters L431-432), everything else unchanged
    z_w, _, _, _ = assignumis.write_tagged_bams(gpu_ctx, data, truncate_read_name=True)
    _, _, out_w = bammodel.parse_bam(bammodel.bgzf_decompress(z_w))
    assert [o["name"] for o in out_w] == [o["name"].split("_")[0] for o in out_bc] and [o["aux"] for o in out_w] == [o["aux"] for o in out_bc]
    # --annotationFile: GE / GS / XF from the refFlat genes (GennameTagger): reads of molecule group k sit around 30,000 + 5,000 k
    refflat = "".join(f"G{k}\tT{k}\tchr1\t{'-' if k & 1 else '+'}\t{25_000 + 5_000 * k}\t{36_000 + 5_000 * k}\t{26_000 + 5_000 * k}\t"
                      f"{35_000 + 5_000 * k}\t1\t{25_000 + 5_000 * k},\t{36_000 + 5_000 * k},\n" for k in range(0, 5, 2))
    z_g, _, names_g, _ = assignumis.write_tagged_bams(gpu_ctx, data, refflat=refflat)
    _, _, out_g = bammodel.parse_bam(bammodel.bgzf_decompress(z_g))
    assert names_g == names and [o["name"] for o in out_g] == [o["name"] for o in out_bc]
    import genemodel

    tree, _ = genemodel.load_refflat(refflat, ["chr1"])
    n_ge = 0
    for o, o0 in zip(out_g, out_bc):
        a = {t: v for t, _ty, v in _parse_aux(o["aux"])}
        ge, gs, xf = genemodel.tag(tree, "chr1", o["flag"], o["pos0"], o["cigar"])
        assert a["XF"] == xf and a.get("GE") == ge and a.get("GS") == gs
        n_ge += ge is not None
        rest = [(t, v) for t, _ty, v in _parse_aux(o["aux"]) if t not in ("XF", "GE", "GS")]
        assert rest == [(t, v) for t, _ty, v in _parse_aux(o0["aux"])]
    assert 20 < n_ge < len(out_g)
    # <out>.genecounts.tsv / <out>.UMIdepths.tsv: the counters fed by the writer = the test's GeneCounts model over the written records
    from test_gene_counts import Model

    regions = {}
    assignumis.assign_umis_bam(gpu_ctx, data, regions=regions)
    gc = libmod.GeneCounts()
    z_c, _, _, _ = assignumis.write_tagged_bams(gpu_ctx, data, refflat=refflat, gene_counts=gc)
    assert z_c == z_g
    model, mrecs = Model(), []
    for o in out_g:
        a = {t: v for t, _ty, v in _parse_aux(o["aux"])}
        ge = a.get("GE")
        mrecs.append({"gene": None if ge is None else ge.split(",")[0], "region": regions.get(by_name[o["name"]]), "bc": a.get("BC"), "u8": a.get("U8"),
                      "flag": o["flag"], "mapq": 30, "cigar": "".join(f"{ln}{op}" for op, ln in o["cigar"]), "nth": 0})
    model.add(mrecs, False)
    assert (gc.genecounts_tsv(16), gc.umi_depths_tsv()) == model.texts(libmod.two_bit_code)
    info = gc.info()
    assert info["records_with_gene"] == model.with_gene > 20 and info["region_entries"] == len(model.regions) >= n_mol // 2
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "in.bam"), "wb") as f:
            f.write(data)
        res = assignumis.assignumis_files(gpu_ctx, os.path.join(td, "in.bam"), os.path.join(td, "out"), refflat=refflat)
        assert bammodel.bgzf_decompress(open(os.path.join(td, "out.bam"), "rb").read()) == bammodel.bgzf_decompress(z_g)
        assert open(os.path.join(td, "out.genecounts.tsv")).read() == gc.genecounts_tsv(16)
        assert open(os.path.join(td, "out.UMIdepths.tsv")).read() == gc.umi_depths_tsv()
        assert os.path.getsize(os.path.join(td, "out_umifound_.bam")) > 100 and res["records"] == len(names)
    # the native pipeline (no per-record Python: smi_bam_chunk_inputs -> smi_assignumis_chunk -> smi_bam_write_batch): the same streams and tables
    for kw in (dict(), dict(refflat=refflat), dict(truncate_read_name=True), dict(chunk_size=37)):
        gc_n, gc_p = libmod.GeneCounts(), libmod.GeneCounts()
        nb, nu, info = assignumis.write_tagged_bams_native(gpu_ctx, data, gene_counts=gc_n, **kw)
        pb, pu, _, _ = assignumis.write_tagged_bams(gpu_ctx, data, gene_counts=gc_p, native=True, **kw)
        assert bammodel.bgzf_decompress(bytes(nb)) == bammodel.bgzf_decompress(pb) and bammodel.bgzf_decompress(bytes(nu)) == bammodel.bgzf_decompress(pu)
        assert (gc_n.genecounts_tsv(16), gc_n.umi_depths_tsv(), gc_n.info()) == (gc_p.genecounts_tsv(16), gc_p.umi_depths_tsv(), gc_p.info())
        assert info["records"] == len(names) and info["clustered"] >= n_clustered and (info["batches"] > 3) == ("chunk_size" in kw)
    assert bammodel.bgzf_decompress(bytes(assignumis.write_tagged_bams_native(gpu_ctx, data)[0])) == raw_bc
    # the same file read in segments (a BAM of any size): BGZF blocks, records and BamReader's chunks all straddle the segment borders
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "in.bam"), "wb") as f:
            f.write(data)
        for seg, kw in ((2_500, dict(chunk_size=37)), (20_000, dict(refflat=refflat)), (7_777, dict(chunk_size=61, truncate_read_name=True)), (1 << 20, dict())):
            gc_w = libmod.GeneCounts()
            wb, wu, _info = assignumis.write_tagged_bams_native(gpu_ctx, data, gene_counts=gc_w, **kw)
            res = assignumis.assignumis_stream(gpu_ctx, os.path.join(td, "in.bam"), os.path.join(td, "s"), segment_bytes=seg, **kw)
            assert bammodel.bgzf_decompress(open(os.path.join(td, "s.bam"), "rb").read()) == bammodel.bgzf_decompress(bytes(wb)), (seg, kw)
            assert bammodel.bgzf_decompress(open(os.path.join(td, "s_umifound_.bam"), "rb").read()) == bammodel.bgzf_decompress(bytes(wu))
            assert open(os.path.join(td, "s.genecounts.tsv")).read() == gc_w.genecounts_tsv(16) and open(os.path.join(td, "s.UMIdepths.tsv")).read() == gc_w.umi_depths_tsv()
            assert res["records"] == len(names) and res["batches"] == _info["batches"]
        # three chromosomes (the end of a chromosome closes a chunk whatever its size) and an unmapped tail
        third = len(rows) // 3
        brecs3 = [bammodel.bam_record(nm, fl | (4 if k >= len(rows) - 5 else 0), -1 if k >= len(rows) - 5 else min(k // third, 2), -1 if k >= len(rows) - 5 else p0,
                                      30, [] if k >= len(rows) - 5 else cg, "C" * L, aux=aux_in) for k, (p0, nm, fl, cg, L) in enumerate(rows)]
        header3 = bammodel.bam_bytes("@HD\tVN:1.6\tSO:coordinate\n", [("chr1", 10 ** 6), ("chr2", 10 ** 6), ("chr3", 10 ** 6)], [])
        data3 = bammodel.bgzf_compress(header3 + b"".join(brecs3), block=4096)
        with open(os.path.join(td, "in3.bam"), "wb") as f:
            f.write(data3)
        for seg, kw in ((3_000, dict(chunk_size=25)), (11_000, dict()), (1 << 20, dict(chunk_size=25))):
            gc_w = libmod.GeneCounts()
            wb, wu, _info = assignumis.write_tagged_bams_native(gpu_ctx, data3, gene_counts=gc_w, **kw)
            res = assignumis.assignumis_stream(gpu_ctx, os.path.join(td, "in3.bam"), os.path.join(td, "s3"), segment_bytes=seg, **kw)
            assert bammodel.bgzf_decompress(open(os.path.join(td, "s3.bam"), "rb").read()) == bammodel.bgzf_decompress(bytes(wb)), (seg, kw)
            assert bammodel.bgzf_decompress(open(os.path.join(td, "s3_umifound_.bam"), "rb").read()) == bammodel.bgzf_decompress(bytes(wu))
            assert open(os.path.join(td, "s3.UMIdepths.tsv")).read() == gc_w.umi_depths_tsv() and res["batches"] == _info["batches"] >= 3
