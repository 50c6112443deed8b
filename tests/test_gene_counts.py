"""<out>.genecounts.tsv / <out>.UMIdepths.tsv of assignumis (smi_gene_counts_*; GeneCounts.java) against the texts the reference's own
GeneCounts bytecode wrote for the same records (tests/golden/ref_exec_genecounts.json: updateGeneCounts per record, printCountTable,
printUmisPerCellTable, mergeGeneCounts).  Rows with equal totals come in ConcurrentHashMap order there (the fixture holds several outcomes);
compared as: the matrix cell by cell, the sort keys along rows and columns, and UMIdepths.tsv character for character (rank and sorted
counts do not depend on the order of ties)."""
import json
import os
import random

import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CIGAR_OPS = "MIDNSHP=X"


@pytest.fixture
def libmod(pkg):
    from sicelore_amd import lib as libmod

    libmod.load_library()
    return libmod


def _cigar_ends(text):
    ops, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            ops.append((int(num) << 4) | CIGAR_OPS.index(ch))
            num = ""
    return (ops[0], ops[-1]) if ops else (0xFFFFFFFF, 0)


def _add(libmod, gc, recs, five_prime):
    code = libmod.two_bit_code
    first, last = zip(*[_cigar_ends(r["cigar"]) for r in recs])
    gc.add([r["gene"] for r in recs], [-1 if r["region"] is None else r["region"] for r in recs],
           [code(r["bc"]) if r["bc"] else 0 for r in recs], [code(r["u8"]) if r["u8"] else 0 for r in recs],
           [1 if r["bc"] and r["u8"] else 0 for r in recs], [r["flag"] for r in recs], [r["mapq"] for r in recs], first, last,
           [r["nth"] for r in recs], five_prime=five_prime)


class Model:
    """GeneCounts restated for the test (the counter arithmetic of GeneCounts$UMIcounts L617-650 as written)"""

    def __init__(self):
        self.genes, self.regions, self.with_gene, self.skipped = {}, {}, 0, 0

    @staticmethod
    def _inc(d, nth):
        d = (d + 1) & 0xFFFFFFFF
        return ((d & 4095) | ((d & 7) << 12)) if nth else d

    @staticmethod
    def excluding(d):
        return d & (4095 - ((d & 61440) >> 12))

    def add(self, recs, five_prime):
        for r in recs:
            if r["flag"] & 4 or r["flag"] & 0x900 or r["mapq"] == 0:
                continue
            first, last = _cigar_ends(r["cigar"])
            rev = bool(r["flag"] & 16)
            ce = first if (not rev if five_prime else rev) else last
            if (ce >> 4) > 150 and (ce & 15) in (4, 5):
                self.skipped += 1
                continue
            if not (r["bc"] and r["u8"]):
                continue
            if r["gene"] is not None:
                self.with_gene += 1
                k = (r["gene"], r["bc"], r["u8"])
                self.genes[k] = self._inc(self.genes.get(k, 0), r["nth"])
            if r["region"] is not None:
                k = (r["region"], r["bc"], r["u8"])
                self.regions[k] = self._inc(self.regions.get(k, 0), r["nth"])

    def per_cell(self, table):
        m = {}
        for (_, cell, _), d in table.items():
            m[cell] = m.get(cell, 0) + (1 if self.excluding(d) else 0)
        return m

    def matrix(self):
        m = {}
        for (gene, cell, _), _d in self.genes.items():
            m[(gene, cell)] = m.get((gene, cell), 0) + 1
        return m

    def texts(self, code):
        """the product's rule for ties: ascending 2-bit code of the cell, ascending gene name"""
        pc = self.per_cell(self.genes)
        cells = sorted(pc, key=lambda c: (-pc[c], code(c)))
        mat = self.matrix()
        tot = {}
        for (gene, _), v in mat.items():
            tot[gene] = tot.get(gene, 0) + v
        a = "\t" + "\t".join(cells) + "\n"
        for gene in sorted(tot, key=lambda x: (-tot[x], x.encode())):
            a += gene + "\t" + "\t".join(str(mat.get((gene, c), 0)) for c in cells) + "\n"
        pr = sorted(self.per_cell(self.regions).values(), reverse=True)
        pg = sorted(pc.values(), reverse=True)
        b = "Cell\tnUMIs based on genomic regions\tnUMIs based on genes\n"
        for i, v in enumerate(pr):
            b += f"{i + 1}\t{v}" + (f"\t{pg[i]}" if i < len(pg) else "") + "\n"
        return a, b


def _parse(text):
    lines = text.split("\n")
    assert lines[-1] == ""
    cells = lines[0].split("\t")[1:] if lines[0] != "\t" else []
    assert lines[0].startswith("\t")
    mat, order = {}, []
    for ln in lines[1:-1]:
        f = ln.split("\t")
        assert len(f) == 1 + len(cells)
        order.append(f[0])
        for c, v in zip(cells, f[1:]):
            if v != "0":
                mat[(f[0], c)] = int(v)
    return cells, order, mat


def _check_against_reference_text(ref, model, got_a, got_b):
    """ref: one outcome of the reference; model: Model of the same records; got_*: the product's texts"""
    assert got_b == ref["umidepths"]
    rc, ro, rm = _parse(ref["genecounts"])
    pc, po, pm = _parse(got_a)
    assert rm == pm == model.matrix()
    assert sorted(rc) == sorted(pc) and sorted(ro) == sorted(po)
    per_cell = model.per_cell(model.genes)
    tot = {}
    for (gene, _), v in rm.items():
        tot[gene] = tot.get(gene, 0) + v
    for cells, genes in ((rc, ro), (pc, po)):              # the sort keys fall along the columns and down the rows, in both texts
        assert [per_cell[c] for c in cells] == sorted((per_cell[c] for c in cells), reverse=True)
        assert [tot[x] for x in genes] == sorted((tot[x] for x in genes), reverse=True)


def test_tables_equal_the_reference_texts(libmod):
    sec = json.load(open(os.path.join(GOLD, "ref_exec_genecounts.json")))["sections"][0]
    n_ties = 0
    for case in sec["cases"]:
        gc = libmod.GeneCounts()
        _add(libmod, gc, case["records"], case["five_prime"])
        model = Model()
        model.add(case["records"], case["five_prime"])
        a, b = gc.genecounts_tsv(16), gc.umi_depths_tsv()
        info = gc.info()
        for ref in case["texts"]:
            _check_against_reference_text(ref, model, a, b)
            assert info["records_with_gene"] == ref["recordsWithGene"] == ref["incrementRecordsForReadsUsedInGeneCounts"] == model.with_gene
            assert info["records_skipped_clipping"] == ref["recordsWithGeneSkippedClipping"] == model.skipped
        assert (a, b) == model.texts(libmod.two_bit_code)
        n_ties += len(case["texts"]) > 1
        gc.close()
    assert len(sec["cases"]) >= 6 and n_ties >= 3            # the reference's own order of ties did vary with the container order


def test_merge_equals_the_reference_merge(libmod):
    """mergeGeneCounts: counters of the same (gene, cell, UMI) through UMIcounts.add, a region number of a later object replaces the first one's"""
    m = json.load(open(os.path.join(GOLD, "ref_exec_genecounts.json")))["sections"][0]["merged"]
    parts = []
    for recs in m["parts"]:
        gc = libmod.GeneCounts()
        _add(libmod, gc, recs, m["five_prime"])
        parts.append(gc)
    for other in parts[1:]:
        parts[0].merge(other)
    a, b = parts[0].genecounts_tsv(16), parts[0].umi_depths_tsv()
    for ref in m["texts"]:
        assert b == ref["umidepths"]
        rc, ro, rm = _parse(ref["genecounts"])
        pc, po, pm = _parse(a)
        assert rm == pm and sorted(rc) == sorted(pc) and sorted(ro) == sorted(po)
    # and it is not what adding all records to one object gives (the region replacement, the AND arithmetic of add())
    one = libmod.GeneCounts()
    for recs in m["parts"]:
        _add(libmod, one, recs, m["five_prime"])
    assert one.umi_depths_tsv() != b


def test_larger_random_sets_equal_the_model(libmod):
    rng = random.Random(99)
    for five_prime in (False, True):
        cells = ["".join(rng.choice("ACGT") for _ in range(16)) for _ in range(300)]
        genes = [f"G{k}" for k in range(500)]
        mols = [(rng.choice(genes) if rng.random() < 0.8 else None, rng.randrange(2000) if rng.random() < 0.9 else None, rng.choice(cells),
                 "".join(rng.choice("ACGT") for _ in range(12))) for _ in range(20000)]
        recs = []
        for _ in range(60000):
            g, reg, cell, umi = rng.choice(mols)
            cl, cr = rng.choice([0, 0, 151, 30]), rng.choice([0, 0, 151, 30])
            recs.append({"gene": g, "region": reg, "bc": cell, "u8": umi if rng.random() < 0.97 else None,
                         "flag": (16 if rng.random() < 0.5 else 0) | (0x800 if rng.random() < 0.03 else 0), "mapq": rng.choice([0, 20, 60, 60]),
                         "cigar": (f"{cl}S" if cl else "") + "500M" + (f"{cr}H" if cr else ""), "nth": 1 if rng.random() < 0.1 else 0})
        gc, model = libmod.GeneCounts(), Model()
        for k in range(0, len(recs), 7000):                      # batch by batch, as the writers hand them over
            _add(libmod, gc, recs[k:k + 7000], five_prime)
        model.add(recs, five_prime)
        assert (gc.genecounts_tsv(16), gc.umi_depths_tsv()) == model.texts(libmod.two_bit_code)
        assert gc.info()["records_skipped_clipping"] == model.skipped > 0


def test_counter_quirks(libmod):
    """UMIcounts as written: a record flagged as a further alignment stores `count & 7` as the duplicate figure, and a counter is "empty
    excluding duplicates" when count AND (4095 - duplicates) is zero -- e.g. one record that is a further alignment: 1 & (4095 - 1) = 0"""
    code = libmod.two_bit_code
    gc = libmod.GeneCounts()
    bc, u = code("ACGTACGTACGTACGT"), code("AAAACCCCGGGG")
    gc.add(["G"], [1], [bc], [u], [1], [0], [60], [100 << 4], [100 << 4], [1])
    assert gc.umi_depths_tsv().split("\n")[1] == "1\t0\t0"                 # the cell is listed with no UMI ...
    assert gc.genecounts_tsv(16) == "\tACGTACGTACGTACGT\nG\t1\n"           # ... while the matrix counts the entry
    gc.add(["G"], [1], [bc], [u], [1], [0], [60], [100 << 4], [100 << 4], [0])
    assert gc.umi_depths_tsv().split("\n")[1] == "1\t1\t1"                 # 2 & (4095 - 1) = 2
    # a base that is not ACGT ORs -2 into the code like the reference's table
    assert code("AN") == 0xFFFFFFFFFFFFFFFE and code("ACGT") == 0b00100111
