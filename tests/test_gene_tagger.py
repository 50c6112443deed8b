"""GE / GS / XF of assignumis: smi_genes_load_refflat + smi_gene_tag_chunk (host side of the C ABI, no GPU) against the object-by-object
model of tests/genemodel.py, on the first 1500 rows of the reference's own annotation (Data/gencode.v38.chr12.refFlat; a data file, kept
gzipped under tests/golden/) and on hand-made rows for the loader's skip rules.

Reference: GennameTagger.java:L73-366, picard RefFlatReader.java:L70-190, Gene.java, htsjdk OverlapDetector (bytecode under Jar/).
"""
import gzip
import os

import numpy as np
import pytest

import genemodel

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def refflat():
    return gzip.open(os.path.join(HERE, "golden", "chr12_head1500.refFlat.gz"), "rt").read()


def _reads(tree, rng, n):
    """reads that walk transcripts exon by exon (N between exons), reads at random places, soft clips, deletions, both strands"""
    nodes = tree["chr12"]
    out = []
    for _ in range(n):
        (s, e), genes = nodes[int(rng.integers(0, len(nodes)))]
        g = genes[0]
        kind = int(rng.integers(0, 6))
        flag = 16 if rng.random() < 0.5 else 0
        if kind <= 2:  # spliced along one transcript
            t = g.transcripts()[int(rng.integers(0, len(g.transcripts())))]
            ex = t["exons"]
            i = int(rng.integers(0, len(ex)))
            j = min(len(ex), i + int(rng.integers(1, 5)))
            start = int(rng.integers(ex[i][0], ex[i][1] + 1))
            cigar, p = [("S", int(rng.integers(0, 30)))] if kind == 1 else [], start
            for k in range(i, j):
                stop = ex[k][1] if k < j - 1 else int(rng.integers(max(p, ex[k][0]), ex[k][1] + 1))
                if k > i:
                    cigar.append(("N", ex[k][0] - p))
                    p = ex[k][0]
                if stop - p + 1 > 0:
                    cigar.append(("M", stop - p + 1))
                    p = stop + 1
            if not any(op == "M" for op, _ in cigar):
                cigar.append(("M", 1))
            out.append(("chr12", flag, start - 1, cigar))
        elif kind == 3:  # anywhere in or around the gene, one block with an insertion and a deletion
            start = int(rng.integers(max(1, s - 3000), e + 3000))
            out.append(("chr12", flag, start - 1, [("M", int(rng.integers(20, 900))), ("I", 3), ("D", int(rng.integers(1, 40))),
                                                  ("=", int(rng.integers(1, 400))), ("X", 2)]))
        elif kind == 4:  # long read over several genes
            start = int(rng.integers(max(1, s - 20000), e))
            out.append(("chr12", flag, start - 1, [("M", int(rng.integers(500, 4000))), ("N", int(rng.integers(100, 60000))),
                                                  ("M", int(rng.integers(100, 3000)))]))
        else:
            out.append((None, flag | 4, -1, []) if rng.random() < 0.5 else ("chrUn", flag, int(rng.integers(0, 10 ** 6)), [("M", 500)]))
    return out


def test_library_equals_model_on_chr12_rows(pkg, refflat):
    from sicelore_amd import lib

    refs = ["chr1", "chr12", "chrUn"]
    tree, n_genes = genemodel.load_refflat(refflat, refs)
    tagger = lib.GeneTagger(refflat, refs)
    assert tagger.n_lines == 1500 and tagger.n_genes == n_genes and n_genes > 250
    rng = np.random.default_rng(77)
    reads = _reads(tree, rng, 6000)
    got = tagger.tag([refs.index(c) if c is not None else -1 for c, _, _, _ in reads], [f for _, f, _, _ in reads],
                     [p for _, _, p, _ in reads], [cg for _, _, _, cg in reads])
    exp = [genemodel.tag(tree, c, f, p, cg) for c, f, p, cg in reads]
    assert got == exp
    xf = [t[2] for t in got]
    assert all(xf.count(k) > 100 for k in ("INTERGENIC", "INTRONIC", "UTR", "CODING"))
    multi = [t for t in got if t[0] is not None and "," in t[0]]
    assert len(multi) >= 10 and all(t[0].count(",") == t[1].count(",") for t in multi)
    assert sum(t[0] is None and t[2] in ("UTR", "CODING") for t in got) > 50  # exonic on the opposite strand only: no GE


ROWS = [
    # name, transcript, chrom, strand, txStart, txEnd, cdsStart, cdsEnd, n, starts, ends
    ("GOOD", "t1", "c1", "+", 100, 1000, 200, 900, 2, "100,600,", "300,1000,"),
    ("GOOD", "t2", "c1", "+", 150, 1200, 200, 900, 1, "150,", "1200,"),
    ("TWOSTRANDS", "t3", "c1", "+", 5000, 6000, 5000, 6000, 1, "5000,", "6000,"),
    ("TWOSTRANDS", "t4", "c1", "-", 5000, 6000, 5000, 6000, 1, "5000,", "6000,"),
    ("TWICE", "t5", "c1", "-", 7000, 8000, 7000, 8000, 1, "7000,", "8000,"),
    ("TWICE", "t5", "c1", "-", 7000, 8000, 7000, 8000, 1, "7000,", "8000,"),
    ("COUNT", "t6", "c1", "+", 9000, 9500, 9000, 9500, 3, "9000,9200,", "9100,9500,"),
    ("OVERLAP", "t7", "c1", "+", 10000, 10500, 10000, 10500, 2, "10000,10100,", "10100,10500,"),   # end 10100 < start 10101: fine
    ("OVERLAP2", "t8", "c1", "+", 11000, 11500, 11000, 11500, 2, "11000,11099,", "11100,11500,"),  # end 11100 >= start 11100: dropped
    ("EMPTY", "t9", "c1", "+", 12000, 12500, 12000, 12500, 1, "12100,", "12100,"),                  # start 12101 > end 12100: dropped
    ("ELSEWHERE", "t10", "c9", "+", 100, 1000, 100, 1000, 1, "100,", "1000,"),
    ("SAMEPLACE_A", "t11", "c1", "-", 20000, 21000, 20000, 21000, 1, "20000,", "21000,"),
    ("SAMEPLACE_B", "t12", "c1", "-", 20000, 21000, 20500, 21000, 1, "20000,", "21000,"),          # equal to the other as a Gene: one survives
    ("ANTISENSE", "t13", "c1", "+", 20000, 21000, 21000, 21000, 1, "20000,", "21000,"),            # same interval, other strand: both kept
]


def test_loader_skip_rules_and_equal_genes(pkg):
    from sicelore_amd import lib

    text = "".join("\t".join(str(x) for x in r) + "\n" for r in ROWS) + "\n# a comment\n"
    tree, n_genes = genemodel.load_refflat(text, ["c1"])
    names = sorted(g.name for _, node in tree["c1"] for g in node)
    assert n_genes == 4 and names[:2] == ["ANTISENSE", "GOOD"] and names[2] == "OVERLAP" and names[3].startswith("SAMEPLACE_")
    tagger = lib.GeneTagger(text, ["c1"])
    assert (tagger.n_genes, tagger.n_lines) == (4, len(ROWS))
    cases = [(0, 0, 40, [("M", 50)]), (0, 0, 100, [("M", 50)]), (0, 0, 250, [("M", 20)]), (0, 0, 350, [("M", 20)]), (0, 16, 250, [("M", 20)]),
             (0, 0, 1100, [("M", 50)]), (0, 0, 1199, [("M", 50)]), (0, 0, 1200, [("M", 50)]), (0, 16, 20100, [("M", 50)]),
             (0, 0, 20100, [("M", 50)]), (0, 0, 5100, [("M", 50)]), (0, 0, 10090, [("M", 5)]), (0, 0, 250, [("S", 10), ("D", 30)]),
             (0, 0, 11050, [("M", 100)]), (0, 0, 250, [("M", 20), ("N", 19800), ("M", 100)])]
    got = tagger.tag([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], [c[3] for c in cases])
    assert got == [genemodel.tag(tree, "c1", f, p, cg) for _, f, p, cg in cases]
    assert got[0] == (None, None, "INTERGENIC") and got[1] == ("GOOD", "+", "UTR")
    assert got[2] == ("GOOD", "+", "CODING") and got[4] == (None, None, "CODING") and got[5][2] == "UTR"
    assert got[12] == (None, None, None)  # mapped, inside GOOD, only a deletion, so no aligned block: the reference's annotateGene throws, no tag is touched
    assert got[14][0] == "GOOD,ANTISENSE" or got[14][0] == "ANTISENSE,GOOD"
    surviving = [g.name for _, node in tree["c1"] for g in node if g.name.startswith("SAMEPLACE_")]
    assert got[8][0] == surviving[0]


def test_bad_refflat_is_refused(pkg):
    from sicelore_amd import lib

    with pytest.raises(lib.SmiError):
        lib.GeneTagger("A\tb\tc1\t+\t1\t2\n", ["c1"])
    with pytest.raises(lib.SmiError):
        lib.GeneTagger("A\tb\tc1\t+\tx\t2\t1\t2\t1\t1,\t2,\n", ["c1"])


def test_tags_enter_the_record_in_reference_order(pkg):
    assignumis = __import__("importlib").import_module("sicelore_amd.assignumis")
    scan = {"pe": 10, "ps": 5, "ae": 40, "reverse": False, "tso": None, "read_id": 3,
            "bc": {"seq": "ACGTACGTACGTACGT", "start": 41, "end": 56, "ed": 0, "ed_sec": None, "rank": 1}}
    calls, has_bc, _ = assignumis.record_tag_sets(scan, None, None, gene=("G1", "+", "CODING"))
    tags = [t for t, _ in calls]
    assert has_bc and tags[-3:] == ["XF", "GE", "GS"] and tags.index("XF") > tags.index("BH")
    calls2, _, _ = assignumis.record_tag_sets(scan, None, "ACGTACGTACGT", gene=(None, None, "INTRONIC"))
    assert ("GE", None) in calls2 and [t for t, _ in calls2][-3:] == ["U7", "U8", "UZ"]
    # an input record that carries GE / GS from an earlier run loses them; XF goes to its place in htsjdk's attribute order
    fields = assignumis.split_aux(b"GEZold\0GSZ+\0NMc\x01")
    out = assignumis.apply_tag_sets(fields, [("XF", "INTRONIC"), ("GE", None), ("GS", None)])
    assert [t for t, _ in out] == ["XF", "NM"]  # in front of GS, the first larger binary tag ("SG" > "FX")
    calls3, _, _ = assignumis.record_tag_sets(scan, None, None, gene=(None, None, None))
    assert not any(t in ("XF", "GE", "GS") for t, _ in calls3)


def test_loader_refuses_a_map_whose_bin_would_be_a_tree(pkg):
    """java.util.HashMap turns a bin of 8 entries into a red-black tree (ordered by hash, then identityHashCode): the iteration order of such a
    map is not modelled, so the loader refuses the annotation instead of guessing (as tools/jvm_natives.py does when the fixtures are made)"""
    from sicelore_amd import lib as libmod

    def jhash(s):
        h = 0
        for ch in s:
            h = (31 * h + ord(ch)) & 0xFFFFFFFF
        return h

    # nine gene names whose spread hashes fall into one bin of a 16-slot table ("Aa" / "BB" blocks hash alike: equal full hashes)
    names = []
    for k in range(512):
        nm = "".join("Aa" if (k >> b) & 1 else "BB" for b in range(9))
        names.append(nm)
    assert len({jhash(n) for n in names[:9]}) == 1
    rows = "".join(f"{nm}\tT{i}\tchr1\t+\t{1000 * i}\t{1000 * i + 500}\t{1000 * i}\t{1000 * i + 500}\t1\t{1000 * i},\t{1000 * i + 500},\n" for i, nm in enumerate(names[:9]))
    with pytest.raises(libmod.SmiError, match="tree"):
        libmod.GeneTagger(rows, ["chr1"])
    libmod.GeneTagger(rows.split("\n", 2)[2], ["chr1"])          # seven of them: modelled
