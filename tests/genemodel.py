"""Python model of the reference's gene tagger for the tests (test infrastructure, like tests/pymodel.py): object by object what the bytecode
does, java.util.HashMap / HashSet as real bucket tables (pymodel.JHashSet) -- a second formulation beside the sorted-bucket shortcut of
sicelore-2.1_amd/csrc/smi_gene.hip.

  GennameTagger                         FJ!umifinder/bamreaders/GennameTagger.java:L73-366
  RefFlatReader.load and helpers        picard-2.23.9.jar!/picard/annotation/RefFlatReader.java:L70-190
  Gene, Gene$Transcript                 picard/annotation/Gene.java
  OverlapDetector, Interval             htsjdk-4.1.3.jar!/htsjdk/samtools/util/
"""
from pymodel import JHashSet

INTERGENIC, INTRONIC, UTR, CODING, RIBOSOMAL = range(5)
NAMES = ["INTERGENIC", "INTRONIC", "UTR", "CODING", "RIBOSOMAL"]
SCORE = {CODING: 4, UTR: 3, INTRONIC: 2, INTERGENIC: 1}


def jhash(s):
    h = 0
    for c in s.encode():
        h = (31 * h + c) & 0xFFFFFFFF
    return h


class JMap:
    """HashMap<K, V> with put-if-absent / replace semantics; iteration = bucket order"""

    def __init__(self):
        self.s = JHashSet()
        self.cell = {}

    def put(self, h, key, value):
        if key in self.cell:
            self.cell[key][1] = value
            return
        c = [key, value]
        self.cell[key] = c
        self.s.add(h & 0xFFFFFFFF, key, c)

    def get(self, key):
        c = self.cell.get(key)
        return None if c is None else c[1]

    def items(self):
        return [(c[0], c[1]) for c in self.s]

    def __len__(self):
        return len(self.cell)


class Gene:
    def __init__(self, contig, start, end, negative, name):
        self.contig, self.start, self.end, self.negative, self.name = contig, start, end, negative, name
        self.tx = JMap()

    def key(self):  # Gene.equals / compareTo: interval + strand
        return (self.contig, self.start, self.end, self.negative)

    def hash(self):  # Interval.hashCode
        return (31 * (31 * jhash(self.contig) + self.start) + self.end) & 0xFFFFFFFF

    def transcripts(self):
        return [t for _, t in self.tx.items()]


class Annotation(Exception):
    pass


def load_refflat(text, ref_names):
    """-> list of Gene as the OverlapDetector holds them, per contig in interval-tree order: {contig: [(start, end, [genes])]}"""
    by_name = JMap()
    for line in text.split("\n"):
        line = line.rstrip("\r")
        if not line or line.startswith("#"):
            continue
        f = line.split("\t")
        assert len(f) == 11
        if f[2] not in ref_names:
            continue
        rows = by_name.get(f[0])
        if rows is None:
            by_name.put(jhash(f[0]), f[0], [f])
        else:
            rows.append(f)
    tree = {}
    n_genes = 0
    for _, rows in by_name.items():
        try:
            g = make_gene(rows)
        except Annotation:
            continue
        nodes = tree.setdefault(g.contig, {})
        node = nodes.setdefault((g.start, g.end), [])
        if all(o.key() != g.key() for o in node):  # a HashSet<Gene> whose members share one hash: insertion order
            node.append(g)
            n_genes += 1
    return {c: sorted(nodes.items()) for c, nodes in tree.items()}, n_genes


def jsplit(s):
    p = s.split(",")
    if len(p) > 1:
        while p and p[-1] == "":
            p.pop()
    return p


def make_gene(rows):
    name, chrom, strand = rows[0][0], rows[0][2], rows[0][3]
    g = Gene(chrom, min(int(r[4]) + 1 for r in rows), max(int(r[5]) for r in rows), strand == "-", name)
    for r in rows:
        if r[3] != strand or r[2] != chrom:
            raise Annotation("strand / chromosome disagreement")
        cnt, es, ee = int(r[8]), jsplit(r[9]), jsplit(r[10])
        if cnt != len(es) or cnt != len(ee):
            raise Annotation("exon count")
        if g.tx.get(r[1]) is not None:
            raise Annotation("transcript twice")
        t = {"name": r[1], "tx": (int(r[4]) + 1, int(r[5])), "cds": (int(r[6]) + 1, int(r[7])), "exons": []}
        g.tx.put(jhash(r[1]), r[1], t)
        for i in range(cnt):
            e = (int(es[i]) + 1, int(ee[i]))
            if e[0] > e[1]:
                raise Annotation("empty exon")
            if i and t["exons"][-1][1] >= e[0]:
                raise Annotation("exons overlap")
            t["exons"].append(e)
    return g


def in_exon(t, locus):
    for s, e in t["exons"]:
        if s > locus:
            return False
        if s <= locus <= e:
            return True
    return False


def top(fs):
    best = None
    for f in fs:
        if best is None or SCORE[f] > SCORE[best]:
            best = f
    return best


def blocks_of(pos1, cigar):
    out, ref = [], pos1
    for op, ln in cigar:
        if op in "M=X":
            out.append((ref, ln))
            ref += ln
        elif op in "DN":
            ref += ln
    return out, ref - 1


def gene_set(genes):
    s = JHashSet()
    for g in genes:
        s.add(g.hash(), g.key(), g)
    return s


def tag(tree, contig, flag, pos0, cigar):
    """-> (GE, GS, XF) with None for removed / untouched, as lib.GeneTagger.tag gives them"""
    unmapped = bool(flag & 4) or contig is None
    blocks, end = ([], pos0) if unmapped else blocks_of(pos0 + 1, cigar)
    over = JHashSet()
    if not unmapped and end >= pos0 + 1:
        for (s, e), node in tree.get(contig, []):
            if s <= end and e >= pos0 + 1:
                for g in node:
                    over.add(g.hash(), g.key(), g)
    fmap = JMap()
    for g in over:
        per_block = []
        for bs, bl in blocks:
            lf = [INTERGENIC] * bl
            for t in g.transcripts():
                for p in range(max(bs, t["tx"][0]), min(t["tx"][1], bs + bl - 1) + 1):
                    if lf[p - bs] > CODING:
                        continue
                    f = (UTR if (p < t["cds"][0] or p > t["cds"][1]) else CODING) if in_exon(t, p) else INTRONIC
                    if f > lf[p - bs]:
                        lf[p - bs] = f
            per_block.append(top(lf))
        rf = top(per_block)
        if rf is None:
            return None, None, None  # Collectors.toMap refuses a null value: annotateGene throws, the caller logs it
        fmap.put(g.hash(), g.key(), (g, rf))
    keys = [g for _, (g, _) in fmap.items()]
    result = JHashSet()
    for bs, bl in blocks:
        bg = JHashSet()
        for g in keys:
            if any(s <= bs + bl - 1 and bs <= e for t in g.transcripts() for s, e in t["exons"]):
                bg.add(g.hash(), g.key(), g)
        for g in bg:
            result.add(g.hash(), g.key(), g)
    genes = [g for g in result if fmap.get(g.key())[1] in (CODING, UTR)]
    f = INTERGENIC if len(fmap) == 0 else top([v for _, (_, v) in fmap.items()])
    neg = bool(flag & 16)
    same = [g for g in genes if g.negative == neg]
    if not same:
        return None, None, NAMES[f]
    return ",".join(g.name for g in same), ",".join("-" if g.negative else "+" for g in same), NAMES[f]
