#!/usr/bin/env python3
"""Generate tests/golden/ref_exec_*.json: answers computed BY THE REFERENCE'S OWN BYTECODE.

Build container only: needs /root/reference/Jar/*.jar (never copied, never shipped) and tools/jvm_exec.py, the bytecode
interpreter written for this purpose (the image has no JVM).  Every fixture file records, per section, the class and
method that was executed, its inputs and its outputs, and the JDK natives the interpreter had to supply while it ran
(`natives`, with their tier: A = language level, B = java.lang value classes with specified behaviour, C = ordered
containers / membership-only sets / ordered sequential streams; see tools/jvm_exec.py).  Hash-ordered iteration is never
emulated: where a method iterates a HashMap / HashSet the case is run under several different iteration orders and kept
only if all of them agree (`hash_orders_agree`), otherwise it is listed under `excluded`.

usage: python tools/make_ref_exec.py [--jobs N] [section ...]     (default: all sections; N processes, one section each)
       python tools/make_ref_exec.py --coverage               merge tests/golden/coverage/*.json -> tests/golden/ref_exec_coverage.json

Coverage: while a section runs, the interpreter marks every executed instruction (tools/jvm_exec.py, JVM.coverage); the
section's hits on the hot-path classes (COVERAGE_CLASSES: SURVEY 8a) are written to tests/golden/coverage/<section>.json,
and `--coverage` folds them with each method's LineNumberTable into lines present / lines executed / lines never reached
(with the reason from tools/coverage_notes.json where one was written down).
"""
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from jvm_exec import JVM, JArray, JBox, JLambda, JObject, JavaThrow, Unsupported, f32  # noqa: E402
import jvm_natives  # noqa: E402

REF = "/root/reference/Jar/"
JARS = [REF + "NanoporeBC_UMI_finder-2.1.jar", REF + "lib/TwoFourBitNucAcidLibraryMaven-1.0.jar",
        REF + "lib/Aliasi_ClusteringLib-1.0.jar", REF + "lib/commons-lang3-3.17.0.jar", REF + "lib/htsjdk-4.1.3.jar",
        REF + "lib/guava-33.3.1-jre.jar", REF + "lib/picard-2.23.9.jar", REF + "lib/DropseqLib-1.0.jar"]
OUT = os.environ.get("REF_EXEC_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")   # committed fixtures that later sections take their inputs from
COV_DIR = os.path.join(OUT, "coverage")

# the classes of SURVEY 8a (rows a-1 .. a-18) + the 8f callers either side whose bytecode the fixtures execute
_FJ = "com/rw/"
COVERAGE_CLASSES = [_FJ + c for c in (
    "nanoporereadscanner/analyzers/Parser", "nanoporereadscanner/analyzers/BarcodeMatchTester",
    "nanoporereadscanner/analyzers/BarcodeMatchTester$Matches", "nanoporereadscanner/analyzers/BarcodeMatchTester$Matches$OneMatch",
    "nuc/encoding/TwoBit/ed/NucTwoBitPerBaseEDtesterBase", "nuc/encoding/TwoBit/LongSeqMutated",
    "nanoporereadscanner/analyzers/PolyATadapterAnalyzer_3pBCUMI", "nanoporereadscanner/analyzers/PolyATadapterAnalyzer_5pBCUMI",
    "nanopore/analyzers/PolyATadapterAnalyzerBase", "nanopore/analyzers/PolyATSearcher", "nanopore/analyzers/PolyATSearcher$PolyAscanResult",
    "nanopore/analyzers/AdapterTSOanalyzer", "nanopore/analyzers/AdapterTSOanalyzer$AdapterScanRslt",
    "nanopore/analyzers/Match", "nanopore/analyzers/NeedlemanMatch", "nanopore/analyzers/apachemod/LevenshteinDistance",
    "nanoporereadscanner/analyzers/ChimeraFindernew", "nanoporereadscanner/analyzers/ChimeraFindernew$AdapterTSOmatch",
    "nanoporereadscanner/analyzers/ChimeraFindernew$SplitPosition", "nanopore/analyzers/PolyATadapterInternalSearcherBase",
    "nanopore/analyzers/PolyATadapterInternalSearcherBase$ATposition",
    "nanoporereadscanner/analyzers/UsedCellBCListGenerator$Worker", "nanoporereadscanner/analyzers/UsedCellBCListGenerator$UsedBarcodesListData",
    "nanoporereadscanner/analyzers/BarcodeDatasetColissionTester",
    "nanoporereadscanner/readerwriter/FastqRecordExt", "nanoporereadscanner/readerwriter/ReadScanResult",
    "nanoporereadscanner/stats/ReadFlags", "nanoporereadscanner/stats/ReadFlags$Flags",
    "clustering/ClusteringEditDistanceBase", "clustering/DistanceMatrix", "clustering/OneUmiCluster",
    "umifinder/analyzers/clustering/ClusterOneBase", "umifinder/analyzers/clustering/ClusterOneHierarchical",
    "umifinder/analyzers/clustering/ClusterOne_MyClustering", "umifinder/analyzers/clustering/UmiClustering",
    "umifinder/bamreaders/ReadGrouper", "umifinder/bamreaders/ReadGrouper$Cluster", "umifinder/bamreaders/ReadGrouper$ClusterList",
    "umifinder/bamreaders/ReadGrouper$NanoporeReadWithOrderedPosition", "umifinder/bamreaders/GennameTagger",
    "umifinder/reads/nanopore/NanoporeRead", "umifinder/reads/nanopore/NanoporeRead$ReadScanData", "umifinder/scanstats/GeneCounts",
    "nuc/encoding/TwoBit/NucleicAcidTwoBitPerBase", "nuc/encoding/onebyte/NucleicAcidInmutableOneBytePerBase",
    "nuc/encoding/onebyte/NucleicAcidInmutableOneBytePerBase$Kmers", "nuc/encoding/NucleicAcidByteCodeBase",
    "nuc/alignment/needleman/NeedlemanWunsch", "nuc/alignment/needleman/DynamicProgramming", "nuc/alignment/needleman/SequenceAlignment",
)] + ["com/aliasi/cluster/CompleteLinkClusterer", "com/aliasi/cluster/SingleLinkClusterer", "com/aliasi/cluster/AbstractHierarchicalClusterer",
      "com/aliasi/cluster/Dendrogram", "com/aliasi/cluster/LinkDendrogram", "com/aliasi/cluster/LeafDendrogram",
      "com/aliasi/util/BoundedPriorityQueue"] + [
      # --annotationFile <x.gtf> (round 5): DropseqLib's GTF reader under GennameTagger
      "org/broadinstitute/dropseqrna/annotation/" + c for c in ("GTFReader", "GTFReader$FilteringGTFParser", "GTFParser", "GTFRecord", "GeneFromGTFBuilder",
                                                               "GeneFromGTFBuilder$Exon", "GeneFromGTFBuilder$GeneAnnotationFilter", "GeneFromGTF",
                                                               "GeneFromGTF$TranscriptFromGTF", "AnnotationUtils")]

TB = "com/rw/nuc/encoding/TwoBit/NucleicAcidTwoBitPerBase"
OBI = "com/rw/nuc/encoding/onebyte/NucleicAcidInmutableOneBytePerBase"
OB = "com/rw/nuc/encoding/onebyte/NucleicAcidOneBytePerBase"
BCB = "com/rw/nuc/encoding/NucleicAcidByteCodeBase"
NS = "com/rw/nuc/alignment/needleman/NeedlemanScores"
NW = "com/rw/nuc/alignment/needleman/NeedlemanWunsch"
MATCH = "com/rw/nanopore/analyzers/Match"
NM = "com/rw/nanopore/analyzers/NeedlemanMatch"
LEV = "com/rw/nanopore/analyzers/apachemod/LevenshteinDistance"
BMT = "com/rw/nanoporereadscanner/analyzers/BarcodeMatchTester"
PS = "com/rw/nanopore/analyzers/PolyATSearcher"


def u64(v):
    return v & 0xFFFFFFFFFFFFFFFF


class Gen:
    def __init__(self):
        self.j = JVM(JARS, max_steps=1 << 40)     # (the interpreter's own default of 2 G steps is a runaway guard for tests; an ed-2 section is 5 G)
        self.t0 = time.time()

    def natives(self):
        used = sorted(self.j.natives_used)
        return [{"native": k, "tier": jvm_natives.tier_of(k)} for k in used]

    def section(self, title, cls, method):
        self.j.natives_used.clear()
        return {"reference_class": cls, "reference_method": method, "title": title, "cases": []}

    def finish(self, sec):
        sec["natives"] = self.natives()
        sec["max_tier"] = max([n["tier"] for n in sec["natives"]] or ["A"])
        return sec

    def hits(self):
        """{class: {"method:desc": hex of the executed-instruction bitmap}} for the hot-path classes this JVM loaded"""
        out = {}
        for cname in COVERAGE_CLASSES:
            jc = self.j.classes.get(cname)
            if jc is None:
                continue
            cm = {f"{n}:{d}": m.hit.hex() for (n, d), m in jc.methods.items() if m.hit is not None and any(m.hit)}
            if cm:
                out[cname] = cm
        return out


def merge_hits(dst, src):
    for cname, cm in src.items():
        d = dst.setdefault(cname, {})
        for mk, hx in cm.items():
            if mk in d:
                a, b = bytes.fromhex(d[mk]), bytes.fromhex(hx)
                d[mk] = bytes(x | y for x, y in zip(a, b)).hex()
            else:
                d[mk] = hx
    return dst


def rnd_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


# ---------------------------------------------------------------------------------------------------------------------
def gen_twobit(g):
    """a-1 / a-2: NucleicAcidTwoBitPerBase -- <clinit> tables, getLongHashForSeq, mutate ops, reverseComplement"""
    j = g.j
    rng = random.Random(101)
    out = {"jar": "TwoFourBitNucAcidLibraryMaven-1.0.jar", "sections": []}
    s = g.section("static tables built by <clinit> (L72-136)", TB, "<clinit>")
    jc = j.init_class(TB)
    for name, v in sorted(jc.statics.items()):
        if isinstance(v, JArray):
            s["cases"].append({"table": name, "values": [[u64(x) for x in r.a] if isinstance(r, JArray) else (u64(r) if v.etype == "J" else r)
                                                          for r in v.a]})
    out["sections"].append(g.finish(s))

    s = g.section("getLongHashForSeq(char[]) (L183-187), incl. non-ACGT characters", TB, "getLongHashForSeq:([C)J")
    seqs = [rnd_seq(rng, 16) for _ in range(40)] + [rnd_seq(rng, n) for n in (1, 2, 8, 12, 15, 17, 24, 31, 32)]
    seqs += ["ACGTNACGTACGTACG", "NNNNNNNNNNNNNNNN", "acgtacgtacgtacgt", "ACGTACGTACGTACGN", "NCGTACGTACGTACGT", "ACGTRYACGTACGTAC"]
    for q in seqs:
        s["cases"].append({"seq": q, "hash": u64(j.call_static(TB, "getLongHashForSeq", "([C)J", j.char_array(q)))})
    out["sections"].append(g.finish(s))

    s = g.section("longTwoBitToString(long, int) (L337-342)", TB, "longTwoBitToString:(JI)Ljava/lang/String;")
    for _ in range(20):
        n = rng.choice([16, 16, 16, 12, 10, 8, 20])
        v = rng.getrandbits(2 * n)
        s["cases"].append({"value": v, "length": n, "string": j.call_static(TB, "longTwoBitToString", "(JI)Ljava/lang/String;", v, n)})
    out["sections"].append(g.finish(s))

    for name, desc, title in (("getLongHashReplaceByteDeg", "(J[JII)V", "substitution variants (L228-233)"),
                              ("getLongHashInsertByteDeg", "(J[JII)V", "insertion variants (L300-309), incl. pos = length-2 (shift count 64 wraps)")):
        s = g.section(title, TB, f"{name}:{desc}")
        for k in range(24):
            n = 16 if k < 16 else rng.choice([12, 10, 14])
            v = rng.getrandbits(2 * n)
            for pos in range(n if "Replace" in name else n - 1):
                res = j.long_array([0, 0, 0, 0])
                j.call_static(TB, name, desc, v, res, pos, n)
                s["cases"].append({"seq": v, "pos": pos, "length": n, "out": [u64(x) for x in res.a]})
        out["sections"].append(g.finish(s))

    s = g.section("deletion (L321-327): getLongHashdeleteByte(seq, 4-bit code of the appended base, pos, length)", TB, "getLongHashdeleteByte:(JBII)J")
    for k in range(12):
        n = 16 if k < 9 else 12
        v = rng.getrandbits(2 * n)
        for pos in range(n - 1):
            for b4 in (1, 2, 4, 8, 15):
                s["cases"].append({"seq": v, "base4": b4, "pos": pos, "length": n,
                                   "out": u64(j.call_static(TB, "getLongHashdeleteByte", "(JBII)J", v, b4, pos, n))})
    out["sections"].append(g.finish(s))

    s = g.section("reverseComplement() (L477-484) on objects built from strings (N poisons the long, L185)", TB, "reverseComplement:()L...;")
    for q in [rnd_seq(rng, 16) for _ in range(30)] + ["ACGTNACGTACGTACG", "NNNNNNNNNNNNNNNN", "ACGTACGTACGTACGN", "NCGTACGTACGTACGT"] + \
            [rnd_seq(rng, 16, "ACGTN") for _ in range(10)] + [rnd_seq(rng, n) for n in (8, 10, 12, 14)]:
        o = j.new(TB, "(Ljava/lang/String;)V", q)
        r = j.call_virtual(o, "reverseComplement", f"()L{TB};")
        s["cases"].append({"seq": q, "sequence": u64(o.f["sequence"]), "length": o.f["seqlength"], "rc_sequence": u64(r.f["sequence"]),
                           "rc_string": j.call_virtual(r, "toString", "()Ljava/lang/String;")})
    out["sections"].append(g.finish(s))
    return out


def gen_onebyte(g):
    """a-3 / a-4: 4-bit IUPAC codec, reverse complement, sub-sequences, the 4-mer gate"""
    j = g.j
    rng = random.Random(202)
    out = {"jar": "TwoFourBitNucAcidLibraryMaven-1.0.jar", "sections": []}
    s = g.section("static tables of the byte codec (L41-133)", BCB, "<clinit>")
    jc = j.init_class(BCB)
    for name, v in sorted(jc.statics.items()):
        if isinstance(v, JArray) and v.etype in ("B", "C", "I", "[B"):
            s["cases"].append({"table": name, "values": [list(r.a) if isinstance(r, JArray) else r for r in v.a]})
    out["sections"].append(g.finish(s))

    s = g.section("encode / toString / reverseComplement / getSubSequence(start1, len) / getByteAt(1-based)", OBI, "<init>(CharSequence), reverseComplement, getSubSequence, getByteAt")
    for q in [rnd_seq(rng, n, "ACGTN") for n in (5, 10, 22, 30, 43)] + ["ACGTRYKMSWBDHVN", "acgtn", "AAAAACCCCCGGGGGTTTTT"]:
        o = j.new(OBI, "(Ljava/lang/CharSequence;)V", q)
        codes = list(o.f["naData"].a)
        n = len(q)
        st, ln = 1 + n // 4, max(1, n // 2)
        sub = j.call_virtual(o, "getSubSequence", f"(II)L{OBI};", st, ln)
        byte_at = [j.call_virtual(o, "getByteAt", "(I)B", k) for k in range(1, n + 1)]
        string = j.call_virtual(o, "toString", "()Ljava/lang/String;")
        rc_copy = j.call_virtual(o, "reverseComplementCopy", f"()L{OBI};")
        untouched = list(o.f["naData"].a) == codes
        rc = j.call_virtual(o, "reverseComplement", f"()L{OBI};")     # works IN PLACE and returns this
        s["cases"].append({"seq": q, "codes": codes, "string": string, "rc_codes": list(rc_copy.f["naData"].a),
                           "copy_leaves_original": untouched, "reverse_complement_in_place": rc is o and list(o.f["naData"].a) == list(rc_copy.f["naData"].a),
                           "sub_start1": st, "sub_len": ln, "sub_codes": list(sub.f["naData"].a), "byte_at": byte_at})
    out["sections"].append(g.finish(s))

    s = g.section("$Kmers.nKmersMatching(read, pos) with 4-mers (nKmersMatching_4mer L533-543)", OBI + "$Kmers", "nKmersMatching:(L...;I)I")
    adapters = ["CTTCCGATCT", "CTACACGACGCTCTTCCGATCT", "AAGCAGTGGTATCAAC", "AAGCAGTGGTATCAACGCAGAGTACAT"]
    for ad in adapters:
        a = j.new(OBI, "(Ljava/lang/CharSequence;)V", ad)
        k = j.new(OBI + "$Kmers", f"(L{OBI};I)V", a, 4)
        for t in range(6):
            # a read with a noisy copy of the adapter somewhere inside
            noisy = "".join(c if rng.random() > 0.12 else rng.choice("ACGTN") for c in ad)
            pre = rnd_seq(rng, rng.randrange(3, 12))
            read = pre + noisy + rnd_seq(rng, 14 + len(ad))
            r = j.new(OBI, "(Ljava/lang/CharSequence;)V", read)
            counts = [j.call_virtual(k, "nKmersMatching", f"(L{OBI};I)I", r, p) for p in range(1, len(read) - len(ad) + 1)]
            s["cases"].append({"adapter": ad, "read": read, "first_pos1": 1, "counts": counts})
    out["sections"].append(g.finish(s))
    return out


def nw_align(j, adapter, read_slice, scores=(-4, -5, -5, -5, -5, -5, 5)):
    a = j.new(OBI, "(Ljava/lang/CharSequence;)V", adapter)
    b = j.new(OBI, "(Ljava/lang/CharSequence;)V", read_slice)
    sc = j.new(NS, "(IIIIIII)V", *scores)
    nw = j.new(NW, f"(L{OBI};L{OBI};L{NS};)V", a, b, sc)
    strs = j.call_virtual(nw, "getAlignmentString", "()[Ljava/lang/String;")
    return nw, list(strs.a), j.call_virtual(nw, "getAlignmentScore", "()I")


def gen_nw(g):
    """a-5 / a-6: Needleman-Wunsch with the scores of NeedlemanParameters$OneSet, alignment strings, error metrics"""
    j = g.j
    rng = random.Random(303)
    out = {"jar": "TwoFourBitNucAcidLibraryMaven-1.0.jar + NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    s = g.section("NeedlemanWunsch(seq1 = pattern, seq2 = read slice, NeedlemanScores(-4,-5,-5,-5,-5,-5,+5)): getAlignmentString, "
                  "getAlignmentScore, DynamicProgramming.getScoreTable; Match.countErrorsInNeedleman (L31-34), "
                  "Match.hasN3pConsecutiveMatchesInNeedleman(nw, 6) (L41-50); NeedlemanMatch(match, pattern, read) statistics "
                  "(L68-196)", NW, "fillInCell L55-80, getTraceback L102-151")
    pats = ["CTTCCGATCT", "CTACACGACGCTCTTCCGATCT", "AAGCAGTGGTATCAAC", "AAGCAGTGGTATCAACGCAGAGTACAT", "TTTCTTATATGGG"]
    cases = []
    for p in pats:
        for t in range(14 if len(p) <= 16 else 8):
            mode = t % 7
            x = list(p)
            if mode == 0:
                pass
            else:
                k = 0
                while k < len(x):
                    r = rng.random()
                    if r < 0.06 * mode / 2:
                        x[k] = rng.choice("ACGTN")
                    elif r < 0.10 * mode / 2:
                        del x[k]
                        continue
                    elif r < 0.14 * mode / 2:
                        x.insert(k, rng.choice("ACGT"))
                        k += 1
                    k += 1
            rs = "".join(x)
            if mode == 6:
                rs = rnd_seq(rng, len(p))
            rs = (rs + rnd_seq(rng, len(p)))[:len(p)]  # the reference always aligns a slice of the pattern's length
            cases.append((p, rs))
    cases += [("CTTCCGATCT", "CTTCCGATCT"), ("CTTCCGATCT", "TTTTTTTTTT"), ("CTTCCGATCT", "NNNNNNNNNN"), ("CTTCCGATCT", "TCTAGCCTTC"),
              ("CTTCCGATCT", "CTTCGATCTA"), ("CTTCCGATCT", "ACTTCCGATC"), ("CTTCCGATCT", "TTCCGATCTA")]
    for p, rs in cases:
        nw, strs, score = nw_align(j, p, rs)
        tab = j.call_virtual(nw, "getScoreTable", "()[[I")
        err = j.call_lambda(j.get_static(MATCH, "countErrorsInNeedleman"), [nw]).v
        n3p = j.call_lambda(j.get_static(MATCH, "hasN3pConsecutiveMatchesInNeedleman"), [nw, JBox("java/lang/Integer", 6)]).v
        o = j.new(NM, "(Ljava/lang/String;Ljava/lang/String;Ljava/lang/String;)V", strs[0], strs[1], strs[2])
        c = {"pattern": p, "read_slice": rs, "alignment": strs, "score": score, "score_table_last_row": list(tab.a[-1].a),
             "count_errors": err, "has_6_3p_matches": int(n3p),
             "nm_nerrors": j.call_virtual(o, "getNerrorsNeedleman", "()I"),
             "nm_subs_del_ins": [o.f["substitutionsNeedleman"], o.f["deletionsNeedleman"], o.f["insertionsNeedleman"]],
             "nm_end_of_read_5": j.call_virtual(o, "countIndelsMismatchesEndOfRead", "(I)F", 5),
             "nm_consecutive": j.call_virtual(o, "getNconsecutiveMatchesNeedleman", "()I"),
             "nm_best_two": j.call_virtual(o, "getSumOfBestTwoMatchStretchesNeedleman", "()I"),
             "nm_offset_for_read_end": j.call_virtual(o, "getOffsetForReadEnd", "()I")}
        s["cases"].append(c)
    out["sections"].append(g.finish(s))
    return out


def gen_lev(g):
    """a-16 (inner): apachemod LevenshteinDistance.limitedCompare(byte[], byte[], threshold)"""
    j = g.j
    rng = random.Random(404)
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    s = g.section("limitedCompare(left, right, threshold) on 4-bit coded 12-mers (L220-283); -1 = above the threshold", LEV, "limitedCompare:([B[BI)I")
    codes = [1, 2, 4, 8]
    for t in range(400):
        n = 12 if t < 320 else rng.choice([5, 8, 10, 14])
        a = [rng.choice(codes) for _ in range(n)]
        b = list(a)
        for _ in range(rng.choice([0, 1, 1, 2, 2, 3, 4, 5, 8])):
            r = rng.random()
            k = rng.randrange(len(b)) if b else 0
            if r < 0.4 and b:
                b[k] = rng.choice(codes + [15])
            elif r < 0.7 and b:
                del b[k]
            else:
                b.insert(k, rng.choice(codes))
        if t < 320:
            b = (b + [rng.choice(codes) for _ in range(n)])[:n]
        th = 4 if t < 360 else rng.choice([1, 2, 3, 6])
        try:
            res = j.call_static(LEV, "limitedCompare", "([B[BI)I", j.byte_array(a), j.byte_array(b), th)
        except JavaThrow as e:  # unequal lengths further apart than the threshold: the reference's own code throws
            res = {"throws": e.obj.cls}
        s["cases"].append({"a": a, "b": b, "threshold": th, "out": res})
    out["sections"].append(g.finish(s))
    return out


COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp_str(q):
    return "".join(COMP[c] for c in reversed(q))


def membership_set(keys):
    """stands where the reference holds a fastutil LongOpenHashSet (jar absent): contains() only, no order"""
    o = JObject("it/unimi/dsi/fastutil/longs/LongOpenHashSet")
    o.native = set(keys)
    return o


def run_tester(j, read, ae, offset, wl, ed, three_p, length=16):
    """what Parser.lambda$assignBarcode$4 (Parser.java:L205-242) does for one offset, through the reference's constructors"""
    if three_p:
        bc_start, bc_end = ae - length + offset, ae - 1 + offset
    else:
        bc_start, bc_end = ae + 1 + offset, ae + length + offset
    bc = j.new(TB, "(Ljava/lang/String;)V", read[bc_start - 1:bc_end])
    if three_p:
        post = j.call_virtual(j.new(OB, "(Ljava/lang/CharSequence;)V", read[bc_start - 5:bc_start]), "reverseComplement", f"()L{OBI};")
        bc = j.call_virtual(bc, "reverseComplement", f"()L{TB};")
    else:
        post = j.new(OB, "(Ljava/lang/CharSequence;)V", read[bc_end:bc_end + 5])
    opt = j.call_native("java/util/Optional.of", [bc])
    t = j.new(BMT, f"(Ljava/util/Optional;IZZLjava/util/Set;SIL{OBI};Z)V", opt, ed, 0, 1, wl, offset, length, post, 1)
    m = j.call_virtual(t, "call", f"()L{BMT}$Matches;")
    if m is None:
        return None
    res = []
    for o, _ in m.native.items_in_insertion_order():
        res.append({"read_seq": u64(o.f["readSeq"]), "bc": u64(o.f["matchingBC"]), "ed": o.f["editDistance"], "subs": o.f["substitutions"],
                    "ins": o.f["insertions"], "dels": o.f["deletions"], "offset": o.f["offsetFromPredicted"],
                    "offset_for_read_end": j.call_virtual(o, "getOffsetForReadEnd", "()I")})
    return sorted(res, key=lambda r: (r["ed"], r["bc"], r["read_seq"]))


def mutate(rng, q, n_err, with_n=False):
    x = list(q)
    for _ in range(n_err):
        r = rng.random()
        k = rng.randrange(len(x))
        if r < 0.4:
            x[k] = rng.choice("ACGT" + ("N" if with_n else ""))
        elif r < 0.7:
            del x[k]
        else:
            x.insert(k, rng.choice("ACGT"))
    return "".join(x)


def gen_bcmatch(g, n_ed1=60, n_ed2=6):
    """a-11: BarcodeMatchTester.call() per offset, driven exactly as Parser.lambda$assignBarcode$4 drives it"""
    j = g.j
    rng = random.Random(505)
    enc = lambda q: j.call_static(TB, "getLongHashForSeq", "([C)J", j.char_array(q))  # noqa: E731
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    s = g.section("BarcodeMatchTester(Optional.of(bc), ed, false, true, searchSet, offset, 16, postBCseq, true).call() for the five offsets "
                  "0, -1, +1, -2, +2 of a stranded read (Parser.java:L205-242; doJob L198-247, substitutions L259-270, insertions "
                  "L286-297, deletions L313-352, checkMatchWithTestSets).  `matches` = the CONTENT of the returned HashSet, sorted "
                  "canonically here (its iteration order is not part of these vectors); null = call() returned null.  searchSet is a "
                  "membership-only stand-in for the absent fastutil set.", BMT, "call:()L...$Matches;")
    # barcode lists with near-duplicates so that 'first hit at a level wins' is exercised
    base = [rnd_seq(rng, 16) for _ in range(30)]
    near = []
    for b in base[:12]:
        for _ in range(3):
            near.append(mutate(rng, b, 1)[:16].ljust(16, "A"))
    bcs = sorted(set(base + near + ["A" * 16, "ACGT" * 4, "T" * 16]))
    keys = [enc(b) for b in bcs]
    wl = membership_set(keys)
    s["barcodes"] = bcs
    s["barcode_keys"] = [u64(k) for k in keys]
    plan = [(1, True)] * n_ed1 + [(1, False)] * (n_ed1 // 2) + [(0, True)] * 6 + [(2, True)] * n_ed2 + [(2, False)] * (n_ed2 // 2)
    for idx, (ed, three_p) in enumerate(plan):
        b = rng.choice(bcs)
        kind = idx % 6
        n_err = [0, 1, 1, 2, 2, 3][kind]
        left, right = rnd_seq(rng, 30), rnd_seq(rng, 30)
        if three_p:
            body = mutate(rng, revcomp_str(b), n_err, with_n=(idx % 11 == 0))
            read = left + body + right          # adapter would start at AE: barcode occupies [AE-16, AE-1]
            ae = len(left) + len(body) + 1 + rng.choice([0, 0, 0, -1, 1, 2, -2])
        else:
            body = mutate(rng, b, n_err, with_n=(idx % 11 == 0))
            read = left + body + right          # adapter ends at AE: barcode occupies [AE+1, AE+16]
            ae = len(left) + rng.choice([0, 0, 0, -1, 1, 2, -2])
        if idx % 17 == 5:
            read = read[:len(left)] + "A" * 24 + read[len(left) + 24:]   # homopolymer windows: identical mutants, dedup set at ed 2
        per_offset = {}
        for off in (0, -1, 1, -2, 2):
            per_offset[str(off)] = run_tester(j, read, ae, off, wl, ed, three_p)
        s["cases"].append({"read": read, "adapter_pos": ae, "ed": ed, "three_prime": three_p, "matches": per_offset})
        if idx % 10 == 0:
            print(f"  bcmatch {idx + 1}/{len(plan)}  {time.time() - g.t0:.0f}s", flush=True)
    out["sections"].append(g.finish(s))
    return out


def gen_polyat(g):
    """a-7: PolyATSearcher.findpolyAT"""
    j = g.j
    rng = random.Random(606)
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    s = g.section("new PolyATSearcher(read, 15, 0.75f, 150).findpolyAT(reverse) (L56-252): polyT in the first 175 bases (reverse = false) / "
                  "in the reverse complement of the last 175 (reverse = true); result null or (polyTbegin, polyTend, length of seqTilPolyAend)",
                  PS, "findpolyAT:(Z)L...$PolyAscanResult;")

    def run(read):
        res = []
        for rev in (0, 1):
            o = j.new(PS, "(Ljava/lang/String;IFI)V", read, 15, f32(0.75), 150)
            try:
                r = j.call_virtual(o, "findpolyAT", f"(Z)L{PS}$PolyAscanResult;", rev)
            except JavaThrow as e:  # reads shorter than the 175-base window: the callers test the read length first
                res.append({"throws": e.obj.cls})
                continue
            res.append(None if r is None else {"begin": r.f["polyTbegin"], "end": r.f["polyTend"], "seq_til_end_len": len(r.f["seqTilPolyAend"].f["naData"].a)})
        return res

    for t in range(70):
        kind = t % 7
        tl = rng.randrange(8, 60)
        run_t = "".join("T" if rng.random() > (0.0, 0.05, 0.12, 0.2, 0.3, 0.1, 0.1)[kind] else rng.choice("ACG") for _ in range(tl))
        pre = rnd_seq(rng, rng.randrange(0, 140) if kind != 5 else rng.randrange(140, 175))
        body = rnd_seq(rng, rng.randrange(150, 500))
        tail_a = "".join("A" if rng.random() > 0.08 else rng.choice("CGT") for _ in range(rng.randrange(10, 50)))
        read = pre + run_t + body + (tail_a + rnd_seq(rng, rng.randrange(20, 60)) if kind in (2, 4, 6) else "")
        if kind == 6:
            read = read[:rng.randrange(100, 170)]
        s["cases"].append({"read": read, "forward_and_reverse": run(read)})
    out["sections"].append(g.finish(s))
    return out


POLYA_PARAM_SETS = ((12, 0.8, 100), (20, 0.7, 140), (15, 0.75, 120), (10, 0.9, 150), (15, 0.6, 150), (18, 0.75, 147), (30, 0.75, 135), (5, 1.0, 60))


def gen_polyat_params(g):
    """a-7 under `scanfastq -p <length> -f <fraction> -w <window>` (NanoporeReadScannerMain.java:L227-234): the finder with other parameters than
    config.xml's 15 / 0.75 / 150 -- the sets tests/test_scan_gpu.py::test_other_polya_windows_equal_oracle runs the kernels with"""
    j = g.j
    rng = random.Random(616)
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    for ml, frac, win in POLYA_PARAM_SETS:
        s = g.section(f"new PolyATSearcher(read, {ml}, {frac}f, {win}).findpolyAT(reverse) (L56-252): polyT in the first window + length + 10 bases (reverse = false) / "
                      "in the reverse complement of the last ones (reverse = true); result null or (polyTbegin, polyTend, length of seqTilPolyAend)",
                      PS, "findpolyAT:(Z)L...$PolyAscanResult;")
        s["polya_len"], s["polya_frac"], s["window_polya"] = ml, frac, win
        sub_n = win + ml + 10
        for t in range(36):
            kind = t % 6
            tl = rng.randrange(max(4, ml // 2), 4 * ml)
            run_t = "".join("T" if rng.random() > (0.0, 0.05, 0.12, 0.2, 0.3, 0.1)[kind] else rng.choice("ACG") for _ in range(tl))
            pre = rnd_seq(rng, rng.randrange(0, max(1, win - 10)) if kind != 5 else rng.randrange(max(1, win - 10), sub_n))
            body = rnd_seq(rng, rng.randrange(sub_n, 2 * sub_n + 100))
            tail_a = "".join("A" if rng.random() > 0.08 else rng.choice("CGT") for _ in range(rng.randrange(ml - 3, 4 * ml)))
            read = pre + run_t + body + (tail_a + rnd_seq(rng, rng.randrange(10, 60)) if kind in (2, 4) else "")
            res = []
            for rev in (0, 1):
                o = j.new(PS, "(Ljava/lang/String;IFI)V", read, ml, f32(frac), win)
                try:
                    r = j.call_virtual(o, "findpolyAT", f"(Z)L{PS}$PolyAscanResult;", rev)
                except JavaThrow as e:
                    res.append({"throws": e.obj.cls})
                    continue
                res.append(None if r is None else {"begin": r.f["polyTbegin"], "end": r.f["polyTend"], "seq_til_end_len": len(r.f["seqTilPolyAend"].f["naData"].a)})
            s["cases"].append({"read": read, "forward_and_reverse": res})
        out["sections"].append(g.finish(s))
    return out

# ---------------------------------------------------------------------------------------------------------------------
# Whole records through the reference's pass 2: ChimeraFindernew.findSplitPositions -> PolyATadapterAnalyzer_{3p,5p}BCUMI.search
# -> Parser.assignBarcode -> ReadFlags$Flags.finalizeFlag -> FastqRecordExt.getRecordForWriting, i.e. what Parser.call /
# processOneRecord and the writer thread do per record (Parser.java:L92-124, L132-185; FastqWriterThreadPool.java:L300-306).
# The driver below only builds the objects those methods take (parameters from Jar/config.xml through tools/ref_params.py,
# the command-line fields NanoporeReadScannerMain sets, a barcode map) and calls them; statistics objects are left out.
# ---------------------------------------------------------------------------------------------------------------------
PAR = "com/rw/nanoporereadscanner/parameters/ParametersReadScannerApp"
SCANTYPE = "com/rw/parameters/ParametersMainBase$SCANTYPE"
GOPT = "com/google/common/base/Optional"
MAIN = "com/rw/nanoporereadscanner/NanoporeReadScannerMain"
FQR = "htsjdk/samtools/fastq/FastqRecord"
FQX = "com/rw/nanoporereadscanner/readerwriter/FastqRecordExt"
PARSER = "com/rw/nanoporereadscanner/analyzers/Parser"
BCMAP = "com/rw/nanoporereadscanner/WorkerReadscanner$BarcodesMapForBCfinding"
CRANK = "com/rw/nanoporereadscanner/WorkerReadscanner$CountsRank"
KMERS = "com/rw/parameters/TSO_AdapterParameterBase$NucleicAcidInmutableOneBytePerBaseKmers"
FLAGS = "com/rw/nanoporereadscanner/stats/ReadFlags$Flags"
L2O = "it/unimi/dsi/fastutil/longs/Long2ObjectOpenHashMap"


def install_long2object(j):
    """membership / lookup stand-in for the absent fastutil Long2ObjectOpenHashMap (superclass of BarcodesMapForBCfinding):
    get / put / containsKey / keySet().contains; no iteration"""
    N = j.natives

    def store(o):
        if o.native is None:
            o.native = {}
        return o.native

    N[L2O + ".<init>"] = lambda jj, o, *a: store(o) and None
    N[L2O + ".put:(JLjava/lang/Object;)Ljava/lang/Object;"] = lambda jj, o, k, v: store(o).__setitem__(k, v)
    N[L2O + ".get:(J)Ljava/lang/Object;"] = lambda jj, o, k: store(o).get(k)
    N[L2O + ".containsKey:(J)Z"] = lambda jj, o, k: 1 if k in store(o) else 0
    N[L2O + ".size"] = lambda jj, o: len(store(o))

    def key_set(jj, o):
        ks = JObject("it/unimi/dsi/fastutil/longs/LongOpenHashSet")
        ks.native = store(o).keys()  # live view: contains only
        return ks

    N[L2O + ".keySet"] = key_set
    import jvm_exec

    jvm_exec.JDK_SUPER[L2O] = "java/lang/Object"
    jvm_exec.JDK_IFACES["it/unimi/dsi/fastutil/longs/LongOpenHashSet"] = ["java/util/Set", "it/unimi/dsi/fastutil/longs/LongSet"]


def config_with(knobs):
    """a copy of the reference's Jar/config.xml (scratch, /tmp) with the texts of the elements `section/knob` replaced -> its path.  The file the
    reference would be started with; nothing of it is kept (the fixture records the replaced knobs only)."""
    import tempfile
    import xml.etree.ElementTree as ET

    tree = ET.parse(REF + "config.xml")
    root = tree.getroot()
    for name, text in knobs.items():
        sec, leaf = name.split("/")
        el = root.find(sec).find(leaf)
        if el is None:
            raise KeyError(name)
        el.text = str(text)
    fd, path = tempfile.mkstemp(suffix=".xml", prefix="ref_config_")
    os.close(fd)
    tree.write(path)
    return path


class Pass2:
    def __init__(self, g, five_prime, ed, dont_search_polya=False, knobs=None):
        import ref_params

        self.j = j = g.j
        self.five = five_prime
        install_long2object(j)
        if knobs:
            path = config_with(knobs)
            try:
                par, self.report = ref_params.load_config(j, PAR, path)
            finally:
                os.unlink(path)
        else:
            par, self.report = ref_params.load_config(j, PAR)
        rs = par.f["readScannerParameters"]
        rs.f["assignCellBCwithEditDistance"] = j.call_static(GOPT, "of", f"(Ljava/lang/Object;)L{GOPT};", JBox("java/lang/Integer", ed))
        par.f["polyAT"].f["dontSearchPolyAFor5pBarcoding"] = 1 if dont_search_polya else 0   # -y (NanoporeReadScannerMain.java:L248)
        par.f["scantype"] = j.get_static(SCANTYPE, "FIVEP_BARCODE" if five_prime else "THREEP_BARCODE")   # -h (L249)
        for k in (("adapter_5p_for5pBarcoding", "adapter_3p_for5pBarcoding", "tso_for5pBarcoding") if five_prime else
                  ("adapter_for3pBarcoding", "tso_for3pBarcoding")):
            o = par.f[k]
            j.invoke(j.find_method(o.cls, "validate", "()V"), [o])        # what ParametersReadScannerApp.validate() L107-130 calls
        jc = j.load(MAIN)
        jc.initialized = True          # the program's own static initialiser only builds the CLI option table
        jc.statics["params"] = par     # PolyATadapterAnalyzer_* are constructed from NanoporeReadScannerMain.params (Parser.java:L96)
        self.par = par
        self.ap = par.f["adapter_5p_for5pBarcoding" if five_prime else "adapter_for3pBarcoding"]
        self.max_mm = self.ap.f["maxNeedlemanMismatches"].v

    def barcode_map(self, keys_ranks):
        j = self.j
        m = j.new_object(BCMAP)
        m.native = {}
        for k, rank in keys_ranks:
            cr = j.new(CRANK, "(Ljava/lang/Integer;Ljava/lang/Integer;)V", JBox("java/lang/Integer", 1), JBox("java/lang/Integer", rank))
            m.native[k] = cr
        return m

    def parser(self, bcmap):
        j = self.j
        p = j.new_object(PARSER)      # fields as Parser.<init> L70-78 assigns them; no chunk / statistics
        p.f["parameters"] = self.par
        p.f["hashMapForBCfinding"] = bcmap
        p.f["assignedBarcodes2ndPass"] = j.natives["java/util/HashMap.<new>"](j)
        p.f["pass"] = j.get_static(MAIN + "$Pass", "SECOND")
        return p

    def record(self, name, seq, qual):
        j = self.j
        rec = j.new(FQR, "(Ljava/lang/String;Ljava/lang/String;Ljava/lang/String;Ljava/lang/String;)V", name, seq, "", qual)
        return j.new(FQX, f"(L{FQR};)V", rec)

    def process(self, parser, fq, pass2=True):
        """Parser.processOneRecord L92-114 without the statistics"""
        j = self.j
        failed = j.call_virtual(j.get_static(FLAGS, "FAILED"), "getValue", "()J")
        sr = fq.f["scanResult"]
        if (sr.f["flag"] & failed) == 0:
            cls = "com/rw/nanoporereadscanner/analyzers/PolyATadapterAnalyzer_" + ("5pBCUMI" if self.five else "3pBCUMI")
            pa = j.new(cls, f"(L{PAR};)V", self.par)
            kmers = self.ap.f["bytesequence"] if pass2 else j.call_virtual(self.ap.f["bytesequence_complete"], "get", "()Ljava/lang/Object;")
            j.call_virtual(pa, "search", f"(L{FQX};L{KMERS};I)V", fq, kmers, self.max_mm + (1 if self.five else 0))
            if pass2 and j.call_virtual(sr, "adapterFound", "()Z"):
                j.invoke(parser.cls and j.load(PARSER).methods[("assignBarcode", f"(L{FQX};)V")], [parser, fq])
        sr.f["flag"] = j.call_static(FLAGS, "finalizeFlag", "(J)J", sr.f["flag"])
        return fq

    def describe(self, fq, read_id):
        j = self.j
        sr = fq.f["scanResult"]

        def iv(o, k):
            v = None if o is None else o.f.get(k)
            if isinstance(v, JObject) and isinstance(v.native, tuple):   # java.util.Optional
                v = v.native[0]
            if isinstance(v, JObject) and "Optional" in v.cls:            # com.google.common.base.Optional
                v = j.call_virtual(v, "orNull", "()Ljava/lang/Object;")
            return None if v is None else (v.v if isinstance(v, JBox) else v)

        ad, pa, tso, bc = sr.f["adapter_result"], sr.f["polyA_Result"], sr.f["tSOresult"], sr.f["barcode_Result"]
        passed = bool(j.call_virtual(fq, "passed", "()Z"))
        rec = j.call_virtual(fq, "getRecordForWriting", f"(Lcom/rw/parameters/ReadScannerParameters;ZLjava/lang/Integer;)L{FQR};",
                             self.par.f["readScannerParameters"], 1 if self.five else 0, JBox("java/lang/Integer", read_id) if passed else None)
        out = {"flag": u64(sr.f["flag"]), "forward": sr.f["forward"].f["$name"], "passed": passed,
               "adapter": None if ad is None else [iv(ad, "start"), iv(ad, "end")],
               "polya": None if pa is None else [iv(pa, "start"), iv(pa, "end")],
               "tso": None if tso is None else [iv(tso, "start"), iv(tso, "end")],
               "barcode": None,
               "written": {"name": rec.f["readName"], "bases": rec.f["readString"], "quality_header": rec.f["qualityHeader"],
                           "qualities": rec.f["baseQualityString"]}}
        if bc is not None and bc.f.get("barcodeseq") is not None:
            out["barcode"] = {"seq": j.call_virtual(bc.f["barcodeseq"], "toString", "()Ljava/lang/String;"), "ed": iv(bc, "editDistance"),
                              "ed_second": iv(bc, "editDistanceSecondBest"), "start": iv(bc, "start"), "end": iv(bc, "end"), "rank": iv(bc, "rank")}
        return out


TSO = "AAGCAGTGGTATCAACGCAGAGTACATGGG"
AD3 = "CTACACGACGCTCTTCCGATCT"


def noisy(rng, q, rate):
    out = []
    for c in q:
        r = rng.random()
        if r < rate * 0.4:
            out.append(rng.choice("ACGT"))
        elif r < rate * 0.7:
            continue
        elif r < rate:
            out.append(c)
            out.append(rng.choice("ACGT"))
        else:
            out.append(c)
    return "".join(out)


def synth_read(rng, bcs, five_prime, kind):
    """one synthetic read; kind varies error rate / strand / structure (SURVEY 8d layout)"""
    bc, umi = rng.choice(bcs), rnd_seq(rng, 12)
    cdna = rnd_seq(rng, rng.randrange(260, 700))
    rate = (0.0, 0.02, 0.05, 0.08, 0.12)[kind % 5]
    if five_prime:
        mol = AD3 + bc + umi + "TTTCTTATATGGG" + cdna + ("A" * rng.randrange(18, 40) if kind % 3 else "") + rnd_seq(rng, rng.randrange(0, 30))
    else:
        mol = TSO + cdna + "A" * rng.randrange(18, 45) + revcomp_str(umi) + revcomp_str(bc) + revcomp_str(AD3)
    if kind % 7 == 6:                       # no adapter at all
        mol = rnd_seq(rng, len(mol))
    mol = noisy(rng, mol, rate)
    if kind % 11 == 10:
        mol = mol[:rng.randrange(120, 199)]  # below minReadLength
    if kind % 13 == 7:
        k = rng.randrange(len(mol))
        mol = mol[:k] + "N" + mol[k + 1:]
    if rng.random() < 0.5:
        mol = revcomp_str(mol)
    qual = "".join(chr(33 + rng.randrange(3, 35)) for _ in mol)
    return mol, qual, bc


def gen_pass2(g, n_reads=36, five_prime=False, ed=1, seed=707, dont_search_polya=False, tag="3p"):
    j = g.j
    rng = random.Random(seed)
    p2 = Pass2(g, five_prime, ed, dont_search_polya)
    enc = lambda q: j.call_static(TB, "getLongHashForSeq", "([C)J", j.char_array(q))  # noqa: E731
    bcs = sorted({rnd_seq(rng, 16) for _ in range(40)})
    extra = [mutate(rng, b, 1)[:16].ljust(16, "C") for b in bcs[:10]]
    bcs = sorted(set(bcs + extra))
    ranks = {b: k + 1 for k, b in enumerate(bcs)}
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar (+ TwoFourBitNucAcidLibraryMaven-1.0.jar, htsjdk-4.1.3.jar, guava, commons-lang3 as shipped in Jar/lib)",
           "sections": []}
    s = g.section(f"pass 2 of scanfastq per record, {'5-prime' if five_prime else '3-prime'} barcoding, --bcEditDistance {ed}"
                  + (", --noPolyARequired" if dont_search_polya else "") +
                  ": PolyATadapterAnalyzer.search -> Parser.assignBarcode -> ReadFlags$Flags.finalizeFlag -> FastqRecordExt.getRecordForWriting "
                  "(read id = 1 + index of the case for passed records).  Parameters: Jar/config.xml as shipped.  Each case ran under three different "
                  "iteration orders of java.util.HashMap / HashSet; `hash_orders_agree` = all three gave this result.",
                  PARSER, "processOneRecord L92-114 / assignBarcode L195-315 / getRecordForWriting L209-311")
    s["barcodes"] = bcs
    s["ranks"] = [ranks[b] for b in bcs]
    s["five_prime"], s["ed"], s["dont_search_polya"] = five_prime, ed, dont_search_polya
    s["config_report"] = p2.report
    vals = j.call_static(FLAGS, "values", f"()[L{FLAGS};")
    s["flag_values"] = {v.f["$name"]: u64(j.call_virtual(v, "getValue", "()J")) for v in vals.a}   # ReadFlags$Flags.getValue()
    for idx in range(n_reads):
        seq, qual, bc = synth_read(rng, bcs, five_prime, idx)
        name = f"read{idx:04d} runid=abc ch={idx % 512}"
        results = []
        for order in ("insertion", "reverse", ("shuffle", idx + 1)):
            j.hash_order = order
            bcmap = p2.barcode_map([(enc(b), ranks[b]) for b in bcs])
            parser = p2.parser(bcmap)
            fq = p2.record(name, seq, qual)
            try:
                p2.process(parser, fq)
                results.append(p2.describe(fq, idx + 1))
            except JavaThrow as e:
                results.append({"throws": e.obj.cls, "message": e.obj.f.get("message"), "in": e.trace[:6]})
        j.hash_order = None
        agree = all(r == results[0] for r in results[1:])
        s["cases"].append({"name": name, "seq": seq, "qual": qual, "planted_barcode": bc, "hash_orders_agree": agree, "result": results[0]})
        if idx % 6 == 0:
            print(f"  pass2[{tag}] {idx + 1}/{n_reads}  {time.time() - g.t0:.0f}s", flush=True)
    out["sections"].append(g.finish(s))
    return out


def gen_pass2_3p(g):
    return gen_pass2(g, 40, False, 1, 707, tag="3p")


def gen_pass2_3p_ed2(g):
    return gen_pass2(g, 8, False, 2, 717, tag="3p ed2")


def gen_pass2_5p(g):
    return gen_pass2(g, 30, True, 1, 727, dont_search_polya=True, tag="5p -y")


def gen_pass2_5p_polya(g):
    return gen_pass2(g, 14, True, 1, 737, dont_search_polya=False, tag="5p")

# ---------------------------------------------------------------------------------------------------------------------
# Whole chunks through Parser.call ITSELF (round 4): ReadChunk of five records -> ChimeraFindernew.findSplitPositions ->
# processOneRecord per fragment (search, assignBarcode, finalizeFlag, the chunk's statistics) -> getRecordForWriting.
# 100 chunks = 500 input reads per configuration, the reads built to reach the branches SURVEY 8a names (`kind`).
# The chunks of a section are independent: with WIDE_JOBS > 1 they are spread over processes (one JVM each); a read is a
# function of (seed, its index) only, so the fixture does not depend on how the chunks were dealt.
# ---------------------------------------------------------------------------------------------------------------------
RCHUNK = "com/rw/nanoporereadscanner/readerwriter/FastqFileReader$ReadChunk"
WHAT_TODO = "com/rw/parameters/ReadScannerParameters$WHAT_TODO"
WIDE_KINDS = ["clean", "err2", "err5", "err8", "err12", "random", "length_edge", "n_in_barcode", "both_sides", "terminal6", "junction_indel",
              "ambiguous_bc", "tso_damaged", "no_tso", "short_polya", "no_polya", "chimera2", "chimera3", "chimera4", "internal_site",
              "adapter_truncated", "low_complexity", "t_rich_umi", "err15"]
TSO16 = "AACGCAGAGTACATGG"


def wide_barcodes(seed):
    """60 random 16-mers + for ten of them a partner two substitutions away (a read half way between the two is ed 1 from both:
    best.ed == second.ed, Parser.java:L252) + for five a partner one deletion away"""
    rng = random.Random(seed)
    base = sorted({rnd_seq(rng, 16) for _ in range(60)})
    pairs = []
    for b in base[:10]:
        i, k = rng.sample(range(16), 2)
        m = list(b)
        for q in (i, k):
            m[q] = rng.choice([c for c in "ACGT" if c != m[q]])
        pairs.append((b, "".join(m), i, k))
    indel = []
    for b in base[10:15]:
        i = rng.randrange(2, 14)
        indel.append(b[:i] + b[i + 1:] + rng.choice("ACGT"))
    bcs = sorted(set(base + [p[1] for p in pairs] + indel))
    return bcs, pairs


def wide_read(seed, idx, bcs, pairs, five_prime, can_split):
    """input read idx of a wide section: (name, bases, qualities, kind)"""
    rng = random.Random(seed * 1000003 + idx)
    kind = WIDE_KINDS[idx % len(WIDE_KINDS)]
    variant = idx // len(WIDE_KINDS)
    rate = {"clean": 0.0, "err2": 0.02, "err5": 0.05, "err8": 0.08, "err12": 0.12, "err15": 0.15}.get(kind, (0.0, 0.02, 0.04)[variant % 3])

    def molecule(bc=None, umi=None, tso=TSO, ad=AD3, pa=None, cdna=None, junction=""):
        bc = bc or rng.choice(bcs)
        umi = umi or rnd_seq(rng, 12)
        cdna = cdna if cdna is not None else rnd_seq(rng, rng.randrange(110, 360))
        pa = pa if pa is not None else "A" * rng.randrange(18, 45)
        if five_prime:   # adapter - BC - UMI - TSO tail - cDNA [- polyA - rc(3' adapter)]
            return ad + junction + bc + umi + "TTTCTTATATGGG" + cdna + pa + (revcomp_str("AAGCAGTGGTATCAACGCAGAGTAC") if variant % 2 else rnd_seq(rng, rng.randrange(0, 25)))
        return tso + cdna + pa + revcomp_str(umi) + revcomp_str(bc) + revcomp_str(junction) + revcomp_str(ad)

    if kind in ("clean", "err2", "err5", "err8", "err12", "err15"):
        mol = molecule()
    elif kind == "random":
        mol = rnd_seq(rng, rng.randrange(200, 600))
    elif kind == "length_edge":     # minReadLength 200 (PolyATadapterAnalyzerBase.java:L131-137), the splitter's 2 * 70 + 100 (ChimeraFindernew.java:L169)
        want = (199, 200, 201, 239, 240, 241)[variant % 6]
        mol = molecule(cdna=rnd_seq(rng, 300))
        mol = mol[:want] if five_prime else mol[-want:]
    elif kind == "n_in_barcode":
        bc = list(rng.choice(bcs))
        for _ in range(1 + variant % 2):
            bc[rng.randrange(16)] = "N"
        mol = molecule(bc="".join(bc))
        if variant % 3 == 2:
            k = rng.randrange(len(mol))
            mol = mol[:k] + "N" + mol[k + 1:]
    elif kind == "both_sides":      # adapter structure at both ends: |delta| < 2 fails, else the side with fewer errors (L145-221)
        m1 = molecule(cdna=rnd_seq(rng, 150))
        if five_prime:
            head = AD3 + rng.choice(bcs) + rnd_seq(rng, 12) + "TTTCTTATATGGG"
            mol = noisy(rng, head, (0.0, 0.03, 0.1)[variant % 3]) + rnd_seq(rng, 200) + revcomp_str(noisy(rng, head, (0.0, 0.1, 0.03)[variant % 3]))
        else:
            tail = "A" * rng.randrange(20, 40) + revcomp_str(rnd_seq(rng, 12)) + revcomp_str(rng.choice(bcs)) + revcomp_str(AD3)
            mol = revcomp_str(noisy(rng, tail, (0.0, 0.03, 0.1)[variant % 3])) + rnd_seq(rng, 200) + noisy(rng, tail, (0.0, 0.1, 0.03)[variant % 3])
        del m1
    elif kind == "terminal6":       # > maxNeedlemanMismatches errors but the last 6 alignment columns match (MIN_3P_CONSEC_MATCHES_TO_OVERRIDE_PASS)
        n_bad = 4 + variant % 2
        lo = len(AD3) - 10 if not five_prime else len(AD3) - 12
        ad = list(AD3)
        for q in range(lo, lo + n_bad):
            ad[q] = rng.choice([c for c in "ACGT" if c != ad[q]])
        mol = molecule(ad="".join(ad))
    elif kind == "junction_indel":  # a base too many / too few between adapter and barcode: the offset +-1 / +-2 windows win
        v = variant % 4
        if v == 0:
            mol = molecule(junction=rng.choice("ACGT"))
        elif v == 1:
            mol = molecule(junction=rnd_seq(rng, 2))
        elif v == 2:
            mol = molecule(ad=AD3[:-1])
        else:
            mol = molecule(ad=AD3[:-2])
    elif kind == "ambiguous_bc":    # one substitution away from two listed barcodes: best.ed == second.ed -> no barcode
        a, b, i, k = pairs[variant % len(pairs)]
        m = list(a)
        m[i] = b[i]
        mol = molecule(bc="".join(m))
    elif kind == "tso_damaged":     # TSO with > 5 errors: the rescues by 8 consecutive matches / two stretches >= 12 (L122-190)
        t = list(TSO)
        i0 = TSO.index(TSO16)
        keep = (range(i0 + 3, i0 + 12), range(i0, i0 + 7), list(range(i0, i0 + 6)) + list(range(i0 + 9, i0 + 16)), range(i0 + 8, i0 + 16))[variant % 4]
        for q in range(len(t)):
            if q not in keep:
                t[q] = rng.choice([c for c in "ACGT" if c != t[q]])
        mol = molecule(tso="".join(t))
    elif kind == "no_tso":
        mol = molecule(tso="")
    elif kind == "short_polya":
        mol = molecule(pa="A" * (8 + variant % 7))
    elif kind == "no_polya":
        mol = molecule(pa="")
    elif kind in ("chimera2", "chimera3", "chimera4"):
        k = int(kind[-1])
        parts = [molecule(cdna=rnd_seq(rng, rng.randrange(110, 220))) for _ in range(k)]
        for q in range(k):
            if (variant >> q) & 1:
                parts[q] = revcomp_str(parts[q])
        mol = "".join(parts)
    elif kind == "internal_site":   # internal polyA + adapter (+ TSO) inside one long cDNA
        a = rnd_seq(rng, 330)
        site = "A" * (25 + variant % 15) + revcomp_str(rnd_seq(rng, 12)) + revcomp_str(rng.choice(bcs)) + revcomp_str(AD3)
        if variant % 2:
            site += rnd_seq(rng, 30) + TSO
        if variant % 4 >= 2:
            site = revcomp_str(site)
        mol = molecule(cdna=a[:160] + site + a[160:])
    elif kind == "adapter_truncated":
        mol = molecule()
        cut = 3 + variant % 9
        mol = mol[cut:] if five_prime else mol[:-cut]
    elif kind == "low_complexity":
        unit = (("A", "AT", "AAG", "ACGT", "T")[variant % 5])
        mol = molecule(cdna=(unit * 400)[:rng.randrange(150, 320)])
    else:                           # t_rich_umi
        mol = molecule(umi="A" * (6 + variant % 6) + rnd_seq(rng, 6 - variant % 6))   # rc(umi) next to the polyA: T/A-rich window at the barcode side
    mol = noisy(rng, mol, rate)
    if kind == "length_edge":       # the edge lengths exactly, noise or not
        want = (199, 200, 201, 239, 240, 241)[variant % 6]
        while len(mol) < want:
            mol = (mol + "C") if five_prime else ("C" + mol)
        mol = mol[:want] if five_prime else mol[-want:]
    if rng.random() < 0.5:
        mol = revcomp_str(mol)
    qual = "".join(chr(33 + rng.randrange(3, 35)) for _ in mol)
    return f"w{idx:04d} runid=r4 ch={idx % 512}", mol, qual, kind


def wide_chunk(g, p2, bc_keys_ranks, reads, first_id):
    """Parser.call on one ReadChunk, once per hash order -> (records per order)"""
    j = g.j
    results = []
    for order in ("insertion", "reverse"):
        j.hash_order = order
        bcmap = p2.barcode_map(bc_keys_ranks)
        al = j.natives["java/util/ArrayList.<new>"](j)
        for name, seq, qual, _kind in reads:
            al.native.append(p2.record(name, seq, qual))
        chunk = j.new(RCHUNK, "(Ljava/util/Optional;ZLjava/util/List;)V", j.natives["java/util/Optional.empty"](j), 0, al)
        parser = j.new(PARSER, f"(L{RCHUNK};L{PAR};L{BCMAP};Ljava/util/Map;L{MAIN}$Pass;)V", chunk, p2.par, bcmap,
                       j.natives["java/util/HashMap.<new>"](j), j.get_static(MAIN + "$Pass", "SECOND"))
        try:
            out = j.call_virtual(parser, "call", f"()L{RCHUNK};")
            recs, rid = [], first_id
            for fq in out.f["fastqRecords"].native:
                passed = bool(j.call_virtual(fq, "passed", "()Z"))
                recs.append(p2.describe(fq, rid))
                rid += passed
            rf = chunk.f["stats"].f["readflags"]
            results.append({"records": recs, "sum_read_length_passed": rf.f["sumReadLengthPassed"].native[0] if hasattr(rf.f["sumReadLengthPassed"], "native") and isinstance(rf.f["sumReadLengthPassed"].native, list) else None})
        except JavaThrow as e:
            results.append({"throws": e.obj.cls, "message": e.obj.f.get("message"), "in": e.trace[:6]})
    j.hash_order = None
    return results


def _wide_worker(args):
    five_prime, ed, dont_search_polya, seed, chunk_ids, per_chunk = args[:6]
    trim, reader = (args[6], args[7]) if len(args) > 6 else (False, None)
    polya = args[8] if len(args) > 8 else None
    knobs = args[9] if len(args) > 9 else None
    reader = reader or wide_read
    g = Gen()
    j = g.j
    p2 = Pass2(g, five_prime, ed, dont_search_polya, knobs=knobs)
    if polya is not None:      # -p / -f / -w as NanoporeReadScannerMain.java:L228-234 stores them (boxed, in params.polyAT)
        pat = p2.par.f["polyAT"]
        pat.f["polyATlength"] = JBox("java/lang/Integer", int(polya[0]))
        pat.f["fractionATInPolyAT"] = JBox("java/lang/Float", f32(polya[1]))
        pat.f["windowSearchForPolyA"] = JBox("java/lang/Integer", int(polya[2]))
    rs = p2.par.f["readScannerParameters"]
    j.call_virtual(rs.f["what_todo"], "add", "(Ljava/lang/Object;)Z", j.get_static(WHAT_TODO, "FIND_BARCODES"))   # -b (ReadScannerParameters.java:L291)
    if trim:
        rs.f["trimFastq"] = 1                                                                                       # -u (NanoporeReadScannerMain.java cli_otions)
    bcs, pairs = wide_barcodes(seed)
    enc = lambda q: j.call_static(TB, "getLongHashForSeq", "([C)J", j.char_array(q))  # noqa: E731
    keys_ranks = [(enc(b), k + 1) for k, b in enumerate(bcs)]
    can_split = not dont_search_polya
    cases = []
    for c in chunk_ids:
        reads = [reader(seed, c * per_chunk + k, bcs, pairs, five_prime, can_split) for k in range(per_chunk)]
        res = wide_chunk(g, p2, keys_ranks, reads, first_id=1000 * c + 1)
        cases.append({"chunk": c, "first_read_id": 1000 * c + 1, "reads": [{"name": n, "seq": s, "qual": q, "kind": k} for n, s, q, k in reads],
                      "hash_orders_agree": all(r == res[0] for r in res[1:]), "result": res[0]})
        print(f"  pass2w chunk {c} done  {time.time() - g.t0:.0f}s", flush=True)
    vals = j.call_static(FLAGS, "values", f"()[L{FLAGS};")
    flag_values = {v.f["$name"]: u64(j.call_virtual(v, "getValue", "()J")) for v in vals.a}
    return cases, sorted(j.natives_used), g.hits(), j.steps, flag_values, p2.report


def gen_pass2w(g, five_prime, ed, dont_search_polya, seed, n_chunks=100, per_chunk=5, trim=False, reader=None, note="", polya=None, knobs=None):
    jobs = max(1, int(os.environ.get("WIDE_JOBS", "1")))
    ids = list(range(n_chunks))
    blocks = [ids[k::jobs] for k in range(jobs)]
    args = [(five_prime, ed, dont_search_polya, seed, b, per_chunk, trim, reader, polya, knobs) for b in blocks if b]
    if len(args) == 1:
        parts = [_wide_worker(args[0])]
    else:
        import multiprocessing as mp

        with mp.get_context("fork").Pool(len(args)) as pool:
            parts = pool.map(_wide_worker, args)
    cases = sorted((c for p in parts for c in p[0]), key=lambda c: c["chunk"])
    natives = sorted(set(n for p in parts for n in p[1]))
    hits = {}
    for p in parts:
        merge_hits(hits, p[2])
    bcs, _pairs = wide_barcodes(seed)
    s = {"reference_class": PARSER, "reference_method": "call L132-185 (-> ChimeraFindernew.findSplitPositions, processOneRecord L92-124, assignBarcode L195-315) / "
         "FastqRecordExt.getRecordForWriting L209-311",
         "title": f"pass 2 of scanfastq per CHUNK through Parser.call itself, {'5-prime' if five_prime else '3-prime'} barcoding, --bcEditDistance {ed}"
                  + (", --noPolyARequired" if dont_search_polya else "") + (", --trimfastq" if trim else "") + note + ": a FastqFileReader$ReadChunk of the five reads of a case -> the records Parser.call leaves in "
                  "chunk.fastqRecords (fragments of split reads included), each described as in ref_exec_pass2_*.json; passed records take read ids first_read_id, "
                  "first_read_id + 1, ... in list order.  Parameters: Jar/config.xml as shipped + what_todo = {FIND_BARCODES}.  Each chunk ran under two iteration orders "
                  "of java.util.HashMap / HashSet; `hash_orders_agree` = both gave this result.",
         "cases": cases, "barcodes": bcs, "ranks": list(range(1, len(bcs) + 1)), "five_prime": five_prime, "ed": ed, "dont_search_polya": dont_search_polya,
         "split_chimeras": not dont_search_polya, "trim_fastq": bool(trim), "polya": None if polya is None else [int(polya[0]), float(polya[1]), int(polya[2])],
         "knobs": None if not knobs else {k: str(v) for k, v in knobs.items()}, "kinds": sorted({r["kind"] for c in cases for r in c["reads"]}) if reader else WIDE_KINDS,
         "flag_values": parts[0][4], "config_report": parts[0][5],
         "natives": [{"native": k, "tier": jvm_natives.tier_of(k)} for k in natives]}
    s["max_tier"] = max([n["tier"] for n in s["natives"]] or ["A"])
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar (+ TwoFourBitNucAcidLibraryMaven-1.0.jar, htsjdk-4.1.3.jar, guava, commons-lang3 as shipped in Jar/lib)",
           "sections": [s], "_steps": sum(p[3] for p in parts), "_hits": hits}
    return out


X_KINDS = ["close_splits", "close_splits_rev", "internal_adapter_6err", "internal_adapter_7err", "internal_adapter_8err"]


def targeted_read(seed, idx, bcs, pairs, five_prime, can_split):
    """reads built for single branches of the splitter that the wide set does not reach (3' barcoding)"""
    rng = random.Random(seed * 1000003 + idx)
    kind = X_KINDS[idx % len(X_KINDS)]
    variant = idx // len(X_KINDS)

    def molecule(cdna_len, tso=TSO):
        return tso + rnd_seq(rng, cdna_len) + "A" * rng.randrange(22, 40) + revcomp_str(rnd_seq(rng, 12)) + revcomp_str(rng.choice(bcs)) + revcomp_str(AD3)

    if kind.startswith("close_splits"):
        # junction (polyA + adapter | TSO: a pair, split at its middle) and, 40 - 90 bases behind it, the start of a reverse molecule
        # (adapter + barcode + UMI + polyT: an isolated forward adapter, split 25 bases in front of it): two split positions less than
        # 100 apart, the second is removed (ChimeraFindernew.java:L273-281)
        gap = 10 + 10 * (variant % 6)
        site = AD3 + rng.choice(bcs) + rnd_seq(rng, 12) + "T" * rng.randrange(24, 36)
        mol = molecule(330) + TSO + rnd_seq(rng, gap) + site + rnd_seq(rng, 420)
        if kind.endswith("rev"):
            mol = revcomp_str(mol)
    else:
        # internal polyA + adapter whose final alignment has more errors than maxCompleteSeqNeedlemanMismatches (PolyATadapterInternalSearcherBase.java:L203-204)
        n_err = int(kind[-4])
        ad = list(AD3)
        keep = set(range(0, 4)) | set(range(9, 13)) | set(range(18, 22))      # three 4-mers stay (the k-mer gate wants three)
        pos = [q for q in range(len(ad)) if q not in keep]
        rng.shuffle(pos)
        for q in pos[:n_err]:
            ad[q] = rng.choice([c for c in "ACGT" if c != ad[q]])
        a = rnd_seq(rng, 600)
        site = "A" * (28 + variant % 10) + revcomp_str(rnd_seq(rng, 12)) + revcomp_str(rng.choice(bcs)) + revcomp_str("".join(ad))
        if variant % 2:
            site = revcomp_str(site)
        mol = TSO + a[:300] + site + a[300:] + "A" * 30 + revcomp_str(rnd_seq(rng, 12)) + revcomp_str(rng.choice(bcs)) + revcomp_str(AD3)
    if rng.random() < 0.5 and not kind.startswith("close_splits"):
        mol = revcomp_str(mol)
    qual = "".join(chr(33 + rng.randrange(3, 35)) for _ in mol)
    return f"x{idx:04d} runid=r4 ch={idx % 512}", mol, qual, kind


def gen_pass2x_3p(g):
    """supplement to pass2w_3p: (0) reads aimed at branches of the splitter the wide set misses, (1) the wide reads of chunks 0 .. 23 with --trimfastq"""
    a = gen_pass2w(g, False, 1, False, 4201, n_chunks=12, per_chunk=5, reader=targeted_read, note=", reads aimed at single branches of ChimeraFindernew / PolyATadapterInternalSearcherBase")
    b = gen_pass2w(g, False, 1, False, 4101, n_chunks=24, per_chunk=5, trim=True)
    a["sections"] += b["sections"]
    a["_steps"] += b["_steps"]
    merge_hits(a["_hits"], b["_hits"])
    return a


def gen_pass2x_5p(g):
    """the wide 5' reads of chunks 0 .. 23 with --trimfastq (FastqRecordExt.getRecordForWriting L210-217, L303-304, the 5' arm)"""
    return gen_pass2w(g, True, 1, True, 4103, n_chunks=24, per_chunk=5, trim=True)


def gen_pass2w_3p(g):
    return gen_pass2w(g, False, 1, False, 4101)


def gen_pass2p(g):
    """the wide reads under `scanfastq -p / -f / -w` (NanoporeReadScannerMain.java:L228-234): 3' with -p 12 -f 0.8 -w 120 (40 chunks), 5' with the polyA search on and
    -p 20 -f 0.7 -w 140 (30 chunks) -- the finder's parameters and, through windowSearchForPolyA, how far the splitter keeps from the read ends"""
    a = gen_pass2w(g, False, 1, False, 4101, n_chunks=40, per_chunk=5, polya=(12, 0.8, 120), note=", -p 12 -f 0.8 -w 120")
    b = gen_pass2w(g, True, 1, False, 4104, n_chunks=30, per_chunk=5, polya=(20, 0.7, 140), note=", -p 20 -f 0.7 -w 140")
    a["sections"] += b["sections"]
    a["_steps"] += b["_steps"]
    merge_hits(a["_hits"], b["_hits"])
    return a


def gen_pass2k(g):
    """round 6: the wide reads through Parser.call under OTHER VALUES OF config.xml's knobs (the file the reference is started with, its elements
    replaced; everything else as shipped) -- the knobs the product takes at run time (smi_run_knobs):
    (0) 3': fewer allowed adapter mismatches, a longer minimal read, a stricter complete TSO / complete adapter in the splitter, another internal
        polyA window, umi_length 10 (the barcode + UMI stretch between an internal polyA and its adapter);
    (1) 5' with the polyA search on: another AdapterSearchWindow, other mismatch limits of both adapters;
    (2) 3': another adapter sequence (two bases of the shipped one changed, in `sequence` and `sequence_complete`) with one more mismatch allowed"""
    ka = {"adapter_for3pBarcoding/maxNeedlemanMismatches": 2, "adapter_for3pBarcoding/maxCompleteSeqNeedlemanMismatches": 3, "readscanner/minReadLength": 300,
          "tso_for3pBarcoding/maxCompleteSeqNeedlemanMismatches": 4, "polyAT/internalpATlength": 12, "polyAT/internalFractionATInPolyAT": 0.8, "umis/umi_length": 10}
    kb = {"fiveprimeadapter_for5pBarcoding/AdapterSearchWindow": 40, "fiveprimeadapter_for5pBarcoding/maxNeedlemanMismatches": 1,
          "fiveprimeadapter_for5pBarcoding/maxCompleteSeqNeedlemanMismatches": 2, "threeprimeadapter_for5pBarcoding/maxCompleteSeqNeedlemanMismatches": 2,
          "readscanner/minReadLength": 260}
    kc = {"adapter_for3pBarcoding/sequence": "CTTCCGTTCA", "adapter_for3pBarcoding/sequence_complete": "CTACACGACGCTCTTCCGTTCA", "adapter_for3pBarcoding/maxNeedlemanMismatches": 4,
          "tso_for3pBarcoding/sequence_complete": "AAGCAGTGGTATCAACGCAGAGTGAAT", "tso_for3pBarcoding/maxCompleteSeqNeedlemanMismatches": 7}
    a = gen_pass2w(g, False, 1, False, 4101, n_chunks=40, per_chunk=5, knobs=ka, note=", config.xml with other knob values (see `knobs`)")
    b = gen_pass2w(g, True, 1, False, 4104, n_chunks=30, per_chunk=5, knobs=kb, note=", config.xml with other knob values (see `knobs`)")
    c = gen_pass2w(g, False, 1, False, 4101, n_chunks=30, per_chunk=5, knobs=kc, note=", config.xml with another adapter / complete TSO sequence (see `knobs`)")
    for x in (b, c):
        a["sections"] += x["sections"]
        a["_steps"] += x["_steps"]
        merge_hits(a["_hits"], x["_hits"])
    return a


def gen_pass2t(g):
    """round 6: the wide 3' reads through Parser.call with OTHER TSO PARAMETERS OF THE READ SCAN in config.xml (tso_for3pBarcoding: sequence -- two bases of the
    shipped one changed --, maxNeedlemanMismatches, the two rescue rules, windowForTSOsearch): PolyATadapterAnalyzer_3pBCUMI.scanReadForTSOs /
    PolyATadapterAnalyzerBase.scanForTSO with the values the product takes at run time (smi_run_knobs.tso_scan*)"""
    kt = {"tso_for3pBarcoding/sequence": "AACGCAGAGTGAATGG", "tso_for3pBarcoding/maxNeedlemanMismatches": 4, "tso_for3pBarcoding/minTSO_NeedlemanConsecutiveMatches": 7,
          "tso_for3pBarcoding/minTSO_TwoBestConsecutiveMatches": 11, "tso_for3pBarcoding/windowForTSOsearch": 70}
    ku = {"tso_for3pBarcoding/maxNeedlemanMismatches": 7, "tso_for3pBarcoding/minTSO_NeedlemanConsecutiveMatches": 10, "tso_for3pBarcoding/windowForTSOsearch": 110}
    a = gen_pass2w(g, False, 1, False, 4101, n_chunks=30, per_chunk=5, knobs=kt, note=", config.xml with another TSO for the read scan (see `knobs`)")
    b = gen_pass2w(g, False, 1, False, 4101, n_chunks=30, per_chunk=5, knobs=ku, note=", config.xml with other limits and another window of the read scan's TSO (see `knobs`)")
    a["sections"] += b["sections"]
    a["_steps"] += b["_steps"]
    merge_hits(a["_hits"], b["_hits"])
    return a


def gen_pass2w_3p_ed2(g):
    return gen_pass2w(g, False, 2, False, 4102)


def gen_pass2w_5p(g):
    return gen_pass2w(g, True, 1, True, 4103)


def gen_pass2w_5p_polya(g):
    return gen_pass2w(g, True, 1, False, 4104)


# ---------------------------------------------------------------------------------------------------------------------
# scan statistics: ReadFlags.addForCounting per record + the two read-length sums as Parser.call adds them (Parser.java:L115-118), then
# ReadFlags.print(PrintStream) -- the text of the table ReadScanner.html shows, for sets of flag words drawn from the flags the reference
# itself gave the records of ref_exec_pass2_*.json (+ the split / multi-chimeric bits) at random multiplicities
# ---------------------------------------------------------------------------------------------------------------------
def gen_stats_print(g):
    j = g.j
    rng = random.Random(4242)
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    s = g.section("new ReadFlags(); per record addForCounting(flag, readLength) and sumReadLengthPassed / sumReadLengthFailed.addAndGet(readLength) "
                  "by PASSED_TOTAL (Parser.call L115-118); nReadsSplit.addAndGet; print(PrintStream): every string handed to the stream", RFLAGS,
                  "print:(Ljava/io/PrintStream;)V")
    pool = []
    fv = None
    for name in ("pass2_3p", "pass2_5p", "pass2_5p_polya", "pass2_3p_ed2"):
        d = json.load(open(os.path.join(GOLD, f"ref_exec_{name}.json")))
        fv = d["sections"][0]["flag_values"]
        pool += [int(c["result"]["flag"]) for c in d["sections"][0]["cases"] if "flag" in c["result"]]
    s["flag_values"] = fv
    captured = []
    j.natives["java/io/PrintStream.println"] = lambda jj, o, *a: captured.append((a[0] if a else "") + "\n")
    j.natives["java/io/PrintStream.print"] = lambda jj, o, *a: captured.append(a[0] if a else "")
    kept = {}
    for case in range(8):
        rf = j.new(RFLAGS)
        kept[case] = rf
        n = [3, 40, 500, 5000, 1, 77, 1234, 20000][case]
        recs = []
        for _ in range(n):
            f = rng.choice(pool)
            if rng.random() < 0.1:
                f |= fv["READS_AFTER_SPLIT"]
            if rng.random() < 0.01:
                f = fv["MULTI_CHIMERIC_READS_DISCARDED"] | fv["FAILED"]
            ln = rng.randrange(150, 4000)
            recs.append([f, ln])
            j.call_virtual(rf, "addForCounting", "(JI)V", f, ln)
            j.call_virtual(rf.f["sumReadLengthPassed" if f & fv["PASSED_TOTAL"] else "sumReadLengthFailed"], "addAndGet", "(J)J", ln)
        n_split = sum(1 for f, _ in recs if f & fv["READS_AFTER_SPLIT"]) // 2
        j.call_virtual(rf.f["nReadsSplit"], "addAndGet", "(I)I", n_split)
        del captured[:]
        ps = JObject("java/io/PrintStream")
        ps.native = []
        throws = None
        try:
            j.call_virtual(rf, "print", "(Ljava/io/PrintStream;)V", ps)
        except JavaThrow as e:          # e.g. no failed read at all: the mean read length is an integer division by the count
            throws = e.obj.cls if hasattr(e, "obj") else str(e)
        s["cases"].append({"throws": throws, "records": recs if n <= 500 else None, "seed_note": "records drawn with random.Random(4242) in tools/make_ref_exec.py::gen_stats_print",
                           "counts": {nm: sum(1 for f, _ in recs if f & v) for nm, v in fv.items()}, "n_records": n,
                           "sum_len_passed": sum(l for f, l in recs if f & fv["PASSED_TOTAL"]), "sum_len_failed": sum(l for f, l in recs if not f & fv["PASSED_TOTAL"]),
                           "n_reads_split": n_split, "text": "".join(captured)})
    # ReadFlags.mergeStats (the `mergestats` sub-command's sum): an empty ReadFlags that merges the sets of cases 1, 2 and 5, then prints
    merged = j.new(RFLAGS)
    for k in (1, 2, 5):
        j.call_virtual(merged, "mergeStats", "(L" + RFLAGS + ";)V", kept[k])
    del captured[:]
    ps = JObject("java/io/PrintStream")
    ps.native = []
    j.call_virtual(merged, "print", "(Ljava/io/PrintStream;)V", ps)
    s["merged"] = {"cases": [1, 2, 5], "text": "".join(captured)}
    out["sections"].append(g.finish(s))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# assignumis: read name -> scan data (FastqRecordExt.getScanDatFromReadName L395-496) -> UMI pair distance
# (ClusteringEditDistanceBase.calcEditDistances = lambda$static$7 L297-350 + calcBestEditDistance L67-80), 3' and 5' (-p)
# ---------------------------------------------------------------------------------------------------------------------
UPAR = "com/rw/umifinder/parameters/ParametersBarcodeUMiFinderAppParams"
CED = "com/rw/clustering/ClusteringEditDistanceBase"
ONR = "com/rw/umifinder/reads/nanopore/OneNanoporeResult"
NREAD = "com/rw/umifinder/reads/nanopore/NanoporeRead"
RSD = "com/rw/umifinder/reads/nanopore/NanoporeRead$ReadScanData"


class UmiSide:
    def __init__(self, g, five_prime):
        import ref_params

        self.j = j = g.j
        self.par, self.report = ref_params.load_config(j, UPAR)
        self.par.f["scantype"] = j.get_static(SCANTYPE, "FIVEP_BARCODE" if five_prime else "THREEP_BARCODE")   # -p (UmiFinderMain.java:L249)
        self.five = five_prime

    def scan_data(self, name):
        """NanoporeRead$ReadScanData.generateReadScanData L86 without the SAM record: the name is all it parses"""
        j = self.j
        sup = j.natives["java/util/function/Function.identity"](j)      # placeholder object; replaced below by a real supplier
        sup.native = lambda *a: j.new(RSD)                                   # Supplier.get -> new ReadScanData()  (L86: ReadScanData::new)
        opt = j.call_static(FQX, "getScanDatFromReadName", f"(Ljava/lang/String;L{UPAR};Ljava/util/function/Supplier;)L{GOPT};", name, self.par, sup)
        return j.call_virtual(opt, "orNull", "()Ljava/lang/Object;")

    def result_for(self, sd):
        j = self.j
        nr = j.new_object(NREAD)
        nr.f["readScanData"] = j.call_static(GOPT, "of", f"(Ljava/lang/Object;)L{GOPT};", sd)
        o = j.new_object(ONR)
        o.f["nanoporeRead"] = nr
        o.f["userObject"] = j.natives["java/util/Optional.empty"](j)   # the field's initialiser (the constructor is not run)
        return o

    def distance(self, r1, r2):
        j = self.j
        f = j.get_static(CED, "calcEditDistances")
        ced = j.call_lambda(f, [r1, r2, self.par])
        b = ced.f["bestEditDistance"]
        return {"ed": j.call_virtual(b, "getED", "()B"),
                "pos1": j.call_virtual(b, "getPos1", "()Lcom/rw/clustering/PlusMinusOnePosData$PlusMinusOneEnum;").f["$name"],
                "pos2": j.call_virtual(b, "getPos2", "()Lcom/rw/clustering/PlusMinusOnePosData$PlusMinusOneEnum;").f["$name"]}


def fake_name(rng, k, five_prime, bc, umi, rev, ae, shift, ed=0):
    """a read name in scanfastq's format around a given UMI (the barcode / adapter neighbourhood is synthetic)"""
    ad3 = "AGA"                      # the three adapter bases the name carries (nbasesOfAdapterSeqInReadname)
    if five_prime:
        bc_start, bc_end = ae + 1 + shift, ae + 16 + shift
        x = ("TCT" + rnd_seq(rng, shift) + bc + umi + rnd_seq(rng, 40))[:42] if shift >= 0 else ("TCT" + bc[-shift:] + umi + rnd_seq(rng, 40))[:42]
        core = f"AE={ae}_bc={bc}_ed={ed}_bcStart={bc_start}_bcEnd={bc_end}"
    else:
        bc_start, bc_end = ae - 1 + shift, ae - 16 + shift
        stranded = rnd_seq(rng, 40) + revcomp_str(umi) + revcomp_str(bc) + (rnd_seq(rng, -shift) if shift < 0 else "")
        if shift > 0:
            stranded = stranded[:-shift]
        x = (stranded + ad3)[-43:]
        core = f"PS={ae - 60}_PE={ae - 30}_AE={ae}_bc={bc}_ed={ed}_bcStart={bc_start}_bcEnd={bc_end}"
    q = f"{10 + (k * 7) % 23}.{k % 10}" if k % 4 else f"{12 + k % 9}"
    return f"read{k}_{'REV' if rev else 'FWD'}_{core}_rk={1 + k % 40}_X={x}_Q={q}_{k + 1:x}"


def gen_umi(g, five_prime, seed, n_mol=14, umi_len=12):
    j = g.j
    rng = random.Random(seed)
    side = UmiSide(g, five_prime)
    if umi_len != 12:                  # <umi_length> of config.xml's <umis> (UMIparameters.umi_length: a primitive int, set by JAXB from the element)
        side.par.f["umis"].f["umi_length"] = umi_len
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    s = g.section(("5-prime (-p)" if five_prime else "3-prime") + " assignumis: FastqRecordExt.getScanDatFromReadName(name) (L395-496), then "
                  "ClusteringEditDistanceBase.calcEditDistances (lambda$static$7, L297-350) for read pairs: 9 limited Levenshtein "
                  "distances between the 12-mers at offsets -1 / 0 / +1 behind the barcode, best = first strict minimum in EnumSet order "
                  "(calcBestEditDistance L67-80)", CED, "calcEditDistances / calcBestEditDistance")
    s["five_prime"] = five_prime
    s["umi_length"] = umi_len
    if umi_len != 12:
        s["title"] += f".  config.xml with <umi_length>{umi_len}</umi_length>: the {umi_len}-mers; `post_bc_umi` = OneNanoporeResult.getPostBCUMIseqOffset(read, params, offset) " \
                      "for offset -1 / 0 / +1 (L107-113: what U7 -- offset 0 -- and a centre's U8 are cut from)"
    bc = rnd_seq(rng, 16)
    names = []
    for m in range(n_mol):
        umi = rnd_seq(rng, umi_len)
        for r in range(rng.choice([1, 2, 3, 3, 4])):
            u = umi if r == 0 else mutate(rng, umi, rng.choice([0, 1, 1, 2, 3]))[:umi_len].ljust(umi_len, "A")
            names.append(fake_name(rng, len(names), five_prime, bc, u, rng.random() < 0.5, rng.randrange(300, 900), rng.choice([0, 0, 0, -1, 1])))
    parsed, results = [], []
    for nm in names:
        sd = side.scan_data(nm)
        ad, bcr = sd.f["adapter_result"], sd.f["barcode_Result"]

        def iv(o, k):
            v = None if o is None else o.f.get(k)
            if isinstance(v, JObject) and isinstance(v.native, tuple):   # java.util.Optional
                v = v.native[0]
            if isinstance(v, JObject) and "Optional" in v.cls:            # com.google.common.base.Optional
                v = j.call_virtual(v, "orNull", "()Ljava/lang/Object;")
            if isinstance(v, JObject):
                raise TypeError(f"{k}: {v.cls}")
            return None if v is None else (v.v if isinstance(v, JBox) else v)

        seq = sd.f["seq"]
        parsed.append({"name": nm, "forward": sd.f["forward"].f["$name"], "adapter_end": iv(ad, "end"),
                       "barcode": None if bcr is None else {"seq": j.call_virtual(bcr.f["barcodeseq"], "toString", "()Ljava/lang/String;"),
                                                            "ed": iv(bcr, "editDistance"), "start": iv(bcr, "start"), "end": iv(bcr, "end"),
                                                            "rank": iv(bcr, "rank")},
                       "x_codes": None if seq is None else list(seq.f["naData"].a), "mean_qv": iv(sd, "mean_qv"), "read_id": sd.f["read_id"]})
        results.append(side.result_for(sd))
        if umi_len != 12:
            post = []
            for off in (-1, 0, 1):
                o = j.call_static(ONR, "getPostBCUMIseqOffset", f"(L{ONR};L{UPAR};I)Ljava/util/Optional;", results[-1], side.par, off)
                v = o.native[0] if isinstance(o.native, tuple) else None
                post.append(None if v is None else j.call_virtual(v, "toString", "()Ljava/lang/String;"))
            parsed[-1]["post_bc_umi"] = post
    s["names"] = parsed
    for a in range(len(names)):
        for b in range(a + 1, len(names)):
            if (a * 31 + b) % 3 == 0 or abs(a - b) < 4:
                s["cases"].append({"i": a, "j": b, "distance": side.distance(results[a], results[b]), "reverse": side.distance(results[b], results[a])})
    out["sections"].append(g.finish(s))
    return out


def gen_umi_3p(g):
    return gen_umi(g, False, 808)


def gen_umi_5p(g):
    return gen_umi(g, True, 818)


def gen_umi_3p_len10(g):
    """round 6: umis/umi_length = 10 (10x 3' v2 / 5' v1-v2 chemistry), a value of config.xml the product takes at run time"""
    return gen_umi(g, False, 828, umi_len=10)


def gen_umi_5p_len10(g):
    return gen_umi(g, True, 838, umi_len=10)

# ---------------------------------------------------------------------------------------------------------------------
# a-14: ChimeraFindernew.findSplitPositions on whole records (pass 2, before the scan)
# ---------------------------------------------------------------------------------------------------------------------
CHIM = "com/rw/nanoporereadscanner/analyzers/ChimeraFindernew"
NPSET = "com/rw/nanopore/analyzers/parameters/NeedlemanParameters$OneSet"
RFLAGS = "com/rw/nanoporereadscanner/stats/ReadFlags"


def gen_chimera(g, five_prime=False, seed=909, n_reads=26):
    j = g.j
    rng = random.Random(seed)
    p2 = Pass2(g, five_prime, 1, dont_search_polya=False)
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    s = g.section(("5-prime" if five_prime else "3-prime") + " barcoding: new ChimeraFindernew(params, new NeedlemanParameters$OneSet())"
                  ".findSplitPositions(record, readFlags, skip = false) (L107-332): the records it returns (names with the _<reason>sp<k> tag, bases) "
                  "and the flag it leaves on each; each case under three iteration orders of the JDK hash containers", CHIM,
                  "findSplitPositions:(L...FastqRecordExt;L...ReadFlags;Z)Ljava/util/Collection;")
    s["five_prime"] = five_prime
    bcs = [rnd_seq(rng, 16) for _ in range(12)]

    def molecule(kind):
        seq, _q, _b = synth_read(rng, bcs, five_prime, kind)
        return seq

    for idx in range(n_reads):
        k = idx % 9
        if k in (0, 1):
            seq = molecule(idx)                                   # a single molecule
        elif k in (2, 3, 4):
            seq = molecule(idx) + molecule(idx + 1)               # ligation chimera of two
        elif k == 5:
            seq = molecule(idx) + molecule(idx + 1) + molecule(idx + 2)
        elif k == 6:
            seq = molecule(1) + molecule(2) + molecule(3) + molecule(4)   # too many split points -> MULTI, read kept whole
        elif k == 7:
            a = molecule(idx)
            seq = a[:len(a) // 2] + "A" * 35 + revcomp_str(AD3) + rnd_seq(rng, 40) + TSO + a[len(a) // 2:]  # internal polyA + adapter + TSO
        else:
            seq = rnd_seq(rng, rng.randrange(150, 260))           # below 2 * 70 + 100
        qual = "".join(chr(33 + rng.randrange(3, 35)) for _ in seq)
        name = f"read{idx:04d} runid=abc ch={idx}"
        results = []
        for order in ("insertion", "reverse", ("shuffle", idx + 1)):
            j.hash_order = order
            fq = p2.record(name, seq, qual)
            cf = j.new(CHIM, f"(L{PAR};L{NPSET};)V", p2.par, j.new(NPSET))
            rf = j.new_object(RFLAGS)   # statistics object: its counters are not outputs of this path; atomics created, the rest left null
            jc = j.load(RFLAGS)
            for fname, fdesc in jc.instance_fields:
                if fdesc in ("Ljava/util/concurrent/atomic/AtomicLong;", "Ljava/util/concurrent/atomic/AtomicInteger;"):
                    rf.f[fname] = j.natives[fdesc[1:-1] + ".<new>"](j)
            try:
                coll = j.call_virtual(cf, "findSplitPositions", f"(L{FQX};L{RFLAGS};Z)Ljava/util/Collection;", fq, rf, 0)  # skip = !doSplitChimericReads (Parser.java:L180)
                recs = [{"name": r.f["readName"], "length": len(r.f["readString"]), "bases_head": r.f["readString"][:24],
                         "flag": u64(r.f["scanResult"].f["flag"])} for r in coll.native]
                results.append(recs)
            except JavaThrow as e:
                results.append({"throws": e.obj.cls, "message": e.obj.f.get("message"), "in": e.trace[:6]})
        j.hash_order = None
        s["cases"].append({"name": name, "seq": seq, "hash_orders_agree": all(r == results[0] for r in results[1:]), "records": results[0]})
        if idx % 5 == 0:
            print(f"  chimera {idx + 1}/{n_reads}  {time.time() - g.t0:.0f}s", flush=True)
    vals = j.call_static(FLAGS, "values", f"()[L{FLAGS};")
    s["flag_values"] = {v.f["$name"]: u64(j.call_virtual(v, "getValue", "()J")) for v in vals.a}
    out["sections"].append(g.finish(s))
    return out


def gen_chimera_3p(g):
    return gen_chimera(g, False, 909)


# ---------------------------------------------------------------------------------------------------------------------
GT = "com/rw/umifinder/bamreaders/GennameTagger"
RFR = "picard/annotation/RefFlatReader"
TTP = "picard/util/TabbedTextFileWithHeaderParser"
SAMREC = "htsjdk/samtools/SAMRecord"
REFFLAT_COLUMNS = ["GENE_NAME", "TRANSCRIPT_NAME", "CHROMOSOME", "STRAND", "TX_START", "TX_END", "CDS_START", "CDS_END", "EXON_COUNT",
                   "EXON_STARTS", "EXON_ENDS"]


def install_gene_io(j, lines, ref_names):
    """file / header stand-ins, all of them INPUT plumbing: the tab-separated rows of the annotation file reach RefFlatReader.load() as real
    TabbedTextFileWithHeaderParser$Row objects (the class-file's own getField / getIntegerField run), the sequence dictionary answers
    getSequence(name) != null for the given names, the logger is inert, and a SAMRecord is a bag of the five values GennameTagger asks for
    (alignment blocks and end computed by htsjdk's own Cigar / SAMUtils bytecode from the CIGAR string)."""
    def parser_init(jj, o, _file, labels):
        m = jj.natives["java/util/HashMap.<new>"](jj)
        for i, lab in enumerate(labels.a):
            m.native.put(lab, JBox("java/lang/Integer", i))
        o.f["columnLabelIndices"] = m
        o.native = {"line": 0}

    def parser_iter(jj, o):
        rows = []
        for ln in lines:
            if not ln or ln.startswith("#"):
                continue  # BasicInputParser skips blank lines and comments
            parts = ln.split("\t")
            arr = JArray("Ljava/lang/String;", parts)
            rows.append(jj.new(TTP + "$Row", f"(L{TTP};[Ljava/lang/String;Ljava/lang/String;)V", o, arr, ln))
        lst = JObject("java/util/ArrayList")
        lst.native = rows
        return jj.natives["java/util/ArrayList.iterator"](jj, lst)

    def guava_stream(jj, it):
        """com.google.common.collect.Streams.stream(Iterable) = a sequential stream over iterable.iterator() (the Iterable's own bytecode)"""
        itr = jj.call_virtual(it, "iterator", "()Ljava/util/Iterator;")
        lst = JObject("java/util/ArrayList")
        lst.native = []
        while jj.call_virtual(itr, "hasNext", "()Z"):
            lst.native.append(jj.call_virtual(itr, "next", "()Ljava/lang/Object;"))
        return jj.natives["java/util/ArrayList.stream"](jj, lst)

    H = j.hooks  # these classes ARE in the jars: their file / header plumbing is replaced, nothing of the path under test
    H["com/google/common/collect/Streams.stream:(Ljava/lang/Iterable;)Ljava/util/stream/Stream;"] = guava_stream
    H["com/google/common/collect/Streams.<clinit>:()V"] = None
    # GennameTagger.<clinit> builds LOCUS_FUNCTION_SCORES with io.vavr.collection.Stream.of(entries).collect(toMap(.., TreeMap::new)): a
    # sequential ordered stream over the four entries (vavr is not among the jars executed)
    j.natives["io/vavr/collection/Stream.of:([Ljava/lang/Object;)Lio/vavr/collection/Stream;"] = \
        j.natives["java/util/stream/Stream.of:([Ljava/lang/Object;)Ljava/util/stream/Stream;"]
    j.natives["io/vavr/collection/Stream.collect"] = j.natives["java/util/stream/Stream.collect"]
    H[TTP + ".<init>:(Ljava/io/File;[Ljava/lang/String;)V"] = parser_init
    H[TTP + ".iterator:()Lhtsjdk/samtools/util/CloseableIterator;"] = parser_iter
    H[TTP + ".getCurrentLineNumber:()I"] = lambda jj, o: 0
    H[TTP + ".close:()V"] = lambda jj, o: None
    H["htsjdk/samtools/util/Log.<clinit>:()V"] = None
    H["htsjdk/samtools/util/Log.*"] = lambda jj, *a: None
    H["htsjdk/samtools/util/Log.getInstance:(Ljava/lang/Class;)Lhtsjdk/samtools/util/Log;"] = lambda jj, *a: JObject("htsjdk/samtools/util/Log")
    names = set(ref_names)
    H["htsjdk/samtools/SAMSequenceDictionary.getSequence:(Ljava/lang/String;)Lhtsjdk/samtools/SAMSequenceRecord;"] = (
        lambda jj, o, nm: JObject("htsjdk/samtools/SAMSequenceRecord") if nm in names else None)
    H[SAMREC + ".<clinit>:()V"] = None
    H[SAMREC + ".getReferenceName:()Ljava/lang/String;"] = lambda jj, o: o.native["ref"]
    H[SAMREC + ".getAlignmentStart:()I"] = lambda jj, o: o.native["start"]
    H[SAMREC + ".getAlignmentEnd:()I"] = lambda jj, o: o.native["end"]
    H[SAMREC + ".getAlignmentBlocks:()Ljava/util/List;"] = lambda jj, o: o.native["blocks"]
    H[SAMREC + ".getReadNegativeStrandFlag:()Z"] = lambda jj, o: 1 if o.native["flag"] & 16 else 0
    H[SAMREC + ".setAttribute:(Ljava/lang/String;Ljava/lang/Object;)V"] = lambda jj, o, t, v: o.native["calls"].append([t, v])


def sam_record(j, ref, flag, pos0, cigar):
    """unmapped: reference "*", start 0, no cigar (SAMRecord.NO_ALIGNMENT_*)"""
    o = JObject(SAMREC)
    if flag & 4 or ref is None:
        lst = JObject("java/util/ArrayList")
        lst.native = []
        o.native = {"ref": "*", "start": 0, "end": 0, "blocks": lst, "flag": flag, "calls": []}  # getAlignmentEnd of an unmapped read: NO_ALIGNMENT_START
        return o
    text = "".join(f"{ln}{op}" for op, ln in cigar)
    cg = j.call_static("htsjdk/samtools/TextCigarCodec", "decode", "(Ljava/lang/String;)Lhtsjdk/samtools/Cigar;", text)
    blocks = j.call_static("htsjdk/samtools/SAMUtils", "getAlignmentBlocks", "(Lhtsjdk/samtools/Cigar;ILjava/lang/String;)Ljava/util/List;",
                           cg, pos0 + 1, "read cigar")
    ref_len = j.call_virtual(cg, "getReferenceLength", "()I")
    o.native = {"ref": ref, "start": pos0 + 1, "end": pos0 + 1 + ref_len - 1, "blocks": blocks, "flag": flag, "calls": []}
    return o


def gene_reads(rng, rows, n):
    """reads along transcripts (N between exons), around genes, across genes, unmapped, on an unknown contig"""
    out = []
    tx = [r for r in rows if r[2] == "chr12"]
    for k in range(n):
        r = tx[rng.randrange(len(tx))]
        es = [int(x) + 1 for x in r[9].split(",") if x]
        ee = [int(x) for x in r[10].split(",") if x]
        flag = 16 if rng.random() < 0.5 else 0
        kind = k % 7
        if kind <= 2:
            i = rng.randrange(len(es))
            jx = min(len(es), i + rng.randrange(1, 5))
            start = rng.randrange(es[i], ee[i] + 1)
            cigar, p = ([("S", rng.randrange(1, 30))] if kind == 1 else []), start
            for q in range(i, jx):
                stop = ee[q] if q < jx - 1 else rng.randrange(max(p, es[q]), ee[q] + 1)
                if q > i:
                    cigar.append(("N", es[q] - p))
                    p = es[q]
                if stop - p + 1 > 0:
                    cigar.append(("M", stop - p + 1))
                    p = stop + 1
            if not any(op == "M" for op, _ in cigar):
                cigar.append(("M", 1))
            out.append(("chr12", flag, start - 1, cigar))
        elif kind == 3:
            start = rng.randrange(max(1, es[0] - 3000), ee[-1] + 3000)
            out.append(("chr12", flag, start - 1, [("M", rng.randrange(20, 900)), ("I", 3), ("D", rng.randrange(1, 40)), ("=", rng.randrange(1, 400)), ("X", 2)]))
        elif kind == 4:
            start = rng.randrange(max(1, es[0] - 20000), ee[-1])
            out.append(("chr12", flag, start - 1, [("M", rng.randrange(500, 4000)), ("N", rng.randrange(100, 60000)), ("M", rng.randrange(100, 3000))]))
        elif kind == 5:
            out.append((None, flag | 4, -1, []) if rng.random() < 0.5 else ("chrUn", flag, rng.randrange(10 ** 6), [("M", 500)]))
        else:
            start = rng.randrange(es[0], ee[-1])
            out.append(("chr12", flag, start - 1, [("S", 5), ("D", rng.randrange(1, 60))]))   # no aligned block at all
    return out


def gen_gene(g, n_rows=1500, n_reads=420, seed=1212):
    import gzip

    j = g.j
    rng = random.Random(seed)
    text = gzip.open(os.path.join(GOLD, "chr12_head1500.refFlat.gz"), "rt").read()
    lines = text.split("\n")[:n_rows]
    extra = ["GOOD\tt1\tc1\t+\t100\t1000\t200\t900\t2\t100,600,\t300,1000,", "GOOD\tt2\tc1\t+\t150\t1200\t200\t900\t1\t150,\t1200,",
             "TWOSTRANDS\tt3\tc1\t+\t5000\t6000\t5000\t6000\t1\t5000,\t6000,", "TWOSTRANDS\tt4\tc1\t-\t5000\t6000\t5000\t6000\t1\t5000,\t6000,",
             "TWICE\tt5\tc1\t-\t7000\t8000\t7000\t8000\t1\t7000,\t8000,", "TWICE\tt5\tc1\t-\t7000\t8000\t7000\t8000\t1\t7000,\t8000,",
             "COUNT\tt6\tc1\t+\t9000\t9500\t9000\t9500\t3\t9000,9200,\t9100,9500,", "OVERLAP\tt7\tc1\t+\t10000\t10500\t10000\t10500\t2\t10000,10100,\t10100,10500,",
             "OVERLAP2\tt8\tc1\t+\t11000\t11500\t11000\t11500\t2\t11000,11099,\t11100,11500,", "EMPTY\tt9\tc1\t+\t12000\t12500\t12000\t12500\t1\t12100,\t12100,",
             "ELSEWHERE\tt10\tc9\t+\t100\t1000\t100\t1000\t1\t100,\t1000,", "SAMEPLACE_A\tt11\tc1\t-\t20000\t21000\t20000\t21000\t1\t20000,\t21000,",
             "SAMEPLACE_B\tt12\tc1\t-\t20000\t21000\t20500\t21000\t1\t20000,\t21000,", "ANTISENSE\tt13\tc1\t+\t20000\t21000\t21000\t21000\t1\t20000,\t21000,"]
    lines = [ln for ln in lines if ln] + extra
    refs = ["chr1", "chr12", "chrUn", "c1"]
    install_gene_io(j, lines, refs)
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar + picard-2.23.9.jar + htsjdk-4.1.3.jar", "sections": []}
    s = g.section("RefFlatReader.load() over the rows of tests/golden/chr12_head1500.refFlat.gz plus the hand-made rows in `extra_rows`, then "
                  "GennameTagger.annotateGene(record) (GennameTagger.java:L73-121, L382) with ALLOW_MULTI_GENE_READS = true as its constructor sets "
                  "it (L65): the setAttribute calls it makes (value null = tag removed) or the exception it throws.  HashMap / HashSet iteration "
                  "in java.util.HashMap table order from the keys' own hashCode() (tier D)", GT, "annotateGene:(Lhtsjdk/samtools/SAMRecord;)Lhtsjdk/samtools/SAMRecord;")
    s["extra_rows"] = extra
    s["ref_names"] = refs
    s["n_rows_of_sample"] = len(lines) - len(extra)
    j.hash_order = "jdk"
    rdr = j.new(RFR, "(Ljava/io/File;Lhtsjdk/samtools/SAMSequenceDictionary;)V", None, JObject("htsjdk/samtools/SAMSequenceDictionary"))
    det = j.call_virtual(rdr, "load", "()Lhtsjdk/samtools/util/OverlapDetector;")
    print(f"  refFlat loaded by the reference's RefFlatReader.load(): {time.time() - g.t0:.0f}s", flush=True)
    allg = j.call_virtual(det, "getAll", "()Ljava/util/Set;")
    s["genes_loaded"] = sorted(o.f["name"] for o, _ in allg.native.items_in_insertion_order())
    tagger = j.new_object(GT)
    tagger.f.update({"TAG": "GE", "STRANDTAG": "GS", "FUNCTIONTAG": "XF", "ALLOW_MULTI_GENE_READS": 1, "geneOverlapDetector": det})
    tagger.f["metrics"] = j.new(GT + "$ReadTaggingMetric", f"(L{GT};)V", tagger)
    rows = [ln.split("\t") for ln in lines]
    reads = gene_reads(rng, rows, n_reads)
    reads += [("c1", f, p, cg) for f, p, cg in [(0, 40, [("M", 50)]), (0, 100, [("M", 50)]), (0, 250, [("M", 20)]), (0, 350, [("M", 20)]),
              (16, 250, [("M", 20)]), (0, 1100, [("M", 50)]), (0, 1199, [("M", 50)]), (0, 1200, [("M", 50)]), (16, 20100, [("M", 50)]),
              (0, 20100, [("M", 50)]), (0, 5100, [("M", 50)]), (0, 10090, [("M", 5)]), (0, 250, [("S", 10), ("D", 30)]),
              (0, 11050, [("M", 100)]), (0, 250, [("M", 20), ("N", 19800), ("M", 100)]), (16, 250, [("M", 20), ("N", 19800), ("M", 100)])]]
    for k, (ref, flag, pos0, cigar) in enumerate(reads):
        rec = sam_record(j, ref, flag, pos0, cigar)
        case = {"ref": ref, "flag": flag, "pos0": pos0, "cigar": [[op, ln] for op, ln in cigar]}
        try:
            j.call_virtual(tagger, "annotateGene", f"(L{SAMREC};)L{SAMREC};", rec)
            case["set_attribute"] = rec.native["calls"]
        except JavaThrow as e:
            case["throws"] = e.obj.cls
            case["set_attribute_before_throw"] = rec.native["calls"]
        s["cases"].append(case)
        if k % 60 == 0:
            print(f"  gene {k + 1}/{len(reads)}  {time.time() - g.t0:.0f}s", flush=True)
    j.hash_order = None
    out["sections"].append(g.finish(s))
    return out


GTFR = "org/broadinstitute/dropseqrna/annotation/GTFReader"


def gtf_lines_from_refflat(rows, rng):
    """GENCODE-shaped GTF lines (gene, transcript, exon, CDS) for refFlat rows; the gene record spans its transcripts"""
    by_gene = {}
    for r in rows:
        by_gene.setdefault(r[0], []).append(r)
    out = []
    for k, (name, rs) in enumerate(by_gene.items()):
        if len({(r[2], r[3]) for r in rs}) != 1:
            continue                       # (a name on two chromosomes / strands: the hand-made lines below cover those rules)
        chrom, strand = rs[0][2], rs[0][3]
        gid = f"ENSG{k:08d}.{1 + k % 4}"
        gs, ge = min(int(r[4]) + 1 for r in rs), max(int(r[5]) for r in rs)
        ga = f'gene_id "{gid}"; gene_type "protein_coding"; gene_name "{name}"; level 2;'
        out.append("\t".join([chrom, "HAVANA", "gene", str(gs), str(ge), ".", strand, ".", ga]))
        for q, r in enumerate(rs):
            tid, tname = f"ENST{k:06d}{q:02d}.1", r[1]
            ta = f'gene_id "{gid}"; transcript_id "{tid}"; gene_type "protein_coding"; gene_name "{name}"; transcript_name "{tname}"; tag "basic";'
            out.append("\t".join([chrom, "HAVANA", "transcript", str(int(r[4]) + 1), r[5], ".", strand, ".", ta]))
            es = [int(x) + 1 for x in r[9].split(",") if x]
            ee = [int(x) for x in r[10].split(",") if x]
            order = list(range(len(es)))
            if strand == "-":
                order.reverse()            # GENCODE lists the exons of a minus-strand transcript from its 5' end: the builder sorts them
            cs, ce = int(r[6]) + 1, int(r[7])
            for i in order:
                out.append("\t".join([chrom, "HAVANA", "exon", str(es[i]), str(ee[i]), ".", strand, ".", ta + f' exon_number {i + 1};']))
                lo, hi = max(es[i], cs), min(ee[i], ce)
                if cs <= ce and lo <= hi:
                    out.append("\t".join([chrom, "HAVANA", "CDS", str(lo), str(hi), ".", strand, "0", ta]))
            if rng.random() < 0.2:
                out.append("\t".join([chrom, "HAVANA", "UTR", str(es[0]), str(min(ee[0], es[0] + 10)), ".", strand, ".", ta]))
    return out


def gen_gene_gtf(g, n_rows=420, n_reads=300, seed=1222):
    """--annotationFile <x.gtf>: GTFReader.load (DropseqLib: GTFParser, GTFRecord.validate, GeneFromGTFBuilder, GeneFromGTF) and GennameTagger over its genes"""
    import gzip

    j = g.j
    rng = random.Random(seed)
    text = gzip.open(os.path.join(GOLD, "chr12_head1500.refFlat.gz"), "rt").read()
    rows = [ln.split("\t") for ln in text.split("\n")[:n_rows] if ln]
    lines = ["##description: made from tests/golden/chr12_head1500.refFlat.gz", "##provider: tools/make_ref_exec.py", ""] + gtf_lines_from_refflat(rows, rng)

    def feat(chrom, kind, a, b, strand, attrs):
        return "\t".join([chrom, "hand", kind, str(a), str(b), ".", strand, ".", attrs])

    def at(gid, name, tid=None, tname=None, more=""):
        s_ = f'gene_id "{gid}"; gene_name "{name}";'
        if tid is not None:
            s_ += f' transcript_id "{tid}";'
        if tname is not None:
            s_ += f' transcript_name "{tname}";'
        return s_ + more

    extra = [
        # GOOD: two transcripts, the gene record wider than both (extent = all records); exons given out of order; a CDS inside t1 only
        feat("c1", "gene", 90, 1300, "+", at("g1", "GOOD")), feat("c1", "transcript", 101, 1000, "+", at("g1", "GOOD", "i1", "t1")),
        feat("c1", "exon", 601, 1000, "+", at("g1", "GOOD", "i1", "t1")), feat("c1", "exon", 101, 300, "+", at("g1", "GOOD", "i1", "t1")),
        feat("c1", "CDS", 201, 300, "+", at("g1", "GOOD", "i1", "t1")), feat("c1", "CDS", 601, 900, "+", at("g1", "GOOD", "i1", "t1")),
        feat("c1", "exon", 151, 1200, "+", at("g1", "GOOD", "i2", "t2")),
        # a gene without a gene record: extent from its features; the stop codon widens it beyond the exons
        feat("c1", "exon", 3001, 3500, "-", at("g2", "NOGENEREC", "i3", "t3")), feat("c1", "stop_codon", 2990, 2992, "-", at("g2", "NOGENEREC", "i3", "t3")),
        # rules that make the LENIENT reader skip a gene
        feat("c1", "exon", 5001, 6000, "+", at("g3", "TWOSTRANDS", "i4", "t4")), feat("c1", "exon", 5001, 6000, "-", at("g3", "TWOSTRANDS", "i5", "t5")),
        feat("c1", "exon", 7001, 8000, "-", at("g4", "TWOCHROMS", "i6", "t6")), feat("chr1", "exon", 7001, 8000, "-", at("g4", "TWOCHROMS", "i7", "t7")),
        feat("c1", "gene", 9001, 9400, "+", at("g5", "GENERECSHORT")), feat("c1", "exon", 9001, 9500, "+", at("g5", "GENERECSHORT", "i8", "t8")),
        feat("c1", "exon", 10001, 10100, "+", at("g6", "TWOIDS", "i9", "t9")), feat("c1", "exon", 10201, 10300, "+", at("g6b", "TWOIDS", "i9", "t9")),
        feat("c1", "transcript", 11001, 11500, "+", at("g7", "NOEXONS", "i10", "t10")), feat("c1", "CDS", 11001, 11100, "+", at("g7", "NOEXONS", "i10", "t10")),
        feat("c1", "exon", 12001, 12100, "+", at("g8", "TXNAMETWICE", "i11", "t11")), feat("c1", "exon", 12201, 12300, "+", at("g8", "TXNAMETWICE", "i12", "t11")),
        feat("c1", "exon", 13001, 13100, "+", at("g9", "OVERLAP", "i13", "t13")), feat("c1", "exon", 13100, 13200, "+", at("g9", "OVERLAP", "i13", "t13")),
        feat("c1", "exon", 14100, 14001, "+", at("g10", "NEGEXTENT", "i14", "t14")),
        feat("c1", "gene", 15001, 15500, "+", at("g11", "ONLYGENEREC")),
        feat("c9", "exon", 101, 1000, "+", at("g12", "ELSEWHERE", "i15", "t15")),
        # two genes of one interval and strand both stay (GeneFromGTF.equals has the name); a third on the other strand
        feat("c1", "exon", 20001, 21000, "-", at("g13", "SAMEPLACE_A", "i16", "t16")), feat("c1", "exon", 20001, 21000, "-", at("g14", "SAMEPLACE_B", "i17", "t17")),
        feat("c1", "CDS", 20501, 21000, "-", at("g14", "SAMEPLACE_B", "i17", "t17")), feat("c1", "exon", 20001, 21000, "+", at("g15", "ANTISENSE", "i18", "t18")),
        # gene versions: only the records of the highest version count (Ensembl's gene_version attribute)
        feat("c1", "exon", 30001, 30500, "+", at("g16", "VERSIONED", "i19", "t19", ' gene_version "3";')), feat("c1", "exon", 31001, 31500, "+", at("g16", "VERSIONED", "i20", "t20", ' gene_version "12";')),
        feat("c1", "exon", 32001, 32200, "+", at("g16", "VERSIONED", "i21", "t21")),
        # the attribute parser: a value with a blank keeps its first word; two blanks leave an empty value
        feat("c1", "exon", 40001, 40500, "+", 'gene_id "g17"; gene_name "BLANK NAME"; transcript_id "i22"; transcript_name "t22 x"; note  "two blanks";'),
        # an empty piece between two ';' and a blank behind the last one are passed over (AnnotationUtils L381-382)
        feat("c1", "exon", 41001, 41500, "-", 'gene_id "g18"; ; gene_name "EMPTYPIECE"; transcript_id "i23"; transcript_name "t23"; '),
        # two exons of one start: $Exon.compareTo goes on to the ends (L239), the builder then finds them overlapping
        feat("c1", "exon", 42001, 42100, "+", at("g19", "SAMESTART", "i24", "t24")), feat("c1", "exon", 42001, 42200, "+", at("g19", "SAMESTART", "i24", "t24")),
    ]
    lines += extra
    refs = ["chr1", "chr12", "chrUn", "c1"]
    install_gene_io(j, lines, refs)
    H = j.hooks

    def guava_stream_it(jj, itr):
        lst = JObject("java/util/ArrayList")
        lst.native = []
        while jj.call_virtual(itr, "hasNext", "()Z"):
            lst.native.append(jj.call_virtual(itr, "next", "()Ljava/lang/Object;"))
        return jj.natives["java/util/ArrayList.stream"](jj, lst)

    H["com/google/common/collect/Streams.stream:(Ljava/util/Iterator;)Ljava/util/stream/Stream;"] = guava_stream_it
    H["htsjdk/samtools/util/ProgressLogger.<init>:(Lhtsjdk/samtools/util/Log;ILjava/lang/String;Ljava/lang/String;)V"] = lambda jj, o, *a: None
    H["htsjdk/samtools/util/AbstractProgressLogger.record:(Ljava/lang/String;I)Z"] = lambda jj, o, *a: 0      # (logging only)
    H["htsjdk/samtools/util/CloserUtil.close:(Ljava/lang/Object;)V"] = lambda jj, *a: None
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar + DropseqLib-1.0.jar + picard-2.23.9.jar + htsjdk-4.1.3.jar", "sections": []}
    s = g.section("GTFReader.load() (DropseqLib: $FilteringGTFParser over GTFParser STRICT, GeneFromGTFBuilder) over GTF lines made from the first rows of "
                  "tests/golden/chr12_head1500.refFlat.gz (gene / transcript / exon / CDS / UTR features, GENCODE attribute order) plus the hand-made lines in "
                  "`extra_lines`, then GennameTagger.annotateGene(record) as in ref_exec_gene.json.  HashMap / HashSet iteration in java.util.HashMap table "
                  "order from the keys' own hashCode() (tier D)", GT, "annotateGene:(Lhtsjdk/samtools/SAMRecord;)Lhtsjdk/samtools/SAMRecord;")
    s["gtf_lines"] = lines
    s["extra_lines"] = extra
    s["ref_names"] = refs
    j.hash_order = "jdk"
    rdr = j.new(GTFR, "(Ljava/io/File;Lhtsjdk/samtools/SAMSequenceDictionary;)V", None, JObject("htsjdk/samtools/SAMSequenceDictionary"))
    det = j.call_virtual(rdr, "load", "()Lhtsjdk/samtools/util/OverlapDetector;")
    print(f"  GTF loaded by the reference's GTFReader.load(): {time.time() - g.t0:.0f}s", flush=True)
    allg = j.call_virtual(det, "getAll", "()Ljava/util/Set;")
    genes = []
    for o, _ in allg.native.items_in_insertion_order():
        txs = []
        itr = j.call_virtual(o, "iterator", "()Ljava/util/Iterator;")
        while j.call_virtual(itr, "hasNext", "()Z"):
            t = j.call_virtual(itr, "next", "()Ljava/lang/Object;")
            txs.append({"name": t.f["name"], "tx": [t.f["transcriptionStart"], t.f["transcriptionEnd"]], "cds": [t.f["codingStart"], t.f["codingEnd"]],
                        "exons": [[e.f["start"], e.f["end"]] for e in t.f["exons"].a]})
        genes.append({"name": o.f["name"], "contig": o.f["contig"], "start": o.f["start"], "end": o.f["end"], "negative": bool(o.f["negativeStrand"]),
                      "transcripts_in_iteration_order": txs})
    s["genes_loaded"] = sorted(genes, key=lambda d: d["name"])
    tagger = j.new_object(GT)
    tagger.f.update({"TAG": "GE", "STRANDTAG": "GS", "FUNCTIONTAG": "XF", "ALLOW_MULTI_GENE_READS": 1, "geneOverlapDetector": det})
    tagger.f["metrics"] = j.new(GT + "$ReadTaggingMetric", f"(L{GT};)V", tagger)
    reads = gene_reads(rng, rows, n_reads)
    reads += [("c1", f, p, cg) for f, p, cg in [(0, 40, [("M", 50)]), (0, 95, [("M", 4)]), (0, 100, [("M", 50)]), (0, 250, [("M", 20)]), (0, 350, [("M", 20)]),
              (16, 250, [("M", 20)]), (0, 1100, [("M", 50)]), (0, 1199, [("M", 50)]), (0, 1200, [("M", 50)]), (0, 1250, [("M", 100)]), (16, 2985, [("M", 10)]),
              (16, 3100, [("M", 50)]), (16, 20100, [("M", 50)]), (0, 20100, [("M", 50)]), (16, 20600, [("M", 50)]), (0, 5100, [("M", 50)]), (0, 9100, [("M", 50)]),
              (0, 10050, [("M", 20)]), (0, 11050, [("M", 20)]), (0, 12050, [("M", 20)]), (0, 13050, [("M", 100)]), (0, 15100, [("M", 100)]),
              (0, 30100, [("M", 100)]), (0, 31100, [("M", 100)]), (0, 32050, [("M", 100)]), (0, 40100, [("M", 100)]), (16, 41100, [("M", 100)]), (0, 42050, [("M", 20)]),
              (0, 250, [("M", 20), ("N", 19800), ("M", 100)]), (16, 250, [("M", 20), ("N", 19800), ("M", 100)]), (16, 3100, [("M", 50), ("N", 16900), ("M", 100)])]]
    for k, (ref, flag, pos0, cigar) in enumerate(reads):
        rec = sam_record(j, ref, flag, pos0, cigar)
        case = {"ref": ref, "flag": flag, "pos0": pos0, "cigar": [[op, ln] for op, ln in cigar]}
        try:
            j.call_virtual(tagger, "annotateGene", f"(L{SAMREC};)L{SAMREC};", rec)
            case["set_attribute"] = rec.native["calls"]
        except JavaThrow as e:
            case["throws"] = e.obj.cls
            case["set_attribute_before_throw"] = rec.native["calls"]
        s["cases"].append(case)
        if k % 60 == 0:
            print(f"  gene_gtf {k + 1}/{len(reads)}  {time.time() - g.t0:.0f}s", flush=True)
    out["sections"].append(g.finish(s))
    # ---- lines the STRICT parser rejects: each one alone behind a valid line -> the exception GTFReader.load ends with
    s2 = g.section("GTFReader.load() over one valid line and ONE offending line: the exception it ends with (GTFParser.next L84-97 under ValidationStringency.STRICT, "
                   "AnnotationUtils.parseOptionalFields L386, Integer.parseInt) -- nothing catches it on the way up to UmiFinderWorker", GTFR,
                   "load:()Lhtsjdk/samtools/util/OverlapDetector;")
    ok_line = feat("c1", "exon", 101, 300, "+", at("g1", "GOOD", "i1", "t1"))
    bad = {"no_gene_id": feat("c1", "exon", 1, 9, "+", 'gene_name "X"; transcript_id "i"; transcript_name "t";'),
           "no_gene_name": feat("c1", "exon", 1, 9, "+", 'gene_id "g"; transcript_id "i"; transcript_name "t";'),
           "no_transcript_name": feat("c1", "exon", 1, 9, "+", 'gene_id "g"; gene_name "X"; transcript_id "i";'),
           "no_transcript_id": feat("c1", "exon", 1, 9, "+", 'gene_id "g"; gene_name "X"; transcript_name "t";'),
           "gene_record_without_transcript_is_fine": feat("c1", "gene", 101, 300, "+", 'gene_id "g1"; gene_name "GOOD";'),
           "comma_in_gene_name": feat("c1", "exon", 1, 9, "+", at("g", "A,B", "i", "t")),
           "attribute_without_value": feat("c1", "exon", 1, 9, "+", at("g", "X", "i", "t") + " basic;"),
           "semicolon_inside_quotes": feat("c1", "exon", 1, 9, "+", at("g", "X", "i", "t") + ' note "a;b";'),
           "eight_fields": "\t".join(["c1", "hand", "exon", "1", "9", ".", "+", "."]),
           "start_not_a_number": feat("c1", "exon", "1e3", 2000, "+", at("g", "X", "i", "t")),
           "gene_version_not_a_number": feat("c1", "exon", 1, 9, "+", at("g", "X", "i", "t", ' gene_version "v2";')),
           "invalid_line_on_unknown_contig": feat("c9", "exon", 1, 9, "+", 'gene_name "X"; transcript_id "i"; transcript_name "t";')}
    for name, ln in bad.items():
        install_gene_io(j, [ok_line, ln], refs)
        case = {"what": name, "line": ln}
        try:
            d2 = j.call_virtual(j.new(GTFR, "(Ljava/io/File;Lhtsjdk/samtools/SAMSequenceDictionary;)V", None, JObject("htsjdk/samtools/SAMSequenceDictionary")),
                                "load", "()Lhtsjdk/samtools/util/OverlapDetector;")
            case["genes_loaded"] = sorted(o.f["name"] for o, _ in j.call_virtual(d2, "getAll", "()Ljava/util/Set;").native.items_in_insertion_order())
        except JavaThrow as e:
            case["throws"] = e.obj.cls
        s2["cases"].append(case)
    j.hash_order = None
    out["sections"].append(g.finish(s2))
    return out


# ---------------------------------------------------------------------------------------------------------------------
UBLD = "com/rw/nanoporereadscanner/analyzers/UsedCellBCListGenerator$UsedBarcodesListData"
BDCT = "com/rw/nanoporereadscanner/analyzers/BarcodeDatasetColissionTester"
L2OM = "it/unimi/dsi/fastutil/longs/Long2ObjectMap"
L2OE = "it/unimi/dsi/fastutil/longs/Long2ObjectMap$Entry"
LSET = "it/unimi/dsi/fastutil/longs/LongOpenHashSet"
LES = "com/google/common/util/concurrent/ListeningExecutorService"


def _install_prim2object_iterable(j, map_cls, iface, entry_cls, keyset_cls, box_cls, key_desc, get_key_name, entry_set_name, maps_cls):
    """an absent fastutil <prim>2ObjectOpenHashMap as a keyed store that can ALSO be iterated -- in the varied orders of jvm.hash_order,
    never in fastutil's own (the jar is missing): a case is kept only when every order gives the same answer"""
    import jvm_exec
    from jvm_natives import HashStore

    N = j.natives

    def st(o):
        if o.native is None or isinstance(o.native, dict):
            o.native = HashStore(j)
        return o.native

    box = lambda k: k if isinstance(k, JBox) else JBox(box_cls, k)  # noqa: E731

    def new_map(jj):
        o = JObject(map_cls)
        o.native = HashStore(jj)
        return o

    def alist(items):
        lst = JObject("java/util/ArrayList")
        lst.native = list(items)
        return lst

    def entry(k, v):
        e = JObject(entry_cls)
        e.native = [k, v]
        return e

    def cells(o, what):
        return st(o).cells_for_iteration(what, map_cls)

    def key_set(jj, o):
        ks = JObject(keyset_cls)
        ks.native = o          # live view
        return ks

    for c in (map_cls, iface):
        N[c + ".<new>"] = new_map
        N[c + ".<init>"] = lambda jj, o, *a: None if st(o) is None else None
        N[c + f".put:({key_desc}Ljava/lang/Object;)Ljava/lang/Object;"] = lambda jj, o, k, v: st(o).put(box(k), v)[0]
        N[c + f".get:({key_desc})Ljava/lang/Object;"] = lambda jj, o, k: (st(o).find(box(k)) or [None, None])[1]
        N[c + f".containsKey:({key_desc})Z"] = lambda jj, o, k: 1 if st(o).find(box(k)) is not None else 0
        N[c + f".remove:({key_desc})Ljava/lang/Object;"] = lambda jj, o, k: (st(o).remove(box(k)) or [None, None])[1]
        N[c + ".size"] = lambda jj, o: len(st(o))
        N[c + ".isEmpty"] = lambda jj, o: 0 if len(st(o)) else 1
        N[c + ".values"] = lambda jj, o: alist(cl[1] for cl in cells(o, "values"))
        N[c + "." + entry_set_name] = lambda jj, o: alist(entry(cl[0], cl[1]) for cl in cells(o, entry_set_name))
        N[c + ".keySet"] = key_set
    N[keyset_cls + f".contains:({key_desc})Z"] = lambda jj, ks, k: 1 if st(ks.native).find(box(k)) is not None else 0
    N[keyset_cls + ".contains:(Ljava/lang/Object;)Z"] = lambda jj, ks, k: 1 if st(ks.native).find(k) is not None else 0
    N[keyset_cls + ".stream"] = lambda jj, ks: jj.natives["java/util/ArrayList.stream"](jj, alist(cl[0] for cl in cells(ks.native, "keySet().stream")))
    N[keyset_cls + ".size"] = lambda jj, ks: len(st(ks.native))
    N[entry_cls + "." + get_key_name] = lambda jj, e: e.native[0].v
    N[entry_cls + ".getKey"] = lambda jj, e: e.native[0]
    N[entry_cls + ".getValue"] = lambda jj, e: e.native[1]
    N[maps_cls + ".synchronize"] = lambda jj, m, *a: m
    jvm_exec.JDK_SUPER[map_cls] = "java/lang/Object"
    jvm_exec.JDK_IFACES[map_cls] = [iface, "java/util/Map"]
    jvm_exec.JDK_IFACES[entry_cls] = ["java/util/Map$Entry"]
    jvm_exec.JDK_IFACES[keyset_cls] = ["java/util/Set", "java/util/Collection"]


def install_long2object_iterable(j):
    _install_prim2object_iterable(j, L2O, L2OM, L2OE, LSET, "java/lang/Long", "J", "getLongKey", "long2ObjectEntrySet",
                                  "it/unimi/dsi/fastutil/longs/Long2ObjectMaps")
    import jvm_exec

    jvm_exec.JDK_IFACES[LSET] = ["java/util/Set", "it/unimi/dsi/fastutil/longs/LongSet", "java/util/Collection"]


def install_parallel_as_sequential(j):
    """ClusterOne_MyClustering switches to parallel streams above 30 reads (L176-189).  Their terminal operations there are collects into
    maps and sets, whose CONTENT does not depend on the encounter order; what is read out of those containers afterwards is iterated in the
    varied orders like everything else.  So the parallel stream is run sequentially."""
    N = j.natives
    N["java/util/stream/Stream.parallel"] = lambda jj, st_: st_
    N["java/util/stream/Collectors.groupingByConcurrent"] = N["java/util/stream/Collectors.groupingBy"]
    for k in list(N):
        if k.endswith(".stream") and not k.startswith("java/util/stream/"):
            N[k[:-len(".stream")] + ".parallelStream"] = N[k]


def install_async_as_sync(j):
    """CompletableFuture.runAsync(runnable) of generateDistanceMatrixParalell (>= 70 reads: one task per matrix row, every task writes its
    own row and the transposed cells) run at once on the calling thread; allOf(...).join() then has nothing to wait for"""
    N = j.natives
    CF = "java/util/concurrent/CompletableFuture"

    def run_async(jj, runnable, *a):
        jj.call_fn(jj, runnable) if isinstance(runnable, JLambda) else jj.call_virtual(runnable, "run", "()V")
        return JObject(CF)

    N[CF + ".runAsync"] = run_async
    N[CF + ".allOf"] = lambda jj, arr: JObject(CF)
    N[CF + ".join"] = lambda jj, f: None
    N[CF + ".get"] = lambda jj, f: None


def install_int2object_iterable(j):
    fi = "it/unimi/dsi/fastutil/ints/"
    _install_prim2object_iterable(j, fi + "Int2ObjectOpenHashMap", fi + "Int2ObjectMap", fi + "Int2ObjectMap$Entry", fi + "IntSet$View",
                                  "java/lang/Integer", "I", "getIntKey", "int2ObjectEntrySet", fi + "Int2ObjectMaps")


def install_sync_executor(j):
    """the thread pool of BarcodeDatasetColissionTester run on one thread: submit() calls the Callable at once, Futures.addCallback()
    queues the callback, CountDownLatch.await() runs the queued callbacks (each of which submits the next barcode, L212-227, L240-243)
    until the latch is released.  Submission order = the order of the reference's own deque."""
    N, H = j.natives, j.hooks
    pending = []
    N["java/util/concurrent/Executors.newWorkStealingPool"] = lambda jj, *a: JObject("$Pool")
    H["com/google/common/util/concurrent/MoreExecutors.<clinit>:()V"] = None
    H["com/google/common/util/concurrent/Futures.<clinit>:()V"] = None
    H["com/google/common/util/concurrent/MoreExecutors.listeningDecorator:(Ljava/util/concurrent/ExecutorService;)L" + LES + ";"] = \
        lambda jj, p: JObject("$ListeningPool")
    H["com/google/common/util/concurrent/MoreExecutors.directExecutor:()Ljava/util/concurrent/Executor;"] = lambda jj: None

    def submit(jj, pool, callable_):
        fut = JObject("$DoneFuture")
        try:
            fut.native = ("ok", jj.call_virtual(callable_, "call", "()Ljava/lang/Object;"))
        except JavaThrow as e:
            fut.native = ("err", e.obj)
        return fut

    N["$ListeningPool.submit"] = submit
    N["$ListeningPool.shutdown"] = lambda jj, p: None
    H["com/google/common/util/concurrent/Futures.addCallback:(Lcom/google/common/util/concurrent/ListenableFuture;"
      "Lcom/google/common/util/concurrent/FutureCallback;Ljava/util/concurrent/Executor;)V"] = lambda jj, fut, cb, ex: pending.append((fut, cb))

    def latch_new(jj):
        o = JObject("java/util/concurrent/CountDownLatch")
        o.native = [1]
        return o

    def run_one(jj):
        fut, cb = pending.pop(0)
        if fut.native[0] == "ok":
            jj.call_virtual(cb, "onSuccess", "(Ljava/lang/Object;)V", fut.native[1])
        else:
            jj.call_virtual(cb, "onFailure", "(Ljava/lang/Throwable;)V", fut.native[1])

    def latch_await(jj, o, *a):
        while o.native[0] > 0:
            if not pending:
                raise Unsupported("CountDownLatch.await with nothing left to run")
            run_one(jj)

    def await_termination(jj, p, *a):
        # the latch opens when the first worker finds the deque empty; the tasks still in flight finish (callbacks included, they run on the
        # workers: directExecutor) before awaitTermination returns (L97-98)
        while pending:
            run_one(jj)
        return 1

    N["$ListeningPool.awaitTermination"] = await_termination

    N["java/util/concurrent/CountDownLatch.<new>"] = latch_new
    N["java/util/concurrent/CountDownLatch.<init>"] = lambda jj, o, n: o.native.__setitem__(0, n)
    N["java/util/concurrent/CountDownLatch.await"] = latch_await
    N["java/util/concurrent/CountDownLatch.countDown"] = lambda jj, o: o.native.__setitem__(0, max(0, o.native[0] - 1))
    for k in list(N):
        if k.startswith("java/util/ArrayDeque."):
            N["java/util/concurrent/ConcurrentLinkedDeque" + k[len("java/util/ArrayDeque"):]] = N[k]
    N["java/util/concurrent/ConcurrentLinkedDeque.remove:()Ljava/lang/Object;"] = N["java/util/ArrayDeque.removeFirst"]

    def cld_new(jj):
        o = JObject("java/util/concurrent/ConcurrentLinkedDeque")
        o.native = []
        return o

    N["java/util/concurrent/ConcurrentLinkedDeque.<new>"] = cld_new
    N["java/util/concurrent/TimeUnit.MINUTES"] = lambda jj: JObject("java/util/concurrent/TimeUnit")
    import jvm_exec

    jvm_exec.JDK_IFACES["$ListeningPool"] = [LES, "java/util/concurrent/ExecutorService"]
    jvm_exec.JDK_IFACES["java/util/concurrent/ConcurrentLinkedDeque"] = ["java/util/Deque", "java/util/Queue", "java/util/Collection"]


def gen_finalize(g, n_sets=14, seed=1313):
    """a-13: the end of pass 1"""
    import ref_params

    j = g.j
    rng = random.Random(seed)
    install_long2object_iterable(j)
    install_sync_executor(j)
    files = {}
    install_file_sink(j, files)
    PSHP = "com/rw/nanoporereadscanner/stats/ParseStatsHtmlPrinter"
    BCNT = "com/rw/nanoporereadscanner/analyzers/Parser$BarcodeCounts"
    par, _report = ref_params.load_config(j, PAR)
    rs = par.f["readScannerParameters"]
    wt = "com/rw/parameters/ReadScannerParameters$WHAT_TODO"
    rs.f["what_todo"] = j.natives["java/util/EnumSet.of"](j, j.get_static(wt, "FIND_BARCODES"))   # the run's mode (-g ...)
    # what ReadScannerParameters.validate_readScannerParameters L236-238 does with the shipped config (<mergeBCsED>null) and -e 1
    rs.f["assignCellBCwithEditDistance"] = j.call_static(GOPT, "of", f"(Ljava/lang/Object;)L{GOPT};", JBox("java/lang/Integer", 1))
    rs.f["mergeBCsEdit"] = JBox("java/lang/Integer", 1)
    par.f["general"].f["nCPU"] = JBox("java/lang/Integer", 4)   # -t: only sizes the pool and the first wave of submissions (L72, L87)
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    s = g.section("new UsedBarcodesListData(params); unfilteredUsedBarcodeMap := {barcode -> AtomicInteger(count)}; recordCount := n; "
                  "finalizeData() (UsedCellBCListGenerator.java:L379-425): low-count filter (L359-363), BarcodeDatasetColissionTester "
                  "(L68-229: one BarcodeMatchTester per barcode in collision mode, merge of the colliding barcodes, low-depth cut), and the "
                  "printable list.  Outputs: finalColissionFilteredData as {barcode: count}, usedBarcodesForTSV as the ordered list it is. "
                  "The fastutil map is a keyed store iterated in three different orders per case (`hash_orders_agree`); the thread pool "
                  "runs on one thread", UBLD, "finalizeData:()V")
    s["parameters"] = {"mergeBCsEdit": rs.f["mergeBCsEdit"].v if rs.f.get("mergeBCsEdit") is not None else None,
                       "minCountFold": rs.f["minCountFold"].v, "cellsWithReadsnFoldBelowMaxToKeep": rs.f["cellsWithReadsnFoldBelowMaxToKeep"].v}
    enc = lambda q: j.call_static(TB, "getLongHashForSeq", "([C)J", j.char_array(q))  # noqa: E731
    for idx in range(n_sets):
        n_cells = rng.randrange(6, 22)
        cells = [rnd_seq(rng, 16) for _ in range(n_cells)]
        counts = {}
        big = rng.randrange(400, 4000)
        for c in cells:
            counts[c] = max(2, int(big * rng.random() ** 2) + rng.randrange(2, 12))
        # sequencing-error children of some cells (ed 1: substitution, insertion, deletion), a few unrelated low-count barcodes,
        # homopolymer barcodes (dropped from the TSV when no whitelist is given)
        for c in cells[:max(2, n_cells // 2)]:
            for _ in range(rng.randrange(1, 4)):
                kind = rng.randrange(3)
                p_ = rng.randrange(16)
                if kind == 0:
                    m = c[:p_] + rng.choice([b for b in "ACGT" if b != c[p_]]) + c[p_ + 1:]
                elif kind == 1:
                    m = (c[:p_] + rng.choice("ACGT") + c[p_:])[:16]
                else:
                    m = (c[:p_] + c[p_ + 1:] + rng.choice("ACGT"))[:16]
                if m not in counts:
                    counts[m] = rng.randrange(1, max(3, counts[c] // rng.randrange(3, 40)))
        for _ in range(rng.randrange(2, 8)):
            counts.setdefault(rnd_seq(rng, 16), rng.randrange(1, 6))
        if idx % 3 == 0:
            counts.setdefault("AAAAAAC" + rnd_seq(rng, 9), rng.randrange(20, 200))
        record_count = rng.choice([1, 40, 2500, 12000, 60000])
        items = list(counts.items())
        results = []
        # a list of possible barcodes was given (the TSV keeps every row) or not (rows with AAAAA / TTTTT are dropped, L418-420)
        with_whitelist = idx % 2 == 0
        rs.f["tenXbarcodeWhiteList"] = (j.call_static(GOPT, "of", f"(Ljava/lang/Object;)L{GOPT};", JObject("java/lang/Object")) if with_whitelist
                                        else j.call_static(GOPT, "absent", f"()L{GOPT};"))
        for order in ("insertion", "reverse", ("shuffle", idx + 1)):
            j.hash_order = order
            ud = j.new(UBLD, f"(L{PAR};)V", par)
            m = ud.f["unfilteredUsedBarcodeMap"]
            for q, c in items:
                ai = j.natives["java/util/concurrent/atomic/AtomicInteger.<new>"](j)
                j.natives["java/util/concurrent/atomic/AtomicInteger.<init>"](j, ai, c)
                m.native.put(JBox("java/lang/Long", enc(q)), ai)
            rc = ud.f["recordCount"]
            j.natives["java/util/concurrent/atomic/AtomicInteger.set"](j, rc, record_count)
            try:
                j.invoke(j.find_method(UBLD, "finalizeData", "()V"), [ud])
                fin = ud.f["finalColissionFilteredData"]
                final = sorted([u64(k.v), v.v] for k, v in fin.native.items_in_insertion_order())
                tsv = ud.f["usedBarcodesForTSV"]
                tsv_rows = [[k, v.v] for k, v in tsv.native] if isinstance(tsv.native, list) else [[k, v.v] for k, v in tsv.native.items_in_insertion_order()]
                info = ud.f["infoOnCollidingBarcodesInUsedBCs"]   # TreeMap<ed, Map<barcode, List<colliding barcode>>> (L133)
                coll = {}
                for ed_box, per_bc in info.native.items_in_insertion_order():
                    coll[str(ed_box.v)] = sorted([u64(k.v), sorted(u64(x.v) for x in lst.native)] for k, lst in per_bc.native.items_in_insertion_order())
                # BarcodeList.tsv: ParseStatsHtmlPrinter.writesedBarcodesListTSV(file, data, params) (L235-285), every string handed to the writer
                files.clear()
                fo = JObject("java/io/File")
                fo.native = "BarcodeList.tsv"
                j.call_static(PSHP, "writesedBarcodesListTSV", f"(Ljava/io/File;L{UBLD};L{PAR};)V", fo, ud, par)
                # BarcodesAssigned.tsv: writeAssignedTSV(params, {barcode -> Parser$BarcodeCounts}, file) (L294-327) over pass-2 counters drawn
                # for the barcodes of the final list (addCountForEd per assigned read)
                amap = j.natives["java/util/HashMap.<new>"](j)
                arng = random.Random(1000 * idx + 7)
                assigned = []
                for key, _cnt in final:
                    n0, n1 = arng.choice([0, 1, 3, 900, 1000, 12345, 2000000]), arng.choice([0, 0, 2, 999, 1001])
                    if n0 + n1 == 0:
                        continue
                    bc = j.new(BCNT, "()V")
                    if n0 > 50 or n1 > 50:   # large counters set directly (the loop of addCountForEd calls would only add bytecode steps)
                        for ed, n in ((0, n0), (1, n1)):
                            if n:
                                ai = j.natives["java/util/concurrent/atomic/AtomicInteger.<new>"](j)
                                j.natives["java/util/concurrent/atomic/AtomicInteger.<init>"](j, ai, n)
                                j.call_virtual(bc.f["edCounts"], "put", "(Ljava/lang/Object;Ljava/lang/Object;)Ljava/lang/Object;", JBox("java/lang/Integer", ed), ai)
                        j.natives["java/util/concurrent/atomic/AtomicInteger.set"](j, bc.f["counts"], n0 + n1)
                    else:
                        for ed, n in ((0, n0), (1, n1)):
                            for _ in range(n):
                                j.call_virtual(bc, "addCountForEd", "(I)V", ed)
                    amap.native.put(JBox("java/lang/Long", key if key < 1 << 63 else key - (1 << 64)), bc)
                    assigned.append([key, n0, n1])
                fa = JObject("java/io/File")
                fa.native = "BarcodesAssigned.tsv"
                j.call_static(PSHP, "writeAssignedTSV", f"(L{PAR};Ljava/util/Map;Ljava/io/File;)V", par, amap, fa)
                results.append({"final": final, "tsv": tsv_rows, "collisions": coll, "barcode_list_text": "".join(files["BarcodeList.tsv"]),
                                "assigned": assigned, "assigned_text": "".join(files["BarcodesAssigned.tsv"])})
            except JavaThrow as e:
                results.append({"throws": e.obj.cls, "message": e.obj.f.get("message"), "in": e.trace[:6]})
        j.hash_order = None
        agree_final = all(r.get("final") == results[0].get("final") and r.get("throws") == results[0].get("throws") for r in results[1:])
        agree_final = agree_final and all(r.get("collisions") == results[0].get("collisions") for r in results[1:])
        agree_tsv = all(r.get("tsv") == results[0].get("tsv") for r in results[1:])
        texts = [[r.get("barcode_list_text"), r.get("assigned_text")] for r in results]
        for r in results[1:]:
            r.pop("assigned", None)
        s["cases"].append({"barcodes": [[q, c] for q, c in items], "keys": [u64(enc(q)) for q, _ in items], "record_count": record_count, "with_whitelist": with_whitelist,
                           "hash_orders_agree": agree_final, "tsv_order_agrees": agree_tsv, **results[0],
                           "texts_under_other_orders": [t for t in texts[1:] if t != texts[0]],
                           **({} if agree_final else {"other_orders": results[1:]})})
        print(f"  finalize {idx + 1}/{n_sets}  {time.time() - g.t0:.0f}s  agree={agree_final}/{agree_tsv}", flush=True)
    out["sections"].append(g.finish(s))
    return out


# ---------------------------------------------------------------------------------------------------------------------
RGRP = "com/rw/umifinder/bamreaders/ReadGrouper"
NREAD = "com/rw/umifinder/reads/nanopore/NanoporeRead"
RSD = NREAD + "$ReadScanData"
NCHUNK = "com/rw/umifinder/bamreaders/BamReader$NanoporeReadChunk"


def merge_back_designs(seed=1424, n=48):
    """chunks built so that ClusterList.refineClusters moves reads BACK (ReadGrouper.java:L765-775): a chain whose far-left group is cut
    off as off-centre, which shifts the centre of the rest towards a small right group that was cut off too and now lies within 500 of
    it (and the mirror image, and both strands)"""
    rng = random.Random(seed)
    out = []
    for k in range(n):
        base = rng.randrange(2000, 9000)
        n_left, n_mid, n_right = rng.randrange(4, 6), rng.randrange(12, 17), rng.randrange(2, 4)
        far = rng.randrange(1340, 1396)
        pos = [0 + rng.randrange(0, 6) for _ in range(n_left)] + [rng.randrange(440, 470)] + [900 + rng.randrange(0, 8) for _ in range(n_mid)] + \
              [far + rng.randrange(0, 5) for _ in range(n_right)]
        if k % 2:
            pos = [1400 - p_ for p_ in pos]          # mirror image: the small group on the left
        flag = 16 if k % 4 >= 2 else 0
        reads = [[base + p_, flag] for p_ in sorted(pos)]
        if k % 3 == 0:                               # a second locus on the other strand, and a read without a position
            reads += [[base + 5000 + 3 * q, 16 - flag] for q in range(5)] + [[None, 0]]
            reads.sort(key=lambda r: (r[0] is None, r[0] or 0))
        out.append((reads, k % 2 == 0))
    # The reads a cluster loses as off-centre never come back: the centre it is compared with afterwards is the stale one of the
    # removal pass.  What does move is a cluster exactly 500 from the centre of a larger one: 500 apart ends a chain (< 500 chains)
    # but "within 500 of the centre" (<=) takes the reads over -- a larger cluster with all its reads on one position (or rounding
    # to it) and a smaller one 500 further on; the emptied cluster is then passed over and dropped (L759-761, L780)
    for k in range(24):
        p = rng.randrange(3000, 9000)
        n_x, n_y = rng.randrange(5, 9), rng.randrange(3, 5)
        side = 1 if k % 2 == 0 else -1
        x = [p] * n_x
        if k % 3 == 1:
            x = [p - side] + [p] * n_x          # mean within half a base of p: Math.round((float) mean) is still p
        y = [p + side * 500] * n_y
        if k % 4 == 3:
            y += [p + side * (500 + rng.randrange(1, 40))]   # one read beyond: it stays behind alone and its cluster is dropped (L783)
        if k % 6 == 5:
            y = [p + side * 501] * n_y                       # one base too far: nothing moves
        flag = 16 if k % 4 >= 2 else 0
        reads = [[q, flag] for q in sorted(x + y)]
        if k % 5 == 0:
            reads += [[p + 4000 + 7 * q, flag] for q in range(4)]
        out.append((reads, k % 2 == 1))
    return out


def gen_group2(g):
    return gen_group(g, designed=merge_back_designs())


def gen_group(g, n_chunks=120, seed=1414, designed=None):
    """a-18: genomic-region grouping of one BamReader chunk"""
    j = g.j
    rng = random.Random(seed)
    N, H = j.natives, j.hooks
    put_log = []
    N["java/util/concurrent/LinkedBlockingQueue.put"] = lambda jj, q, c: put_log.append(c)
    N["java/util/concurrent/LinkedBlockingQueue.size"] = lambda jj, q: len(put_log)
    import jvm_exec

    jvm_exec.JDK_IFACES["java/util/concurrent/LinkedBlockingQueue"] = ["java/util/concurrent/BlockingQueue", "java/util/Queue", "java/util/Collection"]
    for lv in ("DEBUG", "TRACE", "ALL", "INFO"):
        N["org/apache/logging/log4j/Level." + lv] = (lambda name: (lambda jj: JObject("org/apache/logging/log4j/Level:" + name)))(lv)
    H[SAMREC + ".<clinit>:()V"] = None
    H[SAMREC + ".getFlags:()I"] = lambda jj, o: o.native["flag"]

    def parallel_sort(jj, arr):
        """Arrays.parallelSort(Comparable[]): a stable merge sort by compareTo (below 8192 elements it IS Arrays.sort; above, sorted runs are
        merged left-first)"""
        import functools

        arr.a.sort(key=functools.cmp_to_key(lambda a, b: jj.call_virtual(a, "compareTo", "(Ljava/lang/Object;)I", b)))

    N["java/util/Arrays.parallelSort:([Ljava/lang/Comparable;)V"] = parallel_sort
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    s = g.section("new ReadGrouper().groupSams(chunk, queue, keepDataEnd) (ReadGrouper.java:L82-230; $Cluster, $ClusterList.refineClusters L455-785) "
                  "with MAX_GENOME_DISTANCE_FOR_SAME_GENOMIC_REGION = 500 (setMaxGenomeDistance): per read the genomic region it was given "
                  "(numbered here in order of first appearance; the reference's ids come from a global counter), the size of the chunk that "
                  "was handed on and of the one carried to the next cycle.  A read = its clustering position (or none) and its strand flag",
                  RGRP, "groupSams:(L...NanoporeReadChunk;Ljava/util/concurrent/BlockingQueue;Z)L...NanoporeReadChunk;")
    j.call_static(RGRP, "setMaxGenomeDistance", "(I)V", 500)
    grouper = j.new(RGRP, "()V")
    for idx in range(len(designed) if designed else n_chunks):
        n = 0 if designed else rng.choice([0, 1, 2, 3, 5, 12, 40, 90, 160, 300])
        reads = []
        base = rng.randrange(1000, 5000)
        while len(reads) < n:                              # loci of 1..25 reads, coordinate-sorted like a BAM chunk
            base += rng.choice([rng.randrange(0, 400), rng.randrange(400, 700), rng.randrange(700, 6000)])
            spread = rng.choice([5, 60, 250, 600])
            for p_ in sorted(rng.randrange(0, spread) for _ in range(min(rng.randrange(1, 26), n - len(reads)))):
                reads.append([None if rng.random() < 0.06 else base + p_, 16 if rng.random() < 0.45 else 0])
            base += spread
        if idx % 5 == 0:
            reads = [[(p if p is None else 2000 + (k // 9) * 3), f] for k, (p, f) in enumerate(reads)]   # dense ties
        keep = idx % 2 == 0
        if designed:
            reads, keep = designed[idx]
        chunk = j.new(NCHUNK, "(I)V", 1)
        objs = []
        for pos, flag in reads:
            rd = j.new_object(NREAD)
            sam = JObject(SAMREC)
            sam.native = {"flag": flag}
            rd.f["sam"] = sam
            sd = j.new_object(RSD)
            sd.f["positionOnGenomeForClustering"] = (j.call_static(GOPT, "of", f"(Ljava/lang/Object;)L{GOPT};", JBox("java/lang/Integer", pos))
                                                     if pos is not None else j.call_static(GOPT, "absent", f"()L{GOPT};"))
            rd.f["readScanData"] = j.call_static(GOPT, "of", f"(Ljava/lang/Object;)L{GOPT};", sd)
            rd.f["genomicRegionNmber"] = j.call_static(GOPT, "absent", f"()L{GOPT};")
            j.call_virtual(chunk, "add", "(Ljava/lang/Object;)Z", rd)
            objs.append(rd)
        del put_log[:]
        q = JObject("java/util/concurrent/LinkedBlockingQueue")
        case = {"reads": reads, "keep_data_end": keep}
        try:
            nxt = j.call_virtual(grouper, "groupSams", f"(L{NCHUNK};Ljava/util/concurrent/BlockingQueue;Z)L{NCHUNK};", chunk, q, 1 if keep else 0)
            ids, regions = {}, []
            for rd in objs:
                o = rd.f["genomicRegionNmber"]
                if j.call_virtual(o, "isPresent", "()Z"):
                    v = j.call_virtual(o, "get", "()Ljava/lang/Object;").v
                    regions.append(ids.setdefault(v, len(ids)))
                else:
                    regions.append(-1)
            case.update({"region": regions, "returned_null": nxt is None, "n_done": len(put_log[0].native) if put_log else None,
                         "n_carried": None if nxt is None else len(nxt.native)})
        except JavaThrow as e:
            case.update({"throws": e.obj.cls, "message": e.obj.f.get("message"), "in": e.trace[:5]})
        s["cases"].append(case)
    out["sections"].append(g.finish(s))
    return out


# ---------------------------------------------------------------------------------------------------------------------
COH = "com/rw/umifinder/analyzers/clustering/ClusterOneHierarchical"
SSTAT = "com/rw/umifinder/scanstats/ScanStats"
ORSS = "com/rw/umifinder/scanstats/OneReadScanStat"
IPAIR = "org/apache/commons/lang3/tuple/ImmutablePair"


def install_intset_iterable(j):
    """the absent fastutil IntOpenHashSet (superclass of OneUmiCluster) as a set that can be iterated -- in the varied orders of
    jvm.hash_order, never in fastutil's own"""
    import jvm_exec
    from jvm_natives import HashStore

    N = j.natives
    IOS = "it/unimi/dsi/fastutil/ints/IntOpenHashSet"
    box = lambda v: v if isinstance(v, JBox) else JBox("java/lang/Integer", v)  # noqa: E731

    def st(o):
        if not isinstance(o.native, HashStore):
            o.native = HashStore(j)
        return o.native

    def alist(items):
        lst = JObject("java/util/ArrayList")
        lst.native = list(items)
        return lst

    keys = lambda o, what: [c[0] for c in st(o).cells_for_iteration(what, IOS)]  # noqa: E731

    def new(jj):
        o = JObject(IOS)
        o.native = HashStore(jj)
        return o

    def remove_all(jj, o, coll):
        items = coll.native if isinstance(coll.native, list) else [c[0] for c in coll.native.order]
        changed = 0
        for v in list(items):
            if st(o).remove(box(v)) is not None:
                changed = 1
        return changed

    N[IOS + ".<new>"] = new
    N[IOS + ".<init>"] = lambda jj, o, *a: None if st(o) is None else None
    N[IOS + ".add"] = lambda jj, o, v: 1 if st(o).put(box(v), True)[1] else 0
    N[IOS + ".contains"] = lambda jj, o, v: 1 if st(o).find(box(v)) is not None else 0
    N[IOS + ".remove"] = lambda jj, o, v: 1 if st(o).remove(box(v)) is not None else 0
    N[IOS + ".removeAll"] = remove_all
    N[IOS + ".size"] = lambda jj, o: len(st(o))
    N[IOS + ".isEmpty"] = lambda jj, o: 0 if len(st(o)) else 1
    N[IOS + ".stream"] = lambda jj, o: jj.natives["java/util/ArrayList.stream"](jj, alist(keys(o, "stream")))
    N[IOS + ".iterator"] = lambda jj, o: jj.natives["java/util/ArrayList.iterator"](jj, alist(keys(o, "iterator")))
    N[IOS + ".forEach"] = lambda jj, o, f: [jj.call_fn(jj, f, k) for k in keys(o, "forEach")] and None
    jvm_exec.JDK_SUPER[IOS] = "java/util/AbstractSet"
    jvm_exec.JDK_IFACES[IOS] = ["java/util/Set", "java/util/Collection", "it/unimi/dsi/fastutil/ints/IntSet", "java/lang/Iterable"]


COM = "com/rw/umifinder/analyzers/clustering/ClusterOne_MyClustering"


def cluster_once(j, side, par, stats, names, order, cls=None):
    """one ClusterOneHierarchical.call() (or ClusterOne_MyClustering.call()) over the reads behind `names` under one iteration order ->
    per read its setAttribute calls"""
    cls = cls or COH
    j.hash_order = order
    reads = []
    for nm in names:
        sd = side.scan_data(nm)
        r = side.result_for(sd)
        sam = JObject(SAMREC)
        sam.native = {"calls": []}
        r.f["nanoporeRead"].f["sam"] = sam
        r.f["oneReadScanStats"] = j.new(ORSS, "()V") if ("<init>", "()V") in j.load(ORSS).methods else j.new_object(ORSS)
        reads.append(r)
    lst = JObject("java/util/ArrayList")
    lst.native = list(reads)
    pair = j.call_static(IPAIR, "of", f"(Ljava/lang/Object;Ljava/lang/Object;)L{IPAIR};", JBox("java/lang/Boolean", 0), lst)
    try:
        c = j.new(cls, f"(L{UPAR};L{IPAIR};L{SSTAT};)V", par, pair, stats)
        j.call_virtual(c, "call", f"()L{IPAIR};")
        return [r.f["nanoporeRead"].f["sam"].native["calls"] for r in reads]
    except JavaThrow as e:
        return {"throws": e.obj.cls, "message": e.obj.f.get("message"), "in": e.trace[:8]}
    finally:
        j.hash_order = None


def gen_cluster(g, n_groups=70, seed=1515, five_prime=False):
    """a-17: UMI clustering of one (cell, region) group"""
    j = g.j
    rng = random.Random(seed)
    side = UmiSide(g, five_prime)
    par = side.par
    install_intset_iterable(j)
    H = j.hooks
    H[SAMREC + ".<clinit>:()V"] = None
    H[SAMREC + ".setAttribute:(Ljava/lang/String;Ljava/lang/Object;)V"] = lambda jj, o, t, v: o.native["calls"].append([t, v])
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar + Aliasi_ClusteringLib-1.0.jar", "sections": []}
    s = g.section("new ClusterOneHierarchical(params, ImmutablePair.of(false, reads), scanStats).call() (ClusterOneHierarchical.java:L61-217) for the "
                  "reads of one (cell, region) group: DistanceMatrix, LingPipe CompleteLinkClusterer / Dendrogram.partitionDistance, "
                  "OneUmiCluster centre, ClusterOneBase.setSamflagsAndStatsForClustered: per read the setAttribute calls made on its record.  "
                  "Each group under eight iteration orders of the hash containers (JDK sets and the stand-in for fastutil's IntOpenHashSet; the shuffled "
                  "orders differ from container to container); kept when they agree", COH, "call:()L...ImmutablePair;")
    # the statistics objects of the run: their counters are not outputs of this path -- atomics created, the per-flag tables inert
    stats = j.new_object(SSTAT)
    for fname, fdesc in j.load(SSTAT).instance_fields:
        if fdesc in ("Ljava/util/concurrent/atomic/AtomicLong;", "Ljava/util/concurrent/atomic/AtomicInteger;"):
            stats.f[fname] = j.natives[fdesc[1:-1] + ".<new>"](j)
    ASS = "com/rw/umifinder/scanstats/AllSamScanStats"
    H[ASS + ".<clinit>:()V"] = None
    H[ASS + ".*"] = lambda jj, *a: None
    stats.f["scanStatsForSams"] = JObject(ASS)
    for idx in range(n_groups):
        n_mol = rng.randrange(1, 5)
        umis = [rnd_seq(rng, 12) for _ in range(n_mol)]
        bc = rnd_seq(rng, 16)
        names = []
        k = 0
        for u in umis:
            for _ in range(rng.randrange(1, 6)):
                uu = mutate(rng, u, rng.choice([0, 0, 0, 1, 1, 2]))[:12].ljust(12, "A")
                names.append(fake_name(rng, 100 * idx + k, five_prime, bc, uu, rng.random() < 0.5, rng.randrange(120, 200), rng.choice([0, 0, 0, 1, -1])))
                k += 1
        results = [cluster_once(j, side, par, stats, names, order)
                   for order in ("insertion", "reverse") + tuple(("shuffle", 977 * idx + 5 + 31 * t) for t in range(6))]
        j.hash_order = None
        distinct = []
        for r in results:
            if r not in distinct:
                distinct.append(r)
        case = {"names": names, "hash_orders_agree": len(distinct) == 1, "set_attribute": results[0]}
        if len(distinct) > 1:
            case["outcomes_over_orders"] = distinct   # what the reference writes depends on a hash order here: every answer that was seen
        s["cases"].append(case)
        print(f"  cluster {idx + 1}/{n_groups} n={len(names)} agree={s['cases'][-1]['hash_orders_agree']}  {time.time() - g.t0:.0f}s", flush=True)
    out["sections"].append(g.finish(s))
    return out


# ---------------------------------------------------------------------------------------------------------------------
UCG = "com/rw/nanoporereadscanner/analyzers/UsedCellBCListGenerator"
RCHUNK = "com/rw/nanoporereadscanner/readerwriter/FastqFileReader$ReadChunk"


def gen_pass1(g, n_reads=60, seed=1616, five_prime=False, no_whitelist=False):
    """a-12: the pass-1 worker of scanfastq over one chunk.  no_whitelist: `-a none` (generateUsedBarcodesWithoutWhitelist): allPossibleBarcodes is
    null, every barcode cut from a read that passes the filter is counted (UsedCellBCListGenerator.java:L255-256) -- every third read gets an N
    inside its barcode, so the map also shows what key such a barcode is counted under"""
    j = g.j
    rng = random.Random(seed)
    p2 = Pass2(g, five_prime, 1)
    install_long2object_iterable(j)
    par = p2.par
    enc = lambda q: j.call_static(TB, "getLongHashForSeq", "([C)J", j.char_array(q))  # noqa: E731
    bcs = sorted({rnd_seq(rng, 16) for _ in range(30)})
    whitelist = bcs[:24] + [rnd_seq(rng, 16) for _ in range(10)]        # six of the planted barcodes are NOT possible barcodes
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    s = g.section("pass 1 of scanfastq over one chunk: PolyATadapterAnalyzer.search with the complete adapter (as pass 1 scans), then "
                  "new UsedCellBCListGenerator$Worker(generator, chunk).call() (UsedCellBCListGenerator.java:L189-263): per record the "
                  "quality filter lambda$call$0 (L198-202), then the barcode cut from the stranded read, its membership in the list of possible "
                  "barcodes and the counter map.  Outputs: per record `filter`, and the final unfilteredUsedBarcodeMap", UCG + "$Worker", "call:()Ljava/lang/Object;")
    s["whitelist"] = None if no_whitelist else whitelist
    s["five_prime"] = five_prime
    reads = []
    for idx in range(n_reads):
        seq, qual, bc = synth_read(rng, bcs, five_prime, idx)
        if no_whitelist and (idx % 3 == 0 or (five_prime and idx % 3 == 1)):          # an N inside the barcode, wherever the (possibly damaged, possibly reversed) copy still stands
            for probe in (bc, revcomp_str(bc)):
                at = seq.find(probe)
                if at >= 0:
                    k = at + rng.randrange(16)
                    seq = seq[:k] + "N" + seq[k + 1:]
                    break
        if idx % 4 == 1:   # low qualities over the barcode or over the whole read
            qual = "".join(chr(33 + rng.randrange(2, 9)) for _ in qual)
        elif idx % 4 == 2:
            qual = "".join(chr(33 + rng.randrange(6, 14)) for _ in qual)
        reads.append((f"read{idx:04d} runid=abc ch={idx % 512}", seq, qual, bc))

    def run(order):
        j.hash_order = order
        gen = j.new_object(UCG)
        gen.f["params"] = par
        for fname, fdesc in j.load(UCG).instance_fields:
            if fdesc == "Ljava/util/concurrent/atomic/AtomicInteger;":
                gen.f[fname] = j.natives["java/util/concurrent/atomic/AtomicInteger.<new>"](j)
        dbg = j.new_object(UCG + "$DebugInfo")
        for fname, fdesc in j.load(UCG + "$DebugInfo").instance_fields:
            if fdesc in ("Ljava/util/concurrent/atomic/AtomicInteger;", "Ljava/util/concurrent/atomic/AtomicLong;"):
                dbg.f[fname] = j.natives[fdesc[1:-1] + ".<new>"](j)
        gen.f["debugInfo"] = dbg
        wl_map = j.natives[L2O + ".<new>"](j)
        for q in whitelist:
            wl_map.native.put(JBox("java/lang/Long", enc(q)), True)
        gen.f["allPossibleBarcodes"] = None if no_whitelist else j.natives[L2O + ".keySet"](j, wl_map)   # LongOpenHashSet stand-in: contains only
        ud = j.new(UBLD, f"(L{PAR};)V", par)
        gen.f["barcodesUsedData"] = ud
        fqs, cases = [], []
        for name, seq, qual, bc in reads:
            fq = p2.record(name, seq, qual)
            case = {"name": name, "seq": seq, "qual": qual, "planted_barcode": bc}
            try:
                p2.process(None, fq, pass2=False)
                fqs.append(fq)
                case["scanned"] = True
            except JavaThrow as e:
                case.update({"scanned": False, "throws": e.obj.cls})
            cases.append(case)
        lst = JObject("java/util/ArrayList")
        lst.native = list(fqs)
        chunk = j.new_object(RCHUNK)
        chunk.f["fastqRecords"] = lst
        w = j.new(UCG + "$Worker", f"(L{UCG};L{RCHUNK};)V", gen, chunk)
        flt = j.find_method(UCG + "$Worker", "lambda$call$0", f"(L{FQX};)Z")
        k = 0
        for case in cases:
            if case["scanned"]:
                try:
                    case["filter"] = bool(j.invoke(flt, [w, fqs[k]]))
                except JavaThrow as e:
                    case["filter_throws"] = e.obj.cls
                k += 1
        j.call_virtual(w, "call", "()Ljava/lang/Object;")
        m = ud.f["unfilteredUsedBarcodeMap"]
        hist = sorted([u64(kk.v), j.natives["java/util/concurrent/atomic/AtomicInteger.get"](j, v)] for kk, v in m.native.items_in_insertion_order())
        rc = j.natives["java/util/concurrent/atomic/AtomicInteger.get"](j, ud.f["recordCount"])
        j.hash_order = None
        return cases, hist, rc

    runs = [run(order) for order in ("insertion", "reverse", ("shuffle", 3))]
    s["cases"], s["histogram"], s["record_count"] = runs[0]
    s["hash_orders_agree"] = all(r == runs[0] for r in runs[1:])
    out["sections"].append(g.finish(s))
    return out


def own_designs(seed=1727):
    """groups above 100 reads built for the two branches of ClusterOne_MyClustering.call no random group reached:
    (0) a cluster more than 50 x smaller than the largest one is discarded (foldDepthBelowMaxDiscardForClustering, L79-82): 104 reads of
        one UMI and 2 of another;
    (1) members farther than 2 from the centre are ejected and clustered once more with the unclustered reads (L102-112): 40 reads of U,
        60 of U' (one substitution away) and 3 of M, two substitutions from U and three from U' -- the owner key is a U read (largest
        neighbourhood), the centre a U' read (least sum of squares), M is 3 from it"""
    rng = random.Random(seed)
    u = rnd_seq(rng, 12)
    other = "".join({"A": "C", "C": "G", "G": "T", "T": "A"}[c] for c in u)
    g0 = [u] * 96 + [mutate(rng, u, 1)[:12].ljust(12, "A") for _ in range(8)] + [other] * 2
    u1 = list(u)
    u1[3] = {"A": "C", "C": "G", "G": "T", "T": "A"}[u1[3]]
    m = list(u)
    for q in (7, 9):
        m[q] = {"A": "G", "C": "T", "G": "A", "T": "C"}[m[q]]
    g1 = [u] * 40 + ["".join(u1)] * 60 + ["".join(m)] * 3
    return [g0, g1]


def gen_cluster_own2(g):
    return gen_cluster_own(g, seed=1727, designed=own_designs())


def gen_cluster_own(g, seed=1717, designed=None):
    """a-17, the other clusterer: ClusterOne_MyClustering, which the dispatch of UmiClustering$Submitter.lambda$run$2 (L239-261) takes for
    groups of more than 100 reads"""
    j = g.j
    rng = random.Random(seed)
    side = UmiSide(g, False)
    par = side.par
    install_intset_iterable(j)
    install_int2object_iterable(j)
    install_parallel_as_sequential(j)
    install_async_as_sync(j)
    H = j.hooks
    H[SAMREC + ".<clinit>:()V"] = None
    H[SAMREC + ".setAttribute:(Ljava/lang/String;Ljava/lang/Object;)V"] = lambda jj, o, t, v: o.native["calls"].append([t, v])
    stats = j.new_object(SSTAT)
    for fname, fdesc in j.load(SSTAT).instance_fields:
        if fdesc in ("Ljava/util/concurrent/atomic/AtomicLong;", "Ljava/util/concurrent/atomic/AtomicInteger;"):
            stats.f[fname] = j.natives[fdesc[1:-1] + ".<new>"](j)
    ASS = "com/rw/umifinder/scanstats/AllSamScanStats"
    H[ASS + ".<clinit>:()V"] = None
    H[ASS + ".*"] = lambda jj, *a: None
    stats.f["scanStatsForSams"] = JObject(ASS)
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    s = g.section("new ClusterOne_MyClustering(params, ImmutablePair.of(false, reads), scanStats).call() (ClusterOne_MyClustering.java:L59-219): "
                  "per read the setAttribute calls made on its record.  Groups of 30-45 reads (the class itself; the shipped dispatch only sends "
                  "it groups above 100 reads) and one group of 104 reads (parallel distance matrix and parallel streams run on one thread).  "
                  "Each group under several iteration orders of the hash containers; all answers seen are recorded", COM, "call:()L...ImmutablePair;")
    sizes = [len(d) for d in designed] if designed else [8, 31, 36, 44, 40, 33, 104]
    for idx, n in enumerate(sizes):
        umis = [rnd_seq(rng, 12) for _ in range(max(2, n // 5))]
        bc = rnd_seq(rng, 16)
        names = []
        for k in range(n):
            if designed:
                names.append(fake_name(rng, 1000 * idx + k, False, bc, designed[idx][k], rng.random() < 0.5, rng.randrange(120, 200), 0))
                continue
            u = umis[min(int(rng.random() ** 2 * len(umis)), len(umis) - 1)]
            uu = mutate(rng, u, rng.choice([0, 0, 0, 1, 1, 2]))[:12].ljust(12, "A")
            names.append(fake_name(rng, 1000 * idx + k, False, bc, uu, rng.random() < 0.5, rng.randrange(120, 200), rng.choice([0, 0, 0, 1, -1])))
        orders = ("insertion", "reverse") + (() if designed else tuple(("shuffle", 577 * idx + 3 + 31 * t) for t in range(1 if n > 100 else 4)))
        results = [cluster_once(j, side, par, stats, names, order, cls=COM) for order in orders]
        distinct = []
        for r in results:
            if r not in distinct:
                distinct.append(r)
        case = {"names": names, "n_orders": len(orders), "hash_orders_agree": len(distinct) == 1, "set_attribute": results[0]}
        if len(distinct) > 1:
            case["outcomes_over_orders"] = distinct
        s["cases"].append(case)
        print(f"  own clusterer {idx + 1}/{len(sizes)} n={n} distinct={len(distinct)}  {time.time() - g.t0:.0f}s", flush=True)
    out["sections"].append(g.finish(s))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# assignumis: <out>.genecounts.tsv / <out>.UMIdepths.tsv (GeneCounts.updateGeneCounts L375-491, printCountTable L307-357,
# printUmisPerCellTable L256-284, mergeGeneCounts L540-592)
# ---------------------------------------------------------------------------------------------------------------------
GCNT = "com/rw/umifinder/scanstats/GeneCounts"
ORSTAT = "com/rw/umifinder/scanstats/OneReadScanStat"


def install_genecounts_io(j, files):
    """INPUT / OUTPUT plumbing of the GeneCounts fixture: a SAMRecord is a bag of the six values updateGeneCounts asks for (its Cigar is
    htsjdk's own, decoded from text by TextCigarCodec); the file classes are those of install_file_sink"""
    H = j.hooks
    H[SAMREC + ".<clinit>:()V"] = None
    install_file_sink(j, files)
    H[SAMREC + ".getReadUnmappedFlag:()Z"] = lambda jj, o: 1 if o.native["flag"] & 4 else 0
    H[SAMREC + ".isSecondaryOrSupplementary:()Z"] = lambda jj, o: 1 if o.native["flag"] & 0x900 else 0
    H[SAMREC + ".getMappingQuality:()I"] = lambda jj, o: o.native["mapq"]
    H[SAMREC + ".getCigar:()Lhtsjdk/samtools/Cigar;"] = lambda jj, o: o.native["cigar"]
    H[SAMREC + ".getReadNegativeStrandFlag:()Z"] = lambda jj, o: 1 if o.native["flag"] & 16 else 0
    H[SAMREC + ".getStringAttribute:(Ljava/lang/String;)Ljava/lang/String;"] = lambda jj, o, t: o.native["attrs"].get(t)
    H[ORSTAT + ".incrementRecordsForReadsUsedInGeneCounts:()V"] = lambda jj, o: o.native.__setitem__(0, o.native[0] + 1)


def install_file_sink(j, files):
    """OUTPUT plumbing: File / FileOutputStream / BufferedOutputStream / PrintStream / FileWriter / BufferedWriter collect what is written
    into files[path] (a list of strings); System.gc and the Runtime memory figures (printed to stdout only) are inert"""
    N = j.natives

    def file_init(jj, o, *a):
        o.native = a[0] if len(a) == 1 else a[0].native + "/" + a[1]

    for c in ("java/io/File", "java/io/FileOutputStream", "java/io/BufferedOutputStream", "java/io/PrintStream"):
        N[c + ".<new>"] = (lambda cc: lambda jj: JObject(cc))(c)
    N["java/io/File.<init>"] = file_init
    N["java/io/File.getPath"] = lambda jj, o: o.native
    N["java/io/FileOutputStream.<init>"] = lambda jj, o, f, *a: setattr(o, "native", files.setdefault(f.native if isinstance(f, JObject) else f, []))
    N["java/io/BufferedOutputStream.<init>"] = lambda jj, o, inner, *a: setattr(o, "native", inner.native)
    N["java/io/PrintStream.<init>"] = lambda jj, o, inner, *a: setattr(o, "native", inner.native)

    def ps_append(jj, o, text, *a):
        o.native.append(jj.to_jstring(text))
        return o

    for c in ("java/io/FileWriter", "java/io/BufferedWriter"):
        N[c + ".<new>"] = (lambda cc: lambda jj: JObject(cc))(c)
    N["java/io/FileWriter.<init>"] = lambda jj, o, f, *a: setattr(o, "native", files.setdefault(f.native if isinstance(f, JObject) else f, []))
    N["java/io/BufferedWriter.<init>"] = lambda jj, o, inner, *a: setattr(o, "native", inner.native)
    N["java/io/BufferedWriter.write:(Ljava/lang/String;)V"] = lambda jj, o, text: o.native.append(jj.to_jstring(text))
    N["java/io/Writer.write:(Ljava/lang/String;)V"] = N["java/io/BufferedWriter.write:(Ljava/lang/String;)V"]
    N["java/io/BufferedWriter.newLine"] = lambda jj, o: o.native.append("\n")
    N["java/io/BufferedWriter.close"] = lambda jj, o: None
    N["java/io/BufferedWriter.flush"] = lambda jj, o: None
    N["java/io/PrintStream.append"] = ps_append
    N["java/io/PrintStream.flush"] = lambda jj, o: None
    N["java/io/PrintStream.close"] = lambda jj, o: None
    N["java/lang/System.gc"] = lambda jj: None
    rt = JObject("java/lang/Runtime")
    N["java/lang/Runtime.getRuntime"] = lambda jj: rt
    for m in ("freeMemory", "totalMemory", "maxMemory"):
        N["java/lang/Runtime." + m] = lambda jj, o: 0


def gen_genecounts(g, seed=1818):
    j = g.j
    rng = random.Random(seed)
    files = {}
    install_long2object_iterable(j)
    install_parallel_as_sequential(j)
    install_genecounts_io(j, files)
    N = j.natives
    for c in (L2O, L2OM):                                 # Long2ObjectFunction / Map defaults the stand-in did not need before
        N[c + ".putIfAbsent:(JLjava/lang/Object;)Ljava/lang/Object;"] = \
            lambda jj, o, k, v: (lambda cell: cell[1] if cell is not None and cell[1] is not None else (o.native.put(JBox("java/lang/Long", k), v), None)[1])(o.native.find(JBox("java/lang/Long", k)))
        N[c + ".entrySet"] = N[c + ".long2ObjectEntrySet"]
        N[c + ".containsKey:(Ljava/lang/Object;)Z"] = lambda jj, o, k: 1 if o.native.find(k) is not None else 0
        N[c + ".get:(Ljava/lang/Object;)Ljava/lang/Object;"] = lambda jj, o, k: (o.native.find(k) or [None, None])[1]
        N[c + ".put:(Ljava/lang/Object;Ljava/lang/Object;)Ljava/lang/Object;"] = lambda jj, o, k, v: o.native.put(k, v)[0]
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar + TwoFourBitNucAcidLibraryMaven-1.0.jar + htsjdk-4.1.3.jar (Cigar)", "sections": []}
    s = g.section("new GeneCounts(); updateGeneCounts(result, params, null) per record (GeneCounts.java:L375-491: the record's flag / mapping "
                  "quality / CIGAR clip test, BC and U8 attributes, geneNames, genomicRegionNmber, isNthRecordForRead), then printCountTable(file, "
                  "params, null) and printUmisPerCellTable(file, params): the text appended to each file.  Every case under six iteration orders of "
                  "the hash containers (ConcurrentHashMap, the stand-in for fastutil's Long2ObjectOpenHashMap; parallel streams run sequentially): "
                  "`texts` lists the distinct outcomes -- rows of equal totals come in container order", GCNT,
                  "updateGeneCounts:(L...OneNanoporeResult;L...ParametersBarcodeUMiFinderAppParams;L...AllReadsScanStats;)V")
    orders = ("insertion", "reverse") + tuple(("shuffle", 1009 * k + 3) for k in range(4))

    def draw_molecules(n, cells, n_genes, n_regions):
        mols = []                                           # (gene index or None, region or None, cell, umi)
        for _ in range(n):
            gi = rng.randrange(n_genes) if rng.random() < 0.8 else None
            mols.append((gi, rng.randrange(n_regions) if rng.random() < 0.9 else None, rng.choice(cells), rnd_seq(rng, 12)))
        return mols

    def draw_records(n_rec, mols, genes):
        recs = []
        for k in range(n_rec):
            gi, reg, cell, umi = rng.choice(mols)
            flag = (16 if rng.random() < 0.5 else 0) | (4 if rng.random() < 0.03 else 0) | (0x100 if rng.random() < 0.04 else 0) | \
                   (0x800 if rng.random() < 0.04 else 0)
            clip_l = rng.choice([0, 0, 0, 20, 150, 151, 400])
            clip_r = rng.choice([0, 0, 0, 20, 150, 151, 400])
            cig = (f"{clip_l}{rng.choice('SH')}" if clip_l else "") + f"{rng.randrange(50, 900)}M" + (f"{clip_r}{rng.choice('SH')}" if clip_r else "")
            has_bc, has_umi = rng.random() < 0.95, rng.random() < 0.93
            recs.append({"gene": None if gi is None else genes[gi], "region": reg, "bc": cell if has_bc else None, "u8": umi if has_umi else None,
                         "flag": flag, "mapq": rng.choice([0, 1, 30, 60, 60, 60]), "cigar": cig, "nth": 1 if rng.random() < 0.12 else 0})
        return recs

    def fill(recs, par):
        gc = j.new(GCNT, "()V")
        stats_calls = [0]
        for r in recs:
            sam = JObject(SAMREC)
            cg = j.call_static("htsjdk/samtools/TextCigarCodec", "decode", "(Ljava/lang/String;)Lhtsjdk/samtools/Cigar;", r["cigar"])
            sam.native = {"flag": r["flag"], "mapq": r["mapq"], "cigar": cg, "attrs": {k: v for k, v in (("BC", r["bc"]), ("U8", r["u8"])) if v is not None}}
            nr = j.new_object(NREAD)
            nr.f["sam"] = sam
            nr.f["geneNames"] = None if r["gene"] is None else JArray("Ljava/lang/String;", [r["gene"]])
            nr.f["genomicRegionNmber"] = (j.call_static(GOPT, "absent", f"()L{GOPT};") if r["region"] is None else
                                          j.call_static(GOPT, "of", f"(Ljava/lang/Object;)L{GOPT};", JBox("java/lang/Long", r["region"])))
            o = j.new_object(ONR)
            o.f["nanoporeRead"] = nr
            o.f["isNthRecordForRead"] = r["nth"]
            st = JObject(ORSTAT)
            st.native = stats_calls
            o.f["oneReadScanStats"] = st
            j.call_virtual(gc, "updateGeneCounts", f"(L{ONR};L{UPAR};Lcom/rw/umifinder/scanstats/AllReadsScanStats;)V", o, par, None)
        return gc, stats_calls[0]

    def tables(gc, par):
        files.clear()
        fa, fb = JObject("java/io/File"), JObject("java/io/File")
        fa.native, fb.native = "out.genecounts.tsv", "out.UMIdepths.tsv"
        j.call_virtual(gc, "printCountTable", f"(Ljava/io/File;L{UPAR};Lcom/rw/umifinder/scanstats/AllReadsScanStats;)V", fa, par, None)
        j.call_virtual(gc, "printUmisPerCellTable", f"(Ljava/io/File;L{UPAR};)V", fb, par)
        return {"genecounts": "".join(files["out.genecounts.tsv"]), "umidepths": "".join(files["out.UMIdepths.tsv"]),
                "recordsWithGene": gc.f["recordsWithGene"], "recordsWithGeneSkippedClipping": gc.f["recordsWithGeneSkippedClipping"]}

    for case, five_prime in enumerate([False, False, True, False, True, False]):
        side = UmiSide(g, five_prime)
        par = side.par
        n_rec = [12, 60, 60, 400, 400, 1500][case]
        n_cells, n_genes, n_regions = [(2, 2, 3), (4, 5, 8), (4, 5, 8), (12, 20, 40), (12, 20, 40), (30, 40, 100)][case]
        cells = [rnd_seq(rng, 16) for _ in range(n_cells)]
        genes = [f"GENE{k}" if k % 3 else f"Gm{k}.{k % 7}" for k in range(n_genes)]
        recs = draw_records(n_rec, draw_molecules(max(2, n_rec // 3), cells, n_genes, n_regions), genes)
        texts = []
        for order in (orders if case < 3 else orders[:3]):
            j.hash_order = order
            try:
                gc, n_calls = fill(recs, par)
                t = tables(gc, par)
                t["incrementRecordsForReadsUsedInGeneCounts"] = n_calls
                if t not in texts:
                    texts.append(t)
            finally:
                j.hash_order = None
        s["cases"].append({"five_prime": five_prime, "records": recs, "texts": texts})
        print(f"  genecounts case {case}: {n_rec} records, {len(texts)} distinct outcome(s)  {time.time() - g.t0:.0f}s", flush=True)
    # GeneCounts.mergeGeneCounts(List.of(a, b, c)) (L540-592): three objects filled from record sets over the same molecules (so genes, cells
    # and UMIs recur), then the two tables of the merged object
    side = UmiSide(g, False)
    par = side.par
    cells = [rnd_seq(rng, 16) for _ in range(5)]
    genes = [f"GENE{k}" for k in range(6)]
    mols = draw_molecules(40, cells, 6, 10)
    parts = [draw_records(n, mols, genes) for n in (120, 80, 150)]
    texts = []
    for order in orders:
        j.hash_order = order
        try:
            lst = JObject("java/util/ArrayList")
            lst.native = [fill(p, par)[0] for p in parts]
            merged = j.call_static(GCNT, "mergeGeneCounts", f"(Ljava/util/List;)L{GCNT};", lst)
            t = tables(merged, par)
            if t not in texts:
                texts.append(t)
        finally:
            j.hash_order = None
    s["merged"] = {"five_prime": False, "parts": parts, "texts": texts}
    print(f"  genecounts merged: {len(texts)} distinct outcome(s)  {time.time() - g.t0:.0f}s", flush=True)
    out["sections"].append(g.finish(s))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# assignumis: the tags of one record before clustering (OneNanoporeSeqAnalyzer.call L95, L146; UmiFinderWorker.lambda$new$0 L248-255)
# ---------------------------------------------------------------------------------------------------------------------
OTAGS = "com/rw/umifinder/flags/OutputSAMtags"


def gen_samtags(g, seed=1919):
    j = g.j
    rng = random.Random(seed)
    H = j.hooks
    H[SAMREC + ".<clinit>:()V"] = None
    H[SAMREC + ".setAttribute:(Ljava/lang/String;Ljava/lang/Object;)V"] = lambda jj, o, t, v: o.native["calls"].append([t, v])
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar", "sections": []}
    for five_prime, files in ((False, ("pass2_3p", "pass2_3p_ed2")), (True, ("pass2_5p", "pass2_5p_polya"))):
        side = UmiSide(g, five_prime)
        par = side.par
        s = g.section(("5-prime (-p)" if five_prime else "3-prime") + " assignumis, per read name: NanoporeRead$ReadScanData.generateReadScanData's "
                      "parse (FastqRecordExt.getScanDatFromReadName), then what OneNanoporeSeqAnalyzer.call does with it: writeSamFlags(sam, "
                      "params.samFlags) (ReadScanResult.java:L205-237) and writeBCSamFlags(sam, params.samFlags, false, false) (L254-279) -> the "
                      "setAttribute calls in order (tag, value; a boxed Integer is a number, a String a string); and "
                      "OneNanoporeResult.getPostBCUMIseq(result, params) = the U7 value UmiFinderWorker.lambda$new$0 sets for a read with a "
                      "barcode that clustering left without a UMI.  Names: the records the reference's own pass 2 wrote "
                      "(ref_exec_pass2_*.json, up to the blank) and synthetic ones around them", RSD,
                      "writeSamFlags:(Lhtsjdk/samtools/SAMRecord;L...OutputSAMtags;)V")
        names = []
        for f in files:
            d = json.load(open(os.path.join(GOLD, f"ref_exec_{f}.json")))
            names += [c["result"]["written"]["name"].split(" ")[0] for c in d["sections"][0]["cases"] if "written" in c["result"]]
        for k in range(40):
            bc, umi = rnd_seq(rng, 16), rnd_seq(rng, 12)
            names.append(fake_name(rng, 500 + k, five_prime, bc, umi, rng.random() < 0.5, rng.randrange(60, 300), rng.choice([0, 0, 1, -1, 2, -2]), ed=rng.choice([0, 1, 2])))
        names += ["plain_read_name", "x_FWD_AE=40_X=ACGT_Q=9_1", "y_REV_PS=5_PE=9_AE=40_T=12_X=ACGTACGT_Q=11.5_2b"]
        for nm in names:
            case = {"name": nm}
            try:
                sd = side.scan_data(nm)
                if sd is None:
                    case["scan_data"] = None
                else:
                    sam = JObject(SAMREC)
                    sam.native = {"calls": []}
                    j.call_virtual(sd, "writeSamFlags", f"(L{SAMREC};L{OTAGS};)V", sam, par.f["samFlags"])
                    n_first = len(sam.native["calls"])
                    j.call_virtual(sd, "writeBCSamFlags", f"(L{SAMREC};L{OTAGS};ZZ)V", sam, par.f["samFlags"], 0, 0)
                    box = lambda v: v.v if isinstance(v, JBox) else v  # noqa: E731
                    case["calls"] = [[t, box(v), "int" if isinstance(v, JBox) else "str"] for t, v in sam.native["calls"]]
                    case["n_calls_writeSamFlags"] = n_first
                    r = side.result_for(sd)
                    r.f["nanoporeRead"].f["sam"] = sam
                    if j.call_virtual(sd, "barcodeFound", "()Z"):
                        opt = j.call_static(ONR, "getPostBCUMIseq", f"(L{ONR};L{UPAR};)Ljava/util/Optional;", r, par)
                        present = j.call_virtual(opt, "isPresent", "()Z")
                        case["u7"] = j.to_jstring(j.call_virtual(j.call_virtual(opt, "get", "()Ljava/lang/Object;"), "toString", "()Ljava/lang/String;")) if present else None
            except JavaThrow as e:
                case["throws"] = e.obj.cls
                case["message"] = e.obj.f.get("message")
            s["cases"].append(case)
        s["five_prime"] = five_prime
        out["sections"].append(g.finish(s))
        print(f"  samtags {'5p' if five_prime else '3p'}: {len(names)} names  {time.time() - g.t0:.0f}s", flush=True)
    return out


def gen_clusterpos(g, seed=2020):
    """a-18 input: the genome position a read is grouped by"""
    j = g.j
    rng = random.Random(seed)
    H = j.hooks
    H[SAMREC + ".<clinit>:()V"] = None
    H[SAMREC + ".getReadName:()Ljava/lang/String;"] = lambda jj, o: o.native["name"]
    H[SAMREC + ".getReadUnmappedFlag:()Z"] = lambda jj, o: 1 if o.native["flag"] & 4 else 0
    H[SAMREC + ".getAlignmentBlocks:()Ljava/util/List;"] = lambda jj, o: o.native["blocks"]
    DS = "com/rw/umifinder/scanstats/DebugStats"
    out = {"jar": "NanoporeBC_UMI_finder-2.1.jar + htsjdk-4.1.3.jar (TextCigarCodec, SAMUtils.getAlignmentBlocks, CoordMath)", "sections": []}
    for five_prime in (False, True):
        side = UmiSide(g, five_prime)
        par = side.par
        s = g.section(("5-prime (-p)" if five_prime else "3-prime") + " NanoporeRead$ReadScanData.generateReadScanData(sam, params) (NanoporeRead$ReadScanData.java:"
                      "L86-153): name parse, read position = polyA start - distanceFromReadEndForGrouping (3') or adapter end + cell_bc_length + "
                      "umi_length + that distance (5'), getReferencePositionAtReadPosition over the record's alignment blocks -> "
                      "positionOnGenomeForClustering (null = absent).  The SAM record is a stand-in holding name, flag and the alignment blocks "
                      "htsjdk's own SAMUtils.getAlignmentBlocks makes from the CIGAR text", RSD,
                      "generateReadScanData:(Lhtsjdk/samtools/SAMRecord;L...ParametersBarcodeUMiFinderAppParams;)Lcom/google/common/base/Optional;")
        for k in range(160):
            bc, umi = rnd_seq(rng, 16), rnd_seq(rng, 12)
            ae = rng.randrange(30, 600)
            nm = fake_name(rng, k, five_prime, bc, umi, rng.random() < 0.5, ae, rng.choice([0, 0, 1, -1]))
            if k % 23 == 5:
                nm = "noscan_read_%d" % k
            if k % 29 == 7 and not five_prime:
                nm = f"nopolya{k}_FWD_AE={ae}_X=ACGTACGTAC_Q=9_1"        # a 3' name without PS / PE (--noPolyARequired)
            ops = []
            if rng.random() < 0.6:
                ops.append((rng.choice("SH"), rng.randrange(1, 500)))
            for b in range(rng.randrange(1, 7)):
                ops.append((rng.choice("M=X"), rng.randrange(1, 400)))
                if b < 5 and rng.random() < 0.8:
                    ops.append((rng.choice("IDN"), rng.randrange(1, 60) if rng.random() < 0.7 else rng.randrange(100, 3000)))
            while ops and ops[-1][0] in "IDN":
                ops.pop()
            if rng.random() < 0.5:
                ops.append(("S", rng.randrange(1, 700)))
            flag = (16 if rng.random() < 0.5 else 0) | (4 if k % 17 == 3 else 0)
            pos0 = rng.randrange(0, 5_000_000)
            text = "".join(f"{ln}{op}" for op, ln in ops)
            sam = JObject(SAMREC)
            if flag & 4:
                lst = JObject("java/util/ArrayList")
                lst.native = []
                sam.native = {"name": nm, "flag": flag, "blocks": lst}
            else:
                cg = j.call_static("htsjdk/samtools/TextCigarCodec", "decode", "(Ljava/lang/String;)Lhtsjdk/samtools/Cigar;", text)
                sam.native = {"name": nm, "flag": flag, "blocks": j.call_static("htsjdk/samtools/SAMUtils", "getAlignmentBlocks",
                                                                                "(Lhtsjdk/samtools/Cigar;ILjava/lang/String;)Ljava/util/List;", cg, pos0 + 1, "read cigar")}
            case = {"name": nm, "flag": flag, "pos0": pos0, "cigar": text}
            try:
                opt = j.call_static(RSD, "generateReadScanData", f"(L{SAMREC};L{UPAR};)L{GOPT};", sam, par)
                sd = j.call_virtual(opt, "orNull", "()Ljava/lang/Object;")
                case["scan_data"] = sd is not None
                if sd is not None:
                    p = j.call_virtual(sd.f["positionOnGenomeForClustering"], "orNull", "()Ljava/lang/Object;")
                    case["position"] = None if p is None else p.v
            except JavaThrow as e:
                case["throws"] = e.obj.cls
            s["cases"].append(case)
        s["five_prime"] = five_prime
        s["distanceFromReadEndForGrouping"] = par.f["barcodes"].f["distanceFromReadEndForGrouping"].v
        out["sections"].append(g.finish(s))
    return out


def gen_bamorder(g, seed=2121):
    """the order of the records of one written batch"""
    import functools

    j = g.j
    rng = random.Random(seed)
    H = j.hooks
    H[SAMREC + ".<clinit>:()V"] = None
    box = lambda v: JBox("java/lang/Integer", v)  # noqa: E731
    hdr = JObject("htsjdk/samtools/SAMFileHeader")
    H[SAMREC + ".getHeader:()Lhtsjdk/samtools/SAMFileHeader;"] = lambda jj, o: hdr
    H[SAMREC + ".getReferenceIndex:()Ljava/lang/Integer;"] = lambda jj, o: box(o.native["ref"])
    H[SAMREC + ".getAlignmentStart:()I"] = lambda jj, o: o.native["pos0"] + 1
    H[SAMREC + ".getReadNegativeStrandFlag:()Z"] = lambda jj, o: 1 if o.native["flag"] & 16 else 0
    H[SAMREC + ".getReadName:()Ljava/lang/String;"] = lambda jj, o: o.native["name"]
    H[SAMREC + ".getFlags:()I"] = lambda jj, o: o.native["flag"]
    H[SAMREC + ".getMappingQuality:()I"] = lambda jj, o: o.native["mapq"]
    H[SAMREC + ".getMateReferenceIndex:()Ljava/lang/Integer;"] = lambda jj, o: box(o.native["mate_ref"])
    H[SAMREC + ".getMateAlignmentStart:()I"] = lambda jj, o: o.native["mate_pos0"] + 1
    H[SAMREC + ".getInferredInsertSize:()I"] = lambda jj, o: o.native["tlen"]
    CMP = "htsjdk/samtools/SAMRecordCoordinateComparator"
    out = {"jar": "htsjdk-4.1.3.jar", "sections": []}
    s = g.section("Arrays.stream(results).sorted(SAMRecordCoordinateComparator::compare) of UmiFinderWorker$BamWriters.writeSams (L421): a stable sort "
                  "of one batch with htsjdk's SAMRecordCoordinateComparator.compare (SAMRecordCoordinateComparator.java:L48-105) executed for every "
                  "comparison.  Records are stand-ins holding the nine values the comparator reads; drawn with many ties", CMP,
                  "compare:(Lhtsjdk/samtools/SAMRecord;Lhtsjdk/samtools/SAMRecord;)I")
    cmpo = j.new(CMP, "()V")
    for case in range(8):
        n = [6, 40, 40, 120, 120, 120, 300, 300][case]
        names = [f"r{k}_{rng.choice(['FWD', 'REV'])}_x" for k in range(max(3, n // 4))]
        recs = []
        for _ in range(n):
            unm = rng.random() < 0.1
            recs.append({"ref": -1 if unm else rng.randrange(0, 3), "pos0": -1 if unm and rng.random() < 0.7 else rng.randrange(0, 12), "flag": rng.choice([0, 16, 0, 16, 256, 272, 2048, 2064, 4, 20]),
                         "name": rng.choice(names), "mapq": rng.choice([0, 30, 60]), "mate_ref": rng.choice([-1, -1, 0, 2]), "mate_pos0": rng.choice([-1, 5, 900]),
                         "tlen": rng.choice([0, 0, -350, 350])})
        objs = []
        for r in recs:
            o = JObject(SAMREC)
            o.native = r
            objs.append(o)
        n_cmp = [0]

        def cmp(a, b):
            n_cmp[0] += 1
            return j.call_virtual(cmpo, "compare", f"(L{SAMREC};L{SAMREC};)I", objs[a], objs[b])

        order = sorted(range(n), key=functools.cmp_to_key(cmp))     # Python's sort is stable like Stream.sorted
        s["cases"].append({"records": recs, "order": order, "comparisons": n_cmp[0]})
    out["sections"].append(g.finish(s))
    return out


def install_bytebuffer(j):
    """java.nio.ByteBuffer over a byte[] (little endian, as BinaryTagCodec.readTags sets it): the relative reads htsjdk's tag reader uses"""
    import struct

    N = j.natives
    BB = "java/nio/ByteBuffer"

    def wrap(jj, arr, off, ln):
        o = JObject(BB)
        o.native = {"b": bytes((v & 0xFF) for v in arr.a), "pos": off, "lim": off + ln, "mark": None}
        return o

    def take(o, n):
        st = o.native
        if st["pos"] + n > st["lim"]:
            raise JavaThrow(JObject("java/nio/BufferUnderflowException"), [])
        v = st["b"][st["pos"]:st["pos"] + n]
        st["pos"] += n
        return v

    def get_bytes(jj, o, dst):
        v = take(o, len(dst.a))
        dst.a[:] = [x - 256 if x > 127 else x for x in v]
        return o

    N[BB + ".wrap:([BII)Ljava/nio/ByteBuffer;"] = wrap
    N[BB + ".order:(Ljava/nio/ByteOrder;)Ljava/nio/ByteBuffer;"] = lambda jj, o, order: o
    N["java/nio/ByteOrder.LITTLE_ENDIAN"] = lambda jj: JObject("java/nio/ByteOrder")
    N[BB + ".hasRemaining"] = lambda jj, o: 1 if o.native["pos"] < o.native["lim"] else 0
    N[BB + ".get:()B"] = lambda jj, o: struct.unpack("<b", take(o, 1))[0]
    N[BB + ".get:([B)Ljava/nio/ByteBuffer;"] = get_bytes
    N[BB + ".getShort:()S"] = lambda jj, o: struct.unpack("<h", take(o, 2))[0]
    N[BB + ".getInt:()I"] = lambda jj, o: struct.unpack("<i", take(o, 4))[0]
    N[BB + ".getFloat:()F"] = lambda jj, o: f32(struct.unpack("<f", take(o, 4))[0])
    N[BB + ".mark"] = lambda jj, o: (o.native.__setitem__("mark", o.native["pos"]), o)[1]
    N[BB + ".reset"] = lambda jj, o: (o.native.__setitem__("pos", o.native["mark"]), o)[1]
    N[BB + ".position:()I"] = lambda jj, o: o.native["pos"]


def gen_auxorder(g, seed=2222):
    """the attribute list of a record that is read from a BAM, tagged and written again"""
    import struct

    j = g.j
    rng = random.Random(seed)
    install_bytebuffer(j)
    H = j.hooks
    H[SAMREC + ".<clinit>:()V"] = None
    BTC, TAV, STAG = "htsjdk/samtools/BinaryTagCodec", "htsjdk/samtools/SAMBinaryTagAndValue", "htsjdk/samtools/SAMTag"
    H["htsjdk/samtools/util/Log.<clinit>:()V"] = None
    H["htsjdk/samtools/util/Log.*"] = lambda jj, *a: None
    H["htsjdk/samtools/util/Log.getInstance:(Ljava/lang/Class;)Lhtsjdk/samtools/util/Log;"] = lambda jj, *a: JObject("htsjdk/samtools/util/Log")
    out = {"jar": "htsjdk-4.1.3.jar", "sections": []}
    s = g.section("BinaryTagCodec.readTags(aux bytes, 0, n, SILENT) (BinaryTagCodec.java:L271-305: the list is built through SAMBinaryTagAndValue.insert, "
                  "i.e. ordered by binary tag while it is read, a repeated tag replacing the earlier one), then SAMRecord.setAttribute(tag, value) "
                  "per call (SAMRecord.java:L1531-1602; value null = remove), then the list as BAMRecordCodec would write it: [tag, value, the type "
                  "character BinaryTagCodec.getTagValueType picks].  The record is an otherwise empty SAMRecord object", BTC,
                  "readTags:([BIILhtsjdk/samtools/ValidationStringency;)Lhtsjdk/samtools/SAMBinaryTagAndValue;")
    silent = j.get_static("htsjdk/samtools/ValidationStringency", "SILENT")
    pool = ["NM", "ms", "AS", "nn", "tp", "cm", "s1", "s2", "de", "rl", "SA", "MD", "GE", "GS", "XF", "BC", "U8", "zz", "AA", "B1", "ts"]
    new_tags = ["PE", "PS", "AE", "RE", "TE", "BU", "BV", "BE", "BW", "BX", "SX", "BH", "BC", "BZ", "BB", "BF", "B1", "B2", "BZ", "BH", "XF", "GE", "GS",
                "U8", "U7", "UC", "U1", "U2", "UZ"]
    for case in range(60):
        aux, given = b"", []
        for t in rng.sample(pool, rng.randrange(0, 9)) + ([rng.choice(pool)] if case % 7 == 3 else []):
            kind = rng.choice("cCsSiIZAf")
            if kind in "cCsSiI":
                lo, hi = {"c": (-128, 127), "C": (0, 255), "s": (-32768, 32767), "S": (0, 65535), "i": (-2 ** 31, 2 ** 31 - 1), "I": (0, 2 ** 32 - 1)}[kind]
                v = rng.choice([lo, hi, 0, 1, 3, 100, 200, 300, 40000, 70000, rng.randrange(lo, hi + 1)])
                v = min(max(v, lo), hi)
                aux += t.encode() + kind.encode() + struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I"}[kind], v)
            elif kind == "Z":
                v = rng.choice(["", "chr1,100,+,50M,60,0;", "GENE1", "CODING"])
                aux += t.encode() + b"Z" + v.encode() + b"\0"
            elif kind == "A":
                v = rng.choice("PSI")
                aux += t.encode() + b"A" + v.encode()
            else:
                v = rng.choice([0.0, 0.5, 0.0123])
                aux += t.encode() + b"f" + struct.pack("<f", v)
            given.append([t, kind, v])
        calls = []
        for t in new_tags:
            if rng.random() < 0.55:
                calls.append([t, rng.choice([None, "", "ACGT", "17"]) if t in ("GE", "GS") else rng.choice(["", "ACGTACGTACGTACGT", "12", 0, 1, 130, 611, 70000, -3])])
        rec = j.new_object(SAMREC)
        arr = j.byte_array([b - 256 if b > 127 else b for b in aux])
        head = j.call_static(BTC, "readTags", f"([BIILhtsjdk/samtools/ValidationStringency;)L{TAV};", arr, 0, len(aux), silent) if aux else None
        rec.f["mAttributes"] = head
        for t, v in calls:
            j.call_virtual(rec, "setAttribute", "(Ljava/lang/String;Ljava/lang/Object;)V", t, JBox("java/lang/Integer", v) if isinstance(v, int) else v)
        final = []
        node = rec.f["mAttributes"]
        while node is not None:
            tag = j.to_jstring(j.call_static(STAG, "makeStringTag", "(S)Ljava/lang/String;", node.f["tag"]))
            val = node.f["value"]
            ty = chr(j.call_static(BTC, "getTagValueType", "(Ljava/lang/Object;)C", val))
            final.append([tag, val.v if isinstance(val, JBox) else val, ty])
            node = node.f["next"]
        s["cases"].append({"aux_hex": aux.hex(), "aux": given, "calls": calls, "final": final})
    out["sections"].append(g.finish(s))
    return out


SECTIONS = {"pass2t": gen_pass2t, "pass2k": gen_pass2k, "umi_3p_len10": gen_umi_3p_len10, "umi_5p_len10": gen_umi_5p_len10, "auxorder": gen_auxorder, "bamorder": gen_bamorder, "clusterpos": gen_clusterpos, "samtags": gen_samtags, "genecounts": gen_genecounts, "cluster_own": gen_cluster_own, "pass1": gen_pass1, "cluster": gen_cluster, "group": gen_group, "finalize": gen_finalize, "gene": gen_gene, "gene_gtf": gen_gene_gtf, "twobit": gen_twobit, "onebyte": gen_onebyte, "nw": gen_nw, "lev": gen_lev, "bcmatch": gen_bcmatch, "polyat": gen_polyat, "polyat_params": gen_polyat_params,
            "pass2_3p": gen_pass2_3p, "pass2_3p_ed2": gen_pass2_3p_ed2, "pass2_5p": gen_pass2_5p, "pass2_5p_polya": gen_pass2_5p_polya,
            "umi_3p": gen_umi_3p, "umi_5p": gen_umi_5p, "chimera_3p": gen_chimera_3p, "stats_print": gen_stats_print,
            "pass2w_3p": gen_pass2w_3p, "pass2w_3p_ed2": gen_pass2w_3p_ed2, "pass2w_5p": gen_pass2w_5p, "pass2w_5p_polya": gen_pass2w_5p_polya,
            "pass2x_3p": gen_pass2x_3p, "pass2x_5p": gen_pass2x_5p, "pass2p": gen_pass2p, "group2": gen_group2, "cluster_own2": gen_cluster_own2,
            "pass1_5p": lambda g: gen_pass1(g, 32, 1626, five_prime=True),
            "pass1_nowl": lambda g: gen_pass1(g, 60, 1636, no_whitelist=True), "pass1_nowl_5p": lambda g: gen_pass1(g, 66, 1646, five_prime=True, no_whitelist=True)}


def run_section(name):
    g = Gen()
    t0 = time.time()
    data = SECTIONS[name](g)
    data["generated_by"] = "tools/make_ref_exec.py " + name
    data["how"] = ("outputs computed by executing the reference's class files (jar named in `jar`) with tools/jvm_exec.py; "
                   "inputs are seeded random / hand-picked; nothing here comes from oracle/ or from the HIP library")
    data["bytecode_steps"] = data.pop("_steps", 0) + g.j.steps
    hits = merge_hits(data.pop("_hits", {}), g.hits())
    path = os.path.join(OUT, f"ref_exec_{name}.json")
    with open(path, "w") as f:
        json.dump(data, f, separators=(",", ":"))
    os.makedirs(COV_DIR, exist_ok=True)
    with open(os.path.join(COV_DIR, f"{name}.json"), "w") as f:
        json.dump({"section": name, "hits": hits}, f, separators=(",", ":"), sort_keys=True)
    n_cases = sum(len(s["cases"]) for s in data["sections"])
    print(f"{name}: {n_cases} cases, {data['bytecode_steps']} bytecode steps, {time.time() - t0:.1f}s, {os.path.getsize(path) / 1024:.0f} KiB, "
          f"max tier {max(s['max_tier'] for s in data['sections'])}", flush=True)
    return name


def _ranges(lines):
    out, lines = [], sorted(lines)
    for ln in lines:
        if out and ln == out[-1][1] + 1:
            out[-1][1] = ln
        else:
            out.append([ln, ln])
    return [f"L{a}" if a == b else f"L{a}-{b}" for a, b in out]


def merge_coverage():
    """tests/golden/coverage/*.json + every method's LineNumberTable -> tests/golden/ref_exec_coverage.json"""
    import bisect
    import glob

    hits, sections = {}, []
    for path in sorted(glob.glob(os.path.join(COV_DIR, "*.json"))):
        with open(path) as f:
            d = json.load(f)
        sections.append(d["section"])
        merge_hits(hits, d["hits"])
    notes_path = os.path.join(HERE, "coverage_notes.json")
    notes_file = json.load(open(notes_path)) if os.path.exists(notes_path) else {}
    notes, scope = notes_file.get("notes", {}), notes_file.get("scope", {})

    def in_scope(simple, ln):
        for r in scope.get(simple, []):
            a, _, b = r[1:].partition("-")
            if int(a) <= ln <= int(b or a):
                return True
        return False
    j = JVM(JARS)
    rep, tot = {}, {"lines": 0, "hit": 0, "annotated": 0, "instr": 0, "instr_hit": 0}
    stot = {"lines": 0, "hit": 0, "annotated": 0}
    for cname in COVERAGE_CLASSES:
        if not j.has_class(cname):
            continue
        jc = j.load(cname)
        cnotes = notes.get(cname.split("/")[-1], {})

        def reason(ln, mname):
            for key, why in cnotes.items():
                if key == mname:
                    return why
                if key.startswith("L"):
                    a, _, b = key[1:].partition("-")
                    if int(a) <= ln <= int(b or a):
                        return why
            return None

        cm, csum = {}, {"lines": 0, "hit": 0, "annotated": 0}
        cl_present, cl_hit, line_methods = set(), set(), {}
        for (name, desc), m in jc.methods.items():
            if m.code is None:
                continue
            if m.ops is None:
                j._decode(m)
            if not m.lines:
                continue
            hx = hits.get(cname, {}).get(f"{name}:{desc}")
            hit = bytes.fromhex(hx) if hx else bytes(len(m.hit))
            starts = [pc for pc, _ in m.lines]
            present, got, n_hit = set(), set(), 0
            for pc in m.ops:
                ln = m.lines[max(bisect.bisect_right(starts, pc) - 1, 0)][1]
                present.add(ln)
                if hit[pc]:
                    got.add(ln)
                    n_hit += 1
            cl_present |= present
            cl_hit |= got
            for ln in present:
                line_methods.setdefault(ln, set()).add(name)
            tot["instr"] += len(m.ops)
            tot["instr_hit"] += n_hit
            missed = sorted(present - got)
            e = {"lines": _ranges(present), "n_lines": len(present), "n_hit": len(got), "instr": len(m.ops), "instr_hit": n_hit}
            if missed:
                e["never_reached"] = _ranges(missed)
                why = {}
                for ln in missed:
                    r = reason(ln, name)
                    if r:
                        why.setdefault(r, []).append(ln)
                if why:
                    e["reasons"] = {r: _ranges(v) for r, v in why.items()}
                e["unexplained"] = _ranges([ln for ln in missed if not reason(ln, name)])
            cm[f"{name}:{desc}"] = e
        # a source line can belong to several methods (lambdas): count it once per class
        missed = cl_present - cl_hit
        ann = {ln for ln in missed if any(reason(ln, mn) for mn in line_methods[ln])}
        csum = {"lines": len(cl_present), "hit": len(cl_hit), "annotated": len(ann), "unexplained": _ranges(missed - ann),
                "pct_hit": round(100.0 * len(cl_hit) / max(1, len(cl_present)), 1),
                "pct_hit_or_annotated": round(100.0 * (len(cl_hit) + len(ann)) / max(1, len(cl_present)), 1)}
        for k in ("lines", "hit", "annotated"):
            tot[k] += csum[k]
        simple = cname.split("/")[-1]
        if simple in scope:     # the line ranges SURVEY 8a cites for this class
            sp = {ln for ln in cl_present if in_scope(simple, ln)}
            sh, sa = sp & cl_hit, sp & ann
            csum["scope"] = {"ranges": scope[simple], "lines": len(sp), "hit": len(sh), "annotated": len(sa), "unexplained": _ranges(sp - sh - sa),
                             "pct_hit": round(100.0 * len(sh) / max(1, len(sp)), 1), "pct_hit_or_annotated": round(100.0 * (len(sh) + len(sa)) / max(1, len(sp)), 1)}
            stot["lines"] += len(sp)
            stot["hit"] += len(sh)
            stot["annotated"] += len(sa)
        rep[cname] = {"summary": csum, "methods": cm}
    stot["pct_hit"] = round(100.0 * stot["hit"] / max(1, stot["lines"]), 1)
    stot["pct_hit_or_annotated"] = round(100.0 * (stot["hit"] + stot["annotated"]) / max(1, stot["lines"]), 1)
    tot["pct_hit"] = round(100.0 * tot["hit"] / max(1, tot["lines"]), 1)
    tot["pct_hit_or_annotated"] = round(100.0 * (tot["hit"] + tot["annotated"]) / max(1, tot["lines"]), 1)
    out = {"generated_by": "tools/make_ref_exec.py --coverage", "sections_merged": sections,
           "how": "an instruction counts as executed if the interpreter ran it in any section; a source line (LineNumberTable) counts as executed "
                  "if one of its instructions did; `annotated` = never reached, with a reason in tools/coverage_notes.json",
           "scope_note": "`scope` = the source-line ranges of the hot-path methods as SURVEY 8a cites them (tools/coverage_notes.json); the class totals also "
                         "count constructors, accessors, toString and the methods of other callers",
           "scope_total": stot, "total": tot, "classes": rep}
    with open(os.path.join(OUT, "ref_exec_coverage.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=False)
    print(f"coverage: {tot['hit']}/{tot['lines']} lines executed ({tot['pct_hit']} %), + {tot['annotated']} annotated = {tot['pct_hit_or_annotated']} %; "
          f"{tot['instr_hit']}/{tot['instr']} instructions")
    print(f"in scope (SURVEY 8a ranges): {stot['hit']}/{stot['lines']} lines executed ({stot['pct_hit']} %), + {stot['annotated']} annotated = {stot['pct_hit_or_annotated']} %")
    for cname, e in rep.items():
        sc = e["summary"].get("scope")
        if sc:
            print(f"  [scope] {cname.split('/')[-1]:40s} {sc['hit']:4d}/{sc['lines']:4d} {sc['pct_hit']:5.1f} % (+{sc['annotated']}) unexplained: {' '.join(sc['unexplained'])}")
    for cname, e in rep.items():
        sm = e["summary"]
        print(f"  {cname.split('/')[-1]:45s} {sm['hit']:4d}/{sm['lines']:4d} {sm['pct_hit']:5.1f} %  (+{sm['annotated']} annotated)  unexplained: {' '.join(sm['unexplained'][:12])}")


def main():
    args = sys.argv[1:]
    jobs = 1
    if "--jobs" in args:
        k = args.index("--jobs")
        jobs = int(args[k + 1])
        del args[k:k + 2]
    if args == ["--coverage"]:
        return merge_coverage()
    want = args or list(SECTIONS)
    unknown = [w for w in want if w not in SECTIONS]
    if unknown:
        sys.exit(f"unknown section(s): {unknown}; known: {sorted(SECTIONS)}")
    os.makedirs(OUT, exist_ok=True)
    if jobs <= 1 or len(want) == 1:
        if jobs > 1:
            os.environ["WIDE_JOBS"] = str(jobs)   # one section: the chunks of a pass2w_* section are spread over the processes instead
        for name in want:
            run_section(name)
        return
    import multiprocessing as mp

    # longest sections first so that the pool drains evenly
    weight = {"cluster_own": 100, "pass2_3p_ed2": 30, "chimera_3p": 25, "finalize": 10, "cluster": 10, "group": 8}
    want = sorted(want, key=lambda n: -weight.get(n, 1))
    with mp.get_context("fork").Pool(jobs, maxtasksperchild=1) as pool:
        for name in pool.imap_unordered(run_section, want):
            pass


if __name__ == "__main__":
    main()
