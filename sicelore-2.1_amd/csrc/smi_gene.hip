// smi_gene.hip -- the GE / GS / XF tags of `assignumis` (host; no device code): the reference's default gene tagger
//   GennameTagger.setGeneExons and helpers        FJ!umifinder/bamreaders/GennameTagger.java:L73-366   (config.xml:86-89 DefaultTagger)
//   TagReadBase.<init>                            DropseqLib-1.0.jar!/org/broadinstitute/dropseqrna/metrics/TagReadBase.java:L64-71
//                                                 (ALLOW_MULTI_GENE_READS = true as GennameTagger constructs it, L65)
//   RefFlatReader.load / makeGeneFromRefFlatLines  picard-2.23.9.jar!/picard/annotation/RefFlatReader.java:L70-190
//   Gene / Gene$Transcript                        picard/annotation/Gene.java:L46-71, L102-205 (assignLocusFunctionForRange, inExon, utr)
//   OverlapDetector.addLhs / getOverlaps          htsjdk-4.1.3.jar!/htsjdk/samtools/util/OverlapDetector.java:L69-89, L179-200
//   Interval.hashCode / compareTo / intersects    htsjdk/samtools/util/Interval.java:L124-125, L184-195, L228-231
//   SAMUtils.getAlignmentBlocks                   (M, =, X make a block; I, S advance the read; D, N the reference)
// All of these are bytecode under /root/reference/Jar (read with tools/classfold.py).  Call site: OneNanoporeSeqAnalyzer.call L95-103
// (after ReadScanResult.writeSamFlags, before the UMI tags).
//
// Several java.util.HashMap / HashSet objects sit on the way (genes by name, the overlap set, the per-gene function map, the exon-consistent
// set), so the ORDER of the names in a multi-gene GE value ("A,B") and which of two genes with the same interval and strand survives
// (Gene.equals compares interval + strand only) follow the JDK's hash iteration order.  That order is reproduced here from the objects' own
// hashCode() (String.hashCode; Interval.hashCode = 31 * (31 * contig.hashCode() + start) + end): buckets ascending at the table's final
// capacity, insertion order inside a bucket (bins long enough to be treeified -- 8 colliding keys -- are not expected in annotation data).
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "smi_internal.h"

using namespace smi;

namespace {

inline int32_t jstring_hash(const std::string &s) {
    uint32_t h = 0;
    for (unsigned char c : s) h = 31u * h + c;
    return (int32_t)h;
}
inline uint32_t spread(int32_t h) {
    const uint32_t u = (uint32_t)h;
    return u ^ (u >> 16);
}
// iteration order of a java.util.HashMap / HashSet created with the default constructor that received `hashes` in this insertion order
// (no removals): table of 16 doubling when the size passes 3/4 of it, bins in insertion order, bins visited in ascending index.  The model
// ends where the JDK leaves it: a bin that reaches 8 entries is turned into a red-black tree (or, under 64 slots, forces a resize) and
// its iteration order then depends on the tree's shape (ties by System.identityHashCode) -- such a map is REFUSED here (g_jhash_unmodelled,
// checked by the callers), as tools/jvm_natives.py refuses it when the fixtures are made.  Whether the JDK's order is modelled correctly
// below that point rests on the published HashMap source, not on a JVM run (DESIGN.md "Oracle": parity of hash-ordered outputs unpinned).
thread_local bool g_jhash_unmodelled = false;
std::vector<size_t> jhash_order(const std::vector<int32_t> &hashes) {
    size_t cap = 16;
    while (hashes.size() > (cap * 3) / 4) cap <<= 1;
    std::vector<size_t> idx(hashes.size());
    for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return (spread(hashes[a]) & (cap - 1)) < (spread(hashes[b]) & (cap - 1)); });
    size_t run = 0;
    for (size_t i = 0; i < idx.size(); i++) {
        run = (i && (spread(hashes[idx[i]]) & (cap - 1)) == (spread(hashes[idx[i - 1]]) & (cap - 1))) ? run + 1 : 1;
        if (run >= 8) g_jhash_unmodelled = true;  // TREEIFY_THRESHOLD
    }
    return idx;
}

enum LF { INTERGENIC = 0, INTRONIC = 1, UTR = 2, CODING = 3, RIBOSOMAL = 4 };  // LocusFunction ordinals
const char *const LF_NAME[5] = {"INTERGENIC", "INTRONIC", "UTR", "CODING", "RIBOSOMAL"};
// GennameTagger.LOCUS_FUNCTION_SCORES (L39-43); RIBOSOMAL has no score (never produced by assignLocusFunctionForRange)
inline int lf_score(int f) { return f == CODING ? 4 : f == UTR ? 3 : f == INTRONIC ? 2 : 1; }

struct Transcript {
    std::string name;
    int tx_start, tx_end, cds_start, cds_end;
    std::vector<std::pair<int, int>> exons;  // 1-based inclusive, file order
    bool in_exon(int locus) const {          // Gene$Transcript.inExon L200-205: stops at the first exon that starts behind the locus
        for (const auto &e : exons) {
            if (e.first > locus) return false;
            if (locus >= e.first && locus <= e.second) return true;
        }
        return false;
    }
    bool utr(int locus) const { return locus < cds_start || locus > cds_end; }  // L196
};

struct Gene {
    int contig;  // index into the reference names
    int start, end;
    bool negative;
    std::string name;
    std::vector<Transcript> tx;  // in the iteration order of Gene.transcripts (a HashMap<String, Transcript>)
    int32_t hash;                // Interval.hashCode
};

struct Block {
    int ref_start, length;
};

}  // namespace

struct smi_genes {
    std::vector<std::string> refs;
    std::vector<int32_t> ref_hash;
    std::vector<Gene> genes;                 // in the order the reference adds them to the OverlapDetector
    std::vector<std::vector<int>> by_contig; // gene indices per contig, sorted by (start, end) -- the IntervalTree's in-order walk
    size_t n_lines = 0, n_skipped_sequence = 0, n_skipped_genes = 0;
};

static bool same_gene(const Gene &a, const Gene &b) {  // Gene.equals = compareTo == 0: contig, start, end, strand (Gene.java:L61-71)
    return a.contig == b.contig && a.start == b.start && a.end == b.end && a.negative == b.negative;
}

// the loaders' common end: the genes of a contig in the IntervalTree's in-order walk; a map the JDK would have treeified is refused
static int finish_genes(smi_genes *G, int n_refs, const char *who, smi_genes **out) {
    G->by_contig.assign((size_t)n_refs, {});
    for (size_t i = 0; i < G->genes.size(); i++) G->by_contig[(size_t)G->genes[i].contig].push_back((int)i);
    for (auto &v : G->by_contig)
        std::stable_sort(v.begin(), v.end(), [&](int a, int b) {
            const Gene &x = G->genes[(size_t)a], &y = G->genes[(size_t)b];
            return x.start != y.start ? x.start < y.start : x.end < y.end;
        });
    if (g_jhash_unmodelled) {
        g_jhash_unmodelled = false;
        delete G;
        set_error(std::string(who) + ": a java.util.HashMap bin of the reference would hold 8 or more entries here (gene names / transcript names with "
                  "colliding hashes): its iteration order is a tree's, which this library does not model");
        return SMI_ERR_INVALID;
    }
    *out = G;
    return SMI_OK;
}

extern "C" int smi_genes_load_refflat(const char *text, size_t n_bytes, const char *const *ref_names, int n_refs, smi_genes **out) {
    if (!out || (!text && n_bytes) || (n_refs && !ref_names) || n_refs < 0) {
        set_error("smi_genes_load_refflat: null argument");
        return SMI_ERR_INVALID;
    }
    *out = nullptr;
    smi_genes *G = new smi_genes();
    std::unordered_map<std::string, int> ref_index;
    for (int i = 0; i < n_refs; i++) {
        G->refs.emplace_back(ref_names[i]);
        G->ref_hash.push_back(jstring_hash(G->refs.back()));
        ref_index.emplace(G->refs.back(), i);
    }
    // rows by gene name: HashMap<String, List<Row>> (RefFlatReader.java:L74-95)
    struct Row {
        std::vector<std::string> f;
    };
    std::vector<std::string> names;                // insertion order of the map's keys
    std::unordered_map<std::string, size_t> slot;  // name -> index into names / rows
    std::vector<std::vector<Row>> rows;
    size_t p = 0;
    size_t line_no = 0;
    while (p < n_bytes) {
        size_t e = p;
        while (e < n_bytes && text[e] != '\n') e++;
        size_t le = e;
        if (le > p && text[le - 1] == '\r') le--;
        std::string line(text + p, le - p);
        p = e + 1;
        line_no++;
        if (line.empty() || line[0] == '#') continue;  // BasicInputParser: blank lines and comment lines are skipped
        Row r;
        size_t a = 0;
        for (;;) {
            size_t t = line.find('\t', a);
            if (t == std::string::npos) {
                r.f.push_back(line.substr(a));
                break;
            }
            r.f.push_back(line.substr(a, t - a));
            a = t + 1;
        }
        if (r.f.size() != 11) {
            delete G;
            set_error("smi_genes_load_refflat: wrong number of fields in the refFlat text at line " + std::to_string(line_no) +
                      " (the reference throws AnnotationException here, RefFlatReader.java:L79-80)");
            return SMI_ERR_INVALID;
        }
        G->n_lines++;
        if (!ref_index.count(r.f[2])) {  // isSequenceRecognized L87-88
            G->n_skipped_sequence++;
            continue;
        }
        auto it = slot.find(r.f[0]);
        if (it == slot.end()) {
            slot.emplace(r.f[0], names.size());
            names.push_back(r.f[0]);
            rows.emplace_back();
            rows.back().push_back(std::move(r));
        } else
            rows[it->second].push_back(std::move(r));
    }
    std::vector<int32_t> hs(names.size());
    for (size_t i = 0; i < names.size(); i++) hs[i] = jstring_hash(names[i]);
    auto to_int = [](const std::string &s, int *v) {
        if (s.empty()) return false;
        char *endp = nullptr;
        long x = std::strtol(s.c_str(), &endp, 10);
        if (*endp) return false;
        *v = (int)x;
        return true;
    };
    auto split_ints = [&](const std::string &s, std::vector<int> &v) {  // String.split(","): trailing empty strings dropped
        v.clear();
        size_t a = 0;
        std::vector<std::string> parts;
        for (;;) {
            size_t t = s.find(',', a);
            if (t == std::string::npos) {
                parts.push_back(s.substr(a));
                break;
            }
            parts.push_back(s.substr(a, t - a));
            a = t + 1;
        }
        if (parts.size() > 1)  // "".split(",") is [""], but "1,2,".split(",") drops the trailing empty strings
            while (!parts.empty() && parts.back().empty()) parts.pop_back();
        for (const auto &x : parts) {
            int k;
            if (!to_int(x, &k)) return false;
            v.push_back(k);
        }
        return true;
    };
    for (size_t gi : jhash_order(hs)) {  // refFlatLinesByGene.values() L102
        const std::vector<Row> &R = rows[gi];
        Gene g;
        g.name = R[0].f[0];
        const std::string strand = R[0].f[3], chrom = R[0].f[2];
        g.negative = strand == "-";
        g.contig = ref_index[chrom];
        int st = 2147483647, en = -2147483647 - 1;
        bool ok = true;
        for (const Row &r : R) {
            int a, b;
            if (!to_int(r.f[4], &a) || !to_int(r.f[5], &b)) {
                ok = false;
                break;
            }
            st = std::min(st, a + 1);
            en = std::max(en, b);
        }
        if (!ok) {  // NumberFormatException: not an AnnotationException, the reference dies; refuse the file
            delete G;
            set_error("smi_genes_load_refflat: non-numeric coordinate for gene " + g.name);
            return SMI_ERR_INVALID;
        }
        g.start = st;
        g.end = en;
        std::vector<std::string> tnames;
        std::vector<Transcript> txs;
        for (const Row &r : R) {
            if (r.f[3] != strand || r.f[2] != chrom) {  // L138-142: AnnotationException -> gene skipped (L108-109)
                ok = false;
                break;
            }
            if (std::find(tnames.begin(), tnames.end(), r.f[1]) != tnames.end()) {  // Gene.addTranscript L46-47
                ok = false;
                break;
            }
            Transcript t;
            t.name = r.f[1];
            int cnt, cs, ce, a, b;
            std::vector<int> es, ee;
            if (!to_int(r.f[8], &cnt) || !to_int(r.f[4], &a) || !to_int(r.f[5], &b) || !to_int(r.f[6], &cs) || !to_int(r.f[7], &ce) ||
                !split_ints(r.f[9], es) || !split_ints(r.f[10], ee)) {
                delete G;
                set_error("smi_genes_load_refflat: non-numeric field for transcript " + r.f[1]);
                return SMI_ERR_INVALID;
            }
            if (cnt != (int)es.size() || cnt != (int)ee.size()) {  // L164-168: AnnotationException
                ok = false;
                break;
            }
            t.tx_start = a + 1;
            t.tx_end = b;
            t.cds_start = cs + 1;
            t.cds_end = ce;
            for (int i = 0; i < cnt; i++) {
                // L181-186: "Exon has 0 or negative extent", "Exons overlap" -- AnnotationException, the gene is skipped
                if (es[i] + 1 > ee[i] || (i > 0 && t.exons.back().second >= es[i] + 1)) {
                    ok = false;
                    break;
                }
                t.exons.emplace_back(es[i] + 1, ee[i]);
            }
            if (!ok) break;
            tnames.push_back(t.name);
            txs.push_back(std::move(t));
        }
        if (!ok) {
            G->n_skipped_genes++;
            continue;
        }
        // Gene.iterator() = transcripts.values() of a HashMap<String, Transcript> (Gene.java:L57)
        std::vector<int32_t> th(tnames.size());
        for (size_t i = 0; i < tnames.size(); i++) th[i] = jstring_hash(tnames[i]);
        for (size_t k : jhash_order(th)) g.tx.push_back(std::move(txs[k]));
        g.hash = (int32_t)(31u * (31u * (uint32_t)G->ref_hash[(size_t)g.contig] + (uint32_t)g.start) + (uint32_t)g.end);
        // OverlapDetector.addLhs L69-89: interval tree node per (start, end); a node's value is a set, and Gene.equals makes a second gene with
        // the same interval and strand disappear in it (the first one added stays)
        bool dup = false;
        for (const Gene &o : G->genes)
            if (same_gene(o, g)) {
                dup = true;
                break;
            }
        if (dup) {
            G->n_skipped_genes++;
            continue;
        }
        G->genes.push_back(std::move(g));
    }
    return finish_genes(G, n_refs, "smi_genes_load_refflat", out);
}

// ---- the same model from a GTF (README.md:727 "path to refflat or GTF file"; GeneAnnotationReader.loadAnnotationsFile L46-55 picks by the file name) ----
//   GTFReader.load / $FilteringGTFParser        DropseqLib-1.0.jar!/org/broadinstitute/dropseqrna/annotation/GTFReader.java:L78-134
//   GTFParser.next / parseLine                  .../GTFParser.java:L83-137   (STRICT: an invalid line is an AnnotationException nobody catches)
//   GTFRecord.validate                          .../GTFRecord.java:L146-158
//   AnnotationUtils.parseOptionalFields         .../AnnotationUtils.java:L376-389
//   GeneFromGTFBuilder                          .../GeneFromGTFBuilder.java:L47-222 (genes by gene_name, highest gene_version, transcripts by transcript_id)
//   GeneFromGTF.addTranscript / equals / hashCode  .../GeneFromGTF.java:L81-113
// What differs from the refFlat model downstream: a gene's extent is the extent of ALL its records (not of its transcripts), transcripts are kept by
// transcript_name in the gene's own HashMap, and GeneFromGTF.hashCode = 31 * Interval.hashCode + name.hashCode with equals over start, end, name,
// contig and gene_id -- so two genes of one interval and strand both stay, and the order of a multi-gene value follows the other hash.
namespace {

bool java_parse_int(const std::string &s, int *v) {  // Integer.parseInt: optional sign, decimal digits, int range
    size_t i = 0;
    bool neg = false;
    if (!s.empty() && (s[0] == '-' || s[0] == '+')) {
        neg = s[0] == '-';
        i = 1;
    }
    if (i >= s.size()) return false;
    long long x = 0;
    for (; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        x = x * 10 + (s[i] - '0');
        if (x > 2147483648ll) return false;
    }
    if (neg) x = -x;
    if (x > 2147483647ll || x < -2147483648ll) return false;
    *v = (int)x;
    return true;
}

// String.split(one literal character): the pieces, trailing empty strings dropped (a string without the character is one piece, even if empty)
std::vector<std::string> java_split(const std::string &s, char c) {
    std::vector<std::string> parts;
    size_t a = 0;
    for (;;) {
        const size_t t = s.find(c, a);
        if (t == std::string::npos) {
            parts.push_back(s.substr(a));
            break;
        }
        parts.push_back(s.substr(a, t - a));
        a = t + 1;
    }
    if (parts.size() > 1)
        while (!parts.empty() && parts.back().empty()) parts.pop_back();
    return parts;
}

struct GtfRecord {
    std::string chrom, feature, gene_id, gene_name, tx_name, tx_id;
    bool has_gene_id = false, has_gene_name = false, has_tx_name = false, has_tx_id = false, has_version = false;
    int start = 0, end = 0, version = 0;
    bool negative = false;
};

}  // namespace

extern "C" int smi_genes_load_gtf(const char *text, size_t n_bytes, const char *const *ref_names, int n_refs, smi_genes **out) {
    if (!out || (!text && n_bytes) || (n_refs && !ref_names) || n_refs < 0) {
        set_error("smi_genes_load_gtf: null argument");
        return SMI_ERR_INVALID;
    }
    *out = nullptr;
    smi_genes *G = new smi_genes();
    std::unordered_map<std::string, int> ref_index;
    for (int i = 0; i < n_refs; i++) {
        G->refs.emplace_back(ref_names[i]);
        G->ref_hash.push_back(jstring_hash(G->refs.back()));
        ref_index.emplace(G->refs.back(), i);
    }
    auto fail = [&](const std::string &why) {
        delete G;
        set_error("smi_genes_load_gtf: " + why);
        return SMI_ERR_INVALID;
    };
    // ---- the records, gathered by gene name in file order (gatherByGeneName L208: a HashMap<String, List<GTFRecord>>)
    std::vector<std::string> names;
    std::unordered_map<std::string, size_t> slot;
    std::vector<std::vector<GtfRecord>> rows;
    size_t p = 0, line_no = 0;
    while (p < n_bytes) {
        size_t e = p;
        while (e < n_bytes && text[e] != '\n') e++;
        size_t le = e;
        if (le > p && text[le - 1] == '\r') le--;
        const std::string line(text + p, le - p);
        p = e + 1;
        line_no++;
        if (line.empty() || line[0] == '#') continue;  // BasicInputParser: blank lines and comment lines are skipped
        std::vector<std::string> f;
        {
            size_t a = 0;
            for (;;) {
                const size_t t = line.find('\t', a);
                if (t == std::string::npos) {
                    f.push_back(line.substr(a));
                    break;
                }
                f.push_back(line.substr(a, t - a));
                a = t + 1;
            }
        }
        const std::string at = " at line " + std::to_string(line_no);
        if (f.size() != 9) return fail("wrong number of fields in the GTF text" + at + " (AnnotationException, GTFParser.java:L84-86)");
        G->n_lines++;
        GtfRecord r;
        // AnnotationUtils.parseOptionalFields: split(";"), quotes removed, trimmed, empty pieces skipped, split(" "): key = piece 0, value = piece 1
        for (std::string piece : java_split(f[8], ';')) {
            piece.erase(std::remove(piece.begin(), piece.end(), '"'), piece.end());
            size_t a = 0, b = piece.size();
            while (a < b && (unsigned char)piece[a] <= ' ') a++;  // String.trim
            while (b > a && (unsigned char)piece[b - 1] <= ' ') b--;
            piece = piece.substr(a, b - a);
            if (piece.empty()) continue;
            const std::vector<std::string> z = java_split(piece, ' ');
            if (z.size() < 2) return fail("attribute '" + piece + "' without a value" + at + " (ArrayIndexOutOfBoundsException in AnnotationUtils.parseOptionalFields L386)");
            const std::string &k = z[0], &v = z[1];
            if (k == "gene_name") r.gene_name = v, r.has_gene_name = true;
            else if (k == "gene_id") r.gene_id = v, r.has_gene_id = true;
            else if (k == "transcript_name") r.tx_name = v, r.has_tx_name = true;
            else if (k == "transcript_id") r.tx_id = v, r.has_tx_id = true;
            else if (k == "gene_version") {
                if (!java_parse_int(v, &r.version)) return fail("gene_version '" + v + "' is not a number" + at + " (NumberFormatException, GTFParser.java:L130)");
                r.has_version = true;
            }
        }
        r.chrom = f[0];
        r.feature = f[2];
        if (!java_parse_int(f[3], &r.start) || !java_parse_int(f[4], &r.end))
            return fail("start / end is not a number" + at + " (NumberFormatException, GTFParser.java:L117-118)");
        r.negative = f[6] == "-";
        // GTFRecord.validate under ValidationStringency.STRICT ($FilteringGTFParser L121): the AnnotationException leaves GTFReader.load uncaught
        std::string problems;
        if (!r.has_gene_id) problems += " Missing gene_id;";
        if (!r.has_gene_name) problems += " Missing gene_name;";
        if (r.feature != "gene") {
            if (!r.has_tx_name) problems += " Missing transcript_name;";
            if (!r.has_tx_id) problems += " Missing transcript_id;";
        }
        if (r.has_gene_name && r.gene_name.find(',') != std::string::npos) problems += " Reserved character ',' in gene name [" + r.gene_name + "];";
        if (!problems.empty()) return fail("Invalid GTF line" + at + ":" + problems + " (the reference stops here: GTFParser.java:L90-97)");
        if (!ref_index.count(r.chrom)) {  // $FilteringGTFParser.filterOut L126-131: behind the validation
            G->n_skipped_sequence++;
            continue;
        }
        auto it = slot.find(r.gene_name);
        if (it == slot.end()) {
            slot.emplace(r.gene_name, names.size());
            names.push_back(r.gene_name);
            rows.emplace_back();
            rows.back().push_back(std::move(r));
        } else
            rows[it->second].push_back(std::move(r));
    }
    std::vector<int32_t> hs(names.size());
    for (size_t i = 0; i < names.size(); i++) hs[i] = jstring_hash(names[i]);
    for (size_t gi : jhash_order(hs)) {  // gatheredByGene.values().iterator() L49
        // makeGeneFromMultiVersionGTFRecords L72-76: the records of the highest gene_version (none = Integer.MIN_VALUE), in file order
        long long top = -2147483648ll;
        for (const GtfRecord &r : rows[gi]) top = std::max(top, r.has_version ? (long long)r.version : -2147483648ll);
        std::vector<const GtfRecord *> R;
        for (const GtfRecord &r : rows[gi])
            if ((r.has_version ? (long long)r.version : -2147483648ll) == top) R.push_back(&r);
        // makeGeneFromGTFRecords L99-128; every AnnotationException from here on is caught by GTFReader.load L96-100 (LENIENT): the gene is skipped
        const GtfRecord &one = *R[0];
        Gene g;
        g.name = one.gene_name;
        g.negative = one.negative;
        g.contig = ref_index[one.chrom];
        int st = 2147483647, en = -2147483647 - 1;
        bool ok = true;
        std::vector<std::string> gene_ids;
        for (const GtfRecord *r : R) {
            st = std::min(st, r->start);
            en = std::max(en, r->end);
            if (std::find(gene_ids.begin(), gene_ids.end(), r->gene_id) == gene_ids.end()) gene_ids.push_back(r->gene_id);
            if (r->chrom != one.chrom) ok = false;  // "Chromosome disagreement" L116-117
        }
        g.start = st;
        g.end = en;
        for (const GtfRecord *r : R) {  // validateGTFRecord L195-204
            if (r->negative != g.negative) ok = false;                                     // "Strand disagreement"
            if (r->feature == "gene" && (r->start != g.start || r->end != g.end)) ok = false;  // "gene GTFRecord(..) != GeneFromGTF(..)"
        }
        if (gene_ids.size() > 1) ok = false;  // "Multiple gene IDs" L125-126
        // transcripts: the records that are not `gene` features by transcript_id (a HashMap<String, List>: L84, L217), its entries in hash order
        std::vector<std::string> tids;
        std::vector<std::vector<const GtfRecord *>> trecs;
        if (ok)
            for (const GtfRecord *r : R) {
                if (r->feature == "gene") continue;  // $GeneAnnotationFilter L250
                size_t k = 0;
                while (k < tids.size() && tids[k] != r->tx_id) k++;
                if (k == tids.size()) {
                    tids.push_back(r->tx_id);
                    trecs.emplace_back();
                }
                trecs[k].push_back(r);
            }
        std::vector<std::string> tnames;
        std::vector<Transcript> txs;
        if (ok) {
            std::vector<int32_t> ih(tids.size());
            for (size_t i = 0; i < tids.size(); i++) ih[i] = jstring_hash(tids[i]);
            for (size_t k : jhash_order(ih)) {  // addTranscriptToGeneFromGTFRecords L137-188
                const std::vector<const GtfRecord *> &T = trecs[k];
                Transcript t;
                t.name = T[0]->tx_name;
                int ts = 2147483647, te = -2147483647 - 1, cs = 2147483647, ce = -2147483647 - 1;
                for (const GtfRecord *r : T) {
                    if (r->feature == "exon") {
                        t.exons.emplace_back(r->start, r->end);
                        ts = std::min(ts, r->start);
                        te = std::max(te, r->end);
                    }
                    if (r->feature == "CDS") {
                        cs = std::min(cs, r->start);
                        ce = std::max(ce, r->end);
                    }
                }
                std::stable_sort(t.exons.begin(), t.exons.end());  // Collections.sort over $Exon.compareTo (start, then end)
                if (t.exons.empty()) {                             // "<gene>:<transcript> has no exons" L170-171
                    ok = false;
                    break;
                }
                t.tx_start = ts;
                t.tx_end = te;
                t.cds_start = cs == 2147483647 ? ts : cs;
                t.cds_end = ce == -2147483647 - 1 ? te : ce;
                if (std::find(tnames.begin(), tnames.end(), t.name) != tnames.end()) {  // GeneFromGTF.addTranscript L81-82: "appears more than once"
                    ok = false;
                    break;
                }
                for (size_t i = 0; i < t.exons.size(); i++)
                    if (t.exons[i].first > t.exons[i].second || (i > 0 && t.exons[i - 1].second >= t.exons[i].first)) ok = false;  // L180-183
                if (!ok) break;
                tnames.push_back(t.name);
                txs.push_back(std::move(t));
            }
        }
        if (ok && txs.empty()) ok = false;  // "No transcript in GTF for gene" L92-93
        if (!ok) {
            G->n_skipped_genes++;
            continue;
        }
        // GeneFromGTF.iterator() = its own transcripts.values() (a HashMap<String, TranscriptFromGTF> by transcript name, filled in the order above)
        std::vector<int32_t> th(tnames.size());
        for (size_t i = 0; i < tnames.size(); i++) th[i] = jstring_hash(tnames[i]);
        for (size_t k : jhash_order(th)) g.tx.push_back(std::move(txs[k]));
        const uint32_t ih = 31u * (31u * (uint32_t)G->ref_hash[(size_t)g.contig] + (uint32_t)g.start) + (uint32_t)g.end;
        g.hash = (int32_t)(31u * ih + (uint32_t)jstring_hash(g.name));  // GeneFromGTF.hashCode L109-113
        G->genes.push_back(std::move(g));  // (GeneFromGTF.equals includes the name, and the names are distinct: nothing disappears in the OverlapDetector)
    }
    return finish_genes(G, n_refs, "smi_genes_load_gtf", out);
}

extern "C" int smi_genes_free(smi_genes *g) {
    delete g;
    return SMI_OK;
}

extern "C" int smi_genes_count(const smi_genes *g, size_t *n_genes, size_t *n_lines, size_t *n_skipped) {
    if (!g) {
        set_error("smi_genes_count: null argument");
        return SMI_ERR_INVALID;
    }
    if (n_genes) *n_genes = g->genes.size();
    if (n_lines) *n_lines = g->n_lines;
    if (n_skipped) *n_skipped = g->n_skipped_genes + g->n_skipped_sequence;
    return SMI_OK;
}

extern "C" int smi_genes_dump(const smi_genes *g, char *out, size_t cap, size_t *n_out) {
    if (!g || !n_out) {
        set_error("smi_genes_dump: null argument");
        return SMI_ERR_INVALID;
    }
    std::string txt;
    for (const Gene &x : g->genes) {
        txt += x.name + "\t" + g->refs[(size_t)x.contig] + "\t" + std::to_string(x.start) + "\t" + std::to_string(x.end) + "\t" + (x.negative ? "-" : "+") + "\t";
        for (size_t k = 0; k < x.tx.size(); k++) {
            const Transcript &t = x.tx[k];
            txt += (k ? ";" : "") + t.name + "|" + std::to_string(t.tx_start) + "|" + std::to_string(t.tx_end) + "|" + std::to_string(t.cds_start) + "|" +
                   std::to_string(t.cds_end) + "|";
            for (size_t i = 0; i < t.exons.size(); i++) txt += (i ? "," : "") + std::to_string(t.exons[i].first) + "-" + std::to_string(t.exons[i].second);
        }
        txt += "\n";
    }
    *n_out = txt.size();
    if (!out) return SMI_OK;
    if (cap < txt.size()) {
        set_error("smi_genes_dump: output buffer too small");
        return SMI_ERR_INVALID;
    }
    std::memcpy(out, txt.data(), txt.size());
    return SMI_OK;
}

namespace {

// a HashSet<Gene> / HashMap<Gene, .> as the reference fills it: insertion order + hash -> iteration order
struct GeneSet {
    std::vector<int> ids;  // insertion order, distinct
    void add(int g) {
        if (std::find(ids.begin(), ids.end(), g) == ids.end()) ids.push_back(g);
    }
    std::vector<int> iter(const smi_genes &G) const {
        std::vector<int32_t> h(ids.size());
        for (size_t i = 0; i < ids.size(); i++) h[i] = G.genes[(size_t)ids[i]].hash;
        std::vector<int> out;
        for (size_t k : jhash_order(h)) out.push_back(ids[k]);
        return out;
    }
};

int top_scoring(const std::vector<int> &fs) {  // getTopScoringLocusFunction L346-354: first strictly greater score wins; -1 = null
    int best = -1;
    for (int f : fs)
        if (best < 0 || lf_score(f) > lf_score(best)) best = f;
    return best;
}

}  // namespace

extern "C" int smi_gene_tag_chunk(const smi_genes *G, const int32_t *ref_id, const uint16_t *flags, const int32_t *pos0, const uint32_t *cigars,
                                  const uint32_t *cigar_off, int32_t n, char *out, size_t cap, uint32_t *out_off, size_t *n_out) {
    if (!G || !n_out || n < 0 || (n && (!ref_id || !flags || !pos0 || !cigar_off || !out_off))) {
        set_error("smi_gene_tag_chunk: null argument");
        return SMI_ERR_INVALID;
    }
    std::string txt;
    std::vector<uint32_t> offs;
    offs.reserve((size_t)3 * n + 1);
    for (int32_t i = 0; i < n; i++) {
        // alignment blocks (SAMUtils.getAlignmentBlocks) and the read's interval [alignmentStart, alignmentEnd]
        std::vector<Block> blocks;
        const bool unmapped = (flags[i] & 4) != 0 || ref_id[i] < 0;
        int ref = pos0[i] + 1, aln_end = pos0[i];
        if (!unmapped) {
            for (uint32_t k = cigar_off[i]; k < cigar_off[i + 1]; k++) {
                const int op = (int)(cigars[k] & 15u), len = (int)(cigars[k] >> 4);
                if (op == 0 || op == 7 || op == 8) {  // M = X
                    blocks.push_back({ref, len});
                    ref += len;
                } else if (op == 2 || op == 3)  // D N
                    ref += len;
            }
            aln_end = ref - 1;
        }
        // geneOverlapDetector.getOverlaps(readInterval) L231: nodes in (start, end) order, into a HashSet
        GeneSet overlapping;
        if (!unmapped && ref_id[i] < (int32_t)G->by_contig.size() && aln_end >= pos0[i] + 1)
            for (int gi : G->by_contig[(size_t)ref_id[i]]) {
                const Gene &g = G->genes[(size_t)gi];
                if (g.start > aln_end) break;
                if (g.end >= pos0[i] + 1) overlapping.add(gi);
            }
        // map gene -> locus function of the read for that gene (getLocusFunctionForRead(rec, gene) L246-260), a HashMap<Gene, LocusFunction>
        const std::vector<int> over_iter = overlapping.iter(*G);
        GeneSet map_keys;
        std::vector<int> map_val;
        for (int gi : over_iter) {
            const Gene &g = G->genes[(size_t)gi];
            std::vector<int> block_fn;
            for (const Block &b : blocks) {
                std::vector<int> lf((size_t)b.length, INTERGENIC);  // getLocusFunctionsByBlock L358-366
                for (const Transcript &t : g.tx) {
                    const int lo = std::max(b.ref_start, t.tx_start), hi = std::min(t.tx_end, b.ref_start + b.length - 1);
                    for (int p = lo; p <= hi; p++) {  // assignLocusFunctionForRange L151-165
                        int &cur = lf[(size_t)(p - b.ref_start)];
                        if (cur > CODING) continue;
                        const int f = t.in_exon(p) ? (t.utr(p) ? UTR : CODING) : INTRONIC;
                        if (f > cur) cur = f;
                    }
                }
                block_fn.push_back(top_scoring(lf));
            }
            map_keys.add(gi);
            map_val.push_back(top_scoring(block_fn));  // an empty block list gives null; the filter below then drops the gene
        }
        const std::vector<int> map_iter = map_keys.iter(*G);
        auto fn_of = [&](int gi) {
            for (size_t k = 0; k < map_keys.ids.size(); k++)
                if (map_keys.ids[k] == gi) return map_val[k];
            return -1;
        };
        // getConsistentExons(rec, map.keySet(), true) L158-185: per block the genes with an exon that intersects it
        GeneSet exons_for_read;
        for (const Block &b : blocks) {
            GeneSet block_genes;
            for (int gi : map_iter) {
                const Gene &g = G->genes[(size_t)gi];
                bool hit = false;
                for (const Transcript &t : g.tx) {
                    for (const auto &e : t.exons)
                        if (e.first <= b.ref_start + b.length - 1 && b.ref_start <= e.second) {  // CoordMath.overlaps
                            hit = true;
                            break;
                        }
                    if (hit) break;
                }
                if (hit) block_genes.add(gi);
            }
            for (int gi : block_genes.iter(*G)) exons_for_read.add(gi);  // result.addAll(blockGenes)
        }
        // genes whose function is CODING or UTR (lambda$setGeneExons$1 L81-82)
        std::vector<int> genes;
        for (int gi : exons_for_read.iter(*G)) {
            const int f = fn_of(gi);
            if (f == CODING || f == UTR) genes.push_back(gi);
        }
        // f = getLocusFunction(map.values()) L312-316
        int f = INTERGENIC;
        if (!map_keys.ids.empty()) {
            std::vector<int> vals;
            for (int gi : map_iter) vals.push_back(fn_of(gi));
            // a null entry cannot score: LOCUS_FUNCTION_SCORES.get(null) would throw inside annotateGene, which the caller swallows
            // (OneNanoporeSeqAnalyzer.java:L100-102) -- only possible for a mapped read without a single M / = / X operation
            bool has_null = false;
            for (int v : vals) has_null |= v < 0;
            if (has_null) {
                offs.push_back((uint32_t)txt.size());
                offs.push_back((uint32_t)txt.size());
                offs.push_back((uint32_t)txt.size());
                continue;  // no tag touched
            }
            f = top_scoring(vals);
        }
        // getGenesConsistentWithReadStrand L125-154
        const bool neg_read = (flags[i] & 16) != 0;
        std::vector<int> same, opposite;
        for (int gi : genes) (G->genes[(size_t)gi].negative == neg_read ? same : opposite).push_back(gi);
        (void)opposite;  // only counted in the reference's metrics; a read with opposite-strand genes only gets no GE
        std::string ge, gs;
        for (size_t k = 0; k < same.size(); k++) {
            if (k) {
                ge += ",";
                gs += ",";
            }
            ge += G->genes[(size_t)same[k]].name;
            gs += G->genes[(size_t)same[k]].negative ? "-" : "+";
        }
        offs.push_back((uint32_t)txt.size());
        txt += ge;
        offs.push_back((uint32_t)txt.size());
        txt += gs;
        offs.push_back((uint32_t)txt.size());
        txt += LF_NAME[f];
    }
    offs.push_back((uint32_t)txt.size());
    if (g_jhash_unmodelled) {  // a HashSet<Gene> bin of 8 or more genes (never seen: a record overlaps a handful of genes)
        g_jhash_unmodelled = false;
        set_error("smi_gene_tag_chunk: a HashSet<Gene> of the reference would hold a tree bin here: its iteration order is not modelled");
        return SMI_ERR_INVALID;
    }
    *n_out = txt.size();
    if (out_off) std::memcpy(out_off, offs.data(), offs.size() * sizeof(uint32_t));
    if (!out) return SMI_OK;
    if (cap < txt.size()) {
        set_error("smi_gene_tag_chunk: output buffer too small");
        return SMI_ERR_INVALID;
    }
    std::memcpy(out, txt.data(), txt.size());
    return SMI_OK;
}

// the same over an inflated BAM stream and its record index (smi_bam_index_records)
extern "C" int smi_gene_tag_bam(const smi_genes *G, const uint8_t *bam, size_t n_bam, const smi_bam_record *recs, int32_t n, char *out,
                                size_t cap, uint32_t *out_off, size_t *n_out) {
    if (!G || !n_out || n < 0 || (n && (!bam || !recs || !out_off))) {
        set_error("smi_gene_tag_bam: null argument");
        return SMI_ERR_INVALID;
    }
    std::vector<int32_t> rid((size_t)n), p0((size_t)n);
    std::vector<uint16_t> fl((size_t)n);
    std::vector<uint32_t> off((size_t)n + 1, 0), cg;
    for (int32_t i = 0; i < n; i++) {
        const smi_bam_record &r = recs[i];
        if (r.cigar_off + 4ull * r.n_cigar > n_bam) {
            set_error("smi_gene_tag_bam: record outside the stream");
            return SMI_ERR_INVALID;
        }
        rid[(size_t)i] = r.ref_id;
        p0[(size_t)i] = r.pos;
        fl[(size_t)i] = r.flag;
        const size_t at = cg.size();
        cg.resize(at + r.n_cigar);
        if (r.n_cigar) std::memcpy(cg.data() + at, bam + r.cigar_off, 4ull * r.n_cigar);
        off[(size_t)i + 1] = (uint32_t)cg.size();
    }
    if (cg.empty()) cg.push_back(0);
    return smi_gene_tag_chunk(G, rid.data(), fl.data(), p0.data(), cg.data(), off.data(), n, out, cap, out_off, n_out);
}
