// smi_host.hip -- host-side steps of the path that run once per pass, not per read (no device code here).
//
// smi_finalize_used_list: end of pass 1 --
//   UsedCellBCListGenerator$UsedBarcodesListData.finalizeData  FJ!nanoporereadscanner/analyzers/UsedCellBCListGenerator.java:L379-425
//   BarcodeDatasetColissionTester                               FJ!nanoporereadscanner/analyzers/BarcodeDatasetColissionTester.java:L68-229
//   WorkerReadscanner.scan (rank)                               FJ!nanoporereadscanner/WorkerReadscanner.java:L265-273
// Input is the all-reduced pass-1 histogram restricted to non-zero counters (a few 10^4 keys), so this is a small
// host job; the mutation-cycle walk below is the collision-mode variant of BarcodeMatchTester (post sequence absent,
// full matches skipped, descent on hit for substitutions / on miss for indels: BarcodeMatchTester.java:L268,L295,L351).
//
// Canonical order: equal-count ties are broken by ascending key (the reference's order there depends on thread
// timing and on fastutil internals; DESIGN.md section "Pass-1 finalize").
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>

#include "smi_internal.h"
#include "smi_name.h"

namespace smi {

namespace {

// ---- 64-bit mutate ops with Java shift semantics (NucleicAcidTwoBitPerBase.java:L228-233,L300-309,L321-327) ------
inline uint64_t shl(uint64_t v, int s) { return v << (s & 63); }
inline uint64_t shr(uint64_t v, int s) { return v >> (s & 63); }

inline void subs16(uint64_t seq, int pos, uint64_t out[4]) {
    const int sh = (16 - (pos + 1)) << 1;
    seq &= ~shl(3, sh);
    for (uint64_t b = 0; b < 4; b++) out[b] = seq | shl(b, sh);
}
inline void ins16(uint64_t h, int pos, uint64_t out[4]) {
    int sh = (16 - pos - 1) << 1;
    const uint64_t upper = shl(shr(h, sh), sh);
    sh = 64 - sh;
    h = shl(h, sh);
    h = shr(h, sh + 2);  // count 64 at pos 14 wraps to 0: the dropped base stays in bits 62..63
    const int at = 2 * (16 - (pos + 1) - 1);
    for (uint64_t b = 0; b < 4; b++) out[b] = upper | h | shl(b, at);  // A,G,C,T
}
inline uint64_t del16(uint64_t h, uint64_t append2, int pos) {
    int sh = (16 - pos) << 1;
    const uint64_t upper = shl(shr(h, sh), sh);
    sh = 64 - sh;
    h = shl(h, sh + 2);
    h = shr(h, sh);
    return upper | h | append2;
}

struct Item {
    uint64_t seq;
    int16_t prev, pos, level;
};
struct Hit {
    uint64_t bc;
    int ed;
};

// Matches of one barcode against the set, collision mode.  Returns the HashSet content: first hit per level.
void collision_matches(const std::unordered_set<uint64_t> &set, uint64_t seq, int max_ed, std::vector<Hit> &hits) {
    hits.clear();
    std::unordered_set<uint32_t> tested;  // IntHashSet of (int)seq, only when ed >= 2 (NucTwoBitPerBaseEDtesterBase.java:L82-95)
    const bool use_tested = max_ed >= 2;
    auto is_tested = [&](uint64_t s) { return use_tested && tested.count((uint32_t)s) != 0; };
    auto record = [&](uint64_t s, int level) {
        // OneMatch.equals ignores the barcode (BarcodeMatchTester.java:L433-436): one entry per level
        for (const Hit &h : hits)
            if (h.ed == level) return;
        hits.push_back({s, level});
    };
    auto probe = [&](uint64_t s) { return s != seq && set.count(s) != 0; };  // skipFullMatches (L368)
    if (max_ed == 0) return;
    std::vector<Item> dq;
    dq.push_back({seq, -1, -1, 1});
    while (!dq.empty()) {
        Item cur = dq.back();
        dq.pop_back();
        cur.pos++;
        if (cur.pos < 15) dq.push_back(cur);
        if (cur.prev == cur.pos) continue;
        auto descend = [&](uint64_t s) {
            if (max_ed > cur.level) dq.push_back({s, cur.pos, -1, (int16_t)(cur.level + 1)});
        };
        uint64_t v[4];
        subs16(cur.seq, cur.pos, v);
        for (int k = 0; k < 4; k++) {
            if (v[k] == cur.seq || is_tested(v[k])) continue;
            const bool hit = probe(v[k]);
            if (hit) {
                record(v[k], cur.level);
                descend(v[k]);  // substitutions descend on hit
            }
        }
        if (cur.pos < 15) {
            ins16(cur.seq, cur.pos, v);
            for (int k = 0; k < 4; k++) {
                if (is_tested(v[k])) continue;
                const bool hit = probe(v[k]);
                if (hit)
                    record(v[k], cur.level);
                else
                    descend(v[k]);  // indels descend on miss
            }
            const uint64_t m0 = del16(cur.seq, 0, cur.pos);
            for (uint64_t b = 0; b < 4; b++) {  // no post sequence: the four possible appended bases (L336-340)
                const uint64_t s = m0 | b;
                if (is_tested(s)) continue;
                const bool hit = probe(s);
                if (hit)
                    record(s, cur.level);
                else
                    descend(s);
            }
        }
        if (use_tested) tested.insert((uint32_t)cur.seq);
    }
}

struct KC {
    uint64_t key;
    uint32_t count;
};

}  // namespace
}  // namespace smi

using namespace smi;

namespace {
struct Finalized {
    std::vector<KC> f;                       // after the low-count filter, ascending key
    std::vector<std::vector<Hit>> coll;      // per entry of f: its collision matches (one per level at most), empty = none
    std::vector<KC> fin;                     // the used list, by count descending (canonical tie: ascending key)
};

void finalize_core(const uint64_t *keys, const uint32_t *counts, size_t n, uint32_t record_count, int merge_ed, int min_count_fold,
                   int cells_fold_below_max, Finalized &R) {
    const float cutoff = (2.0f * (float)record_count) / 5000000.0f;  // UsedCellBCListGenerator.java:L391
    std::vector<KC> &f = R.f;
    for (size_t i = 0; i < n; i++)
        if ((float)counts[i] > cutoff && counts[i] > 1) f.push_back({keys[i], counts[i]});  // L359-363
    if (f.empty()) return;
    std::sort(f.begin(), f.end(), [](const KC &a, const KC &b) { return a.key < b.key; });
    std::unordered_set<uint64_t> set;
    set.reserve(f.size() * 2);
    for (const KC &e : f) set.insert(e.key);
    auto index_of = [&](uint64_t k) -> long {
        auto it = std::lower_bound(f.begin(), f.end(), k, [](const KC &a, uint64_t v) { return a.key < v; });
        return (it != f.end() && it->key == k) ? (long)(it - f.begin()) : -1;
    };
    R.coll.assign(f.size(), {});
    std::vector<size_t> cs;  // entries with collisions (stored only when non-empty, BarcodeDatasetColissionTester.java:L240-241)
    std::vector<Hit> hits;
    for (size_t i = 0; i < f.size(); i++) {
        collision_matches(set, f[i].key, merge_ed, hits);
        if (!hits.empty()) {
            R.coll[i] = hits;
            cs.push_back(i);
        }
    }
    // entries by count of their key, descending; canonical tie: key ascending (L166-167)
    std::vector<size_t> ord(cs);
    std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) {
        const KC &x = f[a], &y = f[b];
        return x.count != y.count ? x.count > y.count : x.key < y.key;
    });
    // java.util.HashMap<Long, Set<Long>> iteration order of toMergeMap (insertion = ord): buckets ascending at the
    // final capacity, insertion order inside a bucket
    size_t cap = 16;
    while (ord.size() > (cap * 3) / 4) cap <<= 1;
    std::vector<std::pair<uint32_t, size_t>> it(ord.size());
    for (size_t i = 0; i < ord.size(); i++) {
        const uint64_t k = f[ord[i]].key;
        uint32_t h = (uint32_t)(k ^ (k >> 32));
        h ^= h >> 16;
        it[i] = {h & (uint32_t)(cap - 1), i};
    }
    std::stable_sort(it.begin(), it.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    std::vector<uint8_t> alive(f.size(), 1);
    for (const auto &e : it) {  // L188-195
        const size_t self = ord[e.second];
        if (!alive[self]) continue;
        const uint32_t cut = f[self].count / (uint32_t)min_count_fold;  // L168 integer division
        for (const Hit &h : R.coll[self]) {
            if (h.ed > merge_ed) continue;
            const long p = index_of(h.bc);
            if (p >= 0 && f[p].count < cut) alive[p] = 0;  // L171-172, L194
        }
    }
    uint32_t mx = 0;
    for (size_t i = 0; i < f.size(); i++)
        if (alive[i]) mx = std::max(mx, f[i].count);
    const uint32_t min_counts = mx / (uint32_t)cells_fold_below_max;  // L198
    for (size_t i = 0; i < f.size(); i++)
        if (alive[i] && f[i].count >= min_counts) R.fin.push_back(f[i]);
    std::sort(R.fin.begin(), R.fin.end(),
              [](const KC &a, const KC &b) { return a.count != b.count ? a.count > b.count : a.key < b.key; });
}

std::string bc_string(uint64_t key) {  // NucleicAcidTwoBitPerBase.toString of a 16-mer: A 0, G 1, C 2, T 3, first base in the high bits
    std::string s(16, 'A');
    for (int i = 0; i < 16; i++) s[i] = "AGCT"[(key >> (2 * (15 - i))) & 3];
    return s;
}
}  // namespace

extern "C" int smi_finalize_used_list(const uint64_t *keys, const uint32_t *counts, size_t n, uint32_t record_count,
                                      int merge_ed, int min_count_fold, int cells_fold_below_max, uint64_t *out_keys,
                                      uint32_t *out_counts, uint32_t *out_rank, size_t *n_out) {
    if (!n_out || (n && (!keys || !counts || !out_keys || !out_counts || !out_rank)) || merge_ed < 0 || merge_ed > 2 ||
        min_count_fold <= 0 || cells_fold_below_max <= 0) {
        set_error("smi_finalize_used_list: bad argument");
        return SMI_ERR_INVALID;
    }
    *n_out = 0;
    Finalized R;
    finalize_core(keys, counts, n, record_count, merge_ed, min_count_fold, cells_fold_below_max, R);
    for (size_t i = 0; i < R.fin.size(); i++) {
        out_keys[i] = R.fin[i].key;
        out_counts[i] = R.fin[i].count;
        out_rank[i] = (uint32_t)(i + 1);  // WorkerReadscanner.java:L266-270
    }
    *n_out = R.fin.size();
    return SMI_OK;
}

// BarcodeList.tsv (ParseStatsHtmlPrinter.writesedBarcodesListTSV, FJ!nanoporereadscanner/stats/ParseStatsHtmlPrinter.java:L235-285) from the
// same pass-1 counters: header "Barcode\tn Reads with full match" + one "BCs colliding at ED k" column per edit distance that occurs among
// the collision matches (BarcodeDatasetColissionTester.getUnfilteredColissionData L126-144: a TreeMap over the distances); one row per used
// barcode in rank order (LinkedHashMap filled by count descending, UsedCellBCListGenerator.java:L367-371); a collision cell lists the colliding
// barcode as `BC(count x)` when it is itself in the used list and `BC(count m)` (its unfiltered count: merged away) otherwise (L257-263).
// no_whitelist != 0: the run had no list of possible barcodes and rows whose barcode holds AAAAA or TTTTT are left out (L419).
extern "C" int smi_barcode_list_tsv(const uint64_t *keys, const uint32_t *counts, size_t n, uint32_t record_count, int merge_ed,
                                    int min_count_fold, int cells_fold_below_max, int no_whitelist, char *out, size_t cap, size_t *n_out) {
    if (!n_out || (n && (!keys || !counts)) || merge_ed < 0 || merge_ed > 2 || min_count_fold <= 0 || cells_fold_below_max <= 0) {
        set_error("smi_barcode_list_tsv: bad argument");
        return SMI_ERR_INVALID;
    }
    Finalized R;
    finalize_core(keys, counts, n, record_count, merge_ed, min_count_fold, cells_fold_below_max, R);
    bool ed_seen[4] = {false, false, false, false};
    for (const auto &h : R.coll)
        for (const Hit &x : h)
            if (x.ed >= 0 && x.ed < 4) ed_seen[x.ed] = true;
    std::string txt = "Barcode\tn Reads with full match\t";
    bool first = true;
    for (int e = 0; e < 4; e++)
        if (ed_seen[e]) {
            if (!first) txt += "\t";
            txt += "BCs colliding at ED " + std::to_string(e);
            first = false;
        }
    txt += "\n";
    auto idx_f = [&](uint64_t k) -> long {
        auto it = std::lower_bound(R.f.begin(), R.f.end(), k, [](const KC &a, uint64_t v) { return a.key < v; });
        return (it != R.f.end() && it->key == k) ? (long)(it - R.f.begin()) : -1;
    };
    std::unordered_set<uint64_t> used;
    auto shown = [&](uint64_t k) {
        if (!no_whitelist) return true;
        const std::string b = bc_string(k);
        return b.find("TTTTT") == std::string::npos && b.find("AAAAA") == std::string::npos;
    };
    for (const KC &e : R.fin)
        if (shown(e.key)) used.insert(e.key);
    for (const KC &e : R.fin) {
        if (!shown(e.key)) continue;
        txt += bc_string(e.key) + "\t" + std::to_string(e.count);
        const long self = idx_f(e.key);
        for (int ed = 0; ed < 4; ed++) {
            if (!ed_seen[ed]) continue;
            txt += "\t";
            bool any = false;
            for (const Hit &h : R.coll[(size_t)self]) {
                if (h.ed != ed) continue;
                if (any) txt += ",";
                any = true;
                const long p = idx_f(h.bc);
                const uint32_t c = p >= 0 ? R.f[(size_t)p].count : 0;  // unfilteredUsedBarcodeMap count of the colliding barcode
                txt += bc_string(h.bc) + "(" + std::to_string(c) + (used.count(h.bc) ? " x)" : " m)");
            }
        }
        txt += "\n";
    }
    *n_out = txt.size();
    if (!out) return SMI_OK;
    if (cap < txt.size()) {
        set_error("smi_barcode_list_tsv: output buffer too small");
        return SMI_ERR_INVALID;
    }
    std::memcpy(out, txt.data(), txt.size());
    return SMI_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// smi_format_read_name: the read-name suffix `assignumis` later parses = FastqRecordExt.getRecordForWriting
// (FJ!nanoporereadscanner/readerwriter/FastqRecordExt.java:L209-311).  The formatting itself is smi_name.h, which the
// device record writer (smi_write.hip) shares.
// ---------------------------------------------------------------------------------------------------------------
extern "C" int smi_format_read_name(const char *read_name, const char *raw_seq, const char *raw_qual, int32_t len,
                                    const smi_scan_result *scan, const smi_bc_result *bc, int32_t rank, uint32_t read_id,
                                    int five_prime, char *out, size_t cap) {
    if (!read_name || !scan || !out || (len > 0 && (!raw_seq || !raw_qual))) {
        set_error("smi_format_read_name: null argument");
        return SMI_ERR_INVALID;
    }
    NameSink sink{out, 0, cap > 0 ? (int)std::min<size_t>(cap - 1, 1u << 30) : 0};
    for (const char *c = read_name; *c && *c != ' '; c++) sink.put(*c);  // getReadName().split(" ")[0]
    bool quals_set = false;
    const NameWindow w = name_window(*scan, five_prime != 0, len);
    auto at = [&](const char *raw, int k) { return raw[w.rev ? w.lo + w.n_chars - 1 - k : w.lo + k]; };
    const int st = append_name_suffix(
        sink, *scan, bc, rank, read_id, five_prime != 0, len, [&](int k) { return at(raw_seq, k); },
        [&](int k) { return at(raw_qual, k); }, &quals_set);
    if (st == NAME_RANGE) {
        set_error("smi_format_read_name: X=/Q= range outside the read (the reference throws here)");
        return SMI_ERR_INVALID;
    }
    if (cap == 0 || sink.n > sink.cap) {
        set_error("smi_format_read_name: output buffer too small");
        return SMI_ERR_INVALID;
    }
    out[sink.n] = 0;
    return sink.n;
}


// fragment names of a split read (ChimeraFindernew.java:L309,L323)
extern "C" int smi_chimera_fragment_name(const char *read_name, const smi_chimera_result *res, int fragment, char *out,
                                         size_t cap) {
    static const char *const TAGS[] = {"RA", "FA", "RA_FA", "RA_FT", "RT_FA", "RT_FT"};
    if (!read_name || !res || !out || res->n_split == 0 || res->n_split > 2 || fragment < 0 || fragment > res->n_split) {
        set_error("smi_chimera_fragment_name: bad argument");
        return SMI_ERR_INVALID;
    }
    // fragments before a cut carry that cut's tag, the last fragment the tag of the cut it starts at
    const int cut = fragment < res->n_split ? fragment : res->n_split - 1;
    if (res->reason[cut] > 5) {
        set_error("smi_chimera_fragment_name: bad split reason");
        return SMI_ERR_INVALID;
    }
    std::string name(read_name);
    const size_t sp = name.find(' ');
    if (sp != std::string::npos)  // String.replaceFirst(" ", ...): a name without a blank stays as it is
        name.replace(sp, 1, std::string("_") + TAGS[res->reason[cut]] + "sp" + std::to_string(fragment + 1) + " ");
    if (name.size() + 1 > cap) {
        set_error("smi_chimera_fragment_name: output buffer too small");
        return SMI_ERR_INVALID;
    }
    std::memcpy(out, name.c_str(), name.size() + 1);
    return (int)name.size();
}

// BarcodesAssigned.tsv (ParseStatsHtmlPrinter.writeAssignedTSV, FJ!nanoporereadscanner/stats/ParseStatsHtmlPrinter.java:
// L294-327): header `Barcode\tn Reads with ED<=k match\tED=0 .. \tED=k`, one row per barcode that was assigned at least
// once, sorted by its read count, descending; numbers through DecimalFormat("###,###,###,###") (L52; grouping commas).
// The reference sorts the entries of a HashMap with a stable sort, so rows with equal counts come out in that map's
// iteration order, which depends on the insertion history of a run; here they are ordered by ascending barcode key.
extern "C" int smi_assigned_tsv(const uint64_t *keys, const uint32_t *counts, size_t n_keys, int max_ed, char *out, size_t cap,
                                size_t *n_out) {
    if (!n_out || max_ed < 0 || max_ed > 2 || (n_keys && (!keys || !counts))) {
        set_error("smi_assigned_tsv: bad argument");
        return SMI_ERR_INVALID;
    }
    auto grouped = [](unsigned long long v) {
        std::string d = std::to_string(v), r;
        for (size_t i = 0; i < d.size(); i++) {
            if (i && (d.size() - i) % 3 == 0) r += ',';
            r += d[i];
        }
        return r;
    };
    std::vector<size_t> rows;
    std::vector<unsigned long long> tot(n_keys, 0);
    for (size_t i = 0; i < n_keys; i++) {
        tot[i] = (unsigned long long)counts[3 * i] + counts[3 * i + 1] + counts[3 * i + 2];
        if (tot[i]) rows.push_back(i);
    }
    std::stable_sort(rows.begin(), rows.end(), [&](size_t a, size_t b) { return tot[a] != tot[b] ? tot[a] > tot[b] : keys[a] < keys[b]; });
    std::string text = "Barcode\tn Reads with ED<=" + std::to_string(max_ed) + " match";
    for (int e = 0; e <= max_ed; e++) text += "\tED=" + std::to_string(e);
    text += "\n";
    static const char B[4] = {'A', 'G', 'C', 'T'};
    for (size_t i : rows) {
        char bc[17];
        uint64_t k = keys[i];
        for (int j = 15; j >= 0; j--) {
            bc[j] = B[k & 3u];
            k >>= 2;
        }
        bc[16] = 0;
        text += bc;
        text += '\t' + grouped(tot[i]);
        for (int e = 0; e <= max_ed; e++) text += '\t' + (counts[3 * i + e] ? grouped(counts[3 * i + e]) : std::string("0"));
        text += '\n';
    }
    *n_out = text.size();
    if (out) {
        if (text.size() > cap) {
            set_error("smi_assigned_tsv: output buffer too small");
            return SMI_ERR_INVALID;
        }
        std::memcpy(out, text.data(), text.size());
    }
    return SMI_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Scan statistics: the counters behind ReadScanner.html / stats.pojo, as text.
//   ReadFlags$Flags (values, descriptions, print rules)         FJ!nanoporereadscanner/stats/ReadFlags.java:L70-165
//   ReadFlags$Flags.finalizeFlag                                ReadFlags.java:L194-207
//   ReadFlags.addForCounting / mergeStats / generateStatDataForPrinting / print        ReadFlags.java:L229-313
//   Parser.processOneRecord (what is counted, the read-length sums)                     FJ!nanoporereadscanner/analyzers/Parser.java:L92-124
//   Parser.assignBarcode flag bits (BC_FOUND*, BC_ED_DIFF*, BC_OFFSET*), pinned by the `flag` of every record of
//   tests/golden/ref_exec_pass2_*.json (the reference's own flag word, 92 records)
//   ChimeraFindernew (READS_AFTER_SPLIT on fragments, nReadsSplit, MULTI_CHIMERIC_READS_DISCARDED | FAILED)   ChimeraFindernew.java:L284-325
// Not built: the HTML page itself (a Velocity template over these numbers), stats.pojo (Java object serialisation) and the QV histograms.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
struct FlagDef {
    const char *name, *description;
    int bit;          // value = 1 << bit; -1: ALL_READS_AFTER_SPLIT (every bit)
    bool print, only_nonzero;
    int ref;          // index of refForPercentage, -1 none
};
// enum order = ReadFlags$Flags.values()
const FlagDef kFlags[SMI_N_READ_FLAGS] = {
    {"ALL_READS", "All Reads", 0, true, false, -1},
    {"CHIMERIC_READS_SPLIT", "Chimeric reads split", 1, true, true, 0},
    {"MULTI_CHIMERIC_READS_DISCARDED", "Multi Chimeric reads discarded n>3", 2, true, true, 0},
    {"ALL_READS_AFTER_SPLIT", "Reads after chimera split", -1, true, false, -1},
    {"READS_AFTER_SPLIT", "Reads from split chimeric", 3, true, true, 3},
    {"PASSED_TOTAL", "Passed (Adapter found)", 4, true, false, 3},
    {"FAILED", "Adapter NOT found", 5, false, false, 3},
    {"MEAN_LENGTH_PASSED", "Mean read length pA and Adapter found", 6, true, false, -1},
    {"MEAN_LENGTH_FAILED", "Mean read length pA and Adapter NOT found", 7, true, false, -1},
    {"PASSED_FWD", "Passed forward", 8, true, false, 3},
    {"PASSED_REV", "Passed reverse", 9, true, false, 3},
    {"PASSED_TOT_TSO", "Passed total, found TSO other end", 10, true, true, 5},
    {"POLY_T_5P", "PolyT found only at 5\\'", 11, true, true, 3},
    {"POLY_A_3P", "PolyA found only at 3\\'", 12, true, true, 3},
    {"POLY_A_NOT_FOUND", "PolyA not found", 13, true, true, 3},
    {"POLY_T_5P_POLY_A_3P", "PolyT 5\\' and PolyA at 3\\'", 14, true, true, 3},
    {"ADAPTER_5P", "Adapter at 5\\'(3\\' for 5p barcoding)", 15, true, false, 3},
    {"ADAPTER_3P", "Adapter at 3\\'(5\\' for 5p barcoding)", 16, true, false, 3},
    {"TSO_5P", "TSO at 5\\'", 17, true, true, 3},
    {"TSO_3P", "TSO at 3\\'", 18, true, true, 3},
    {"ADAPTER_SELECTED_DESP_ADAPTER_BOTH_SIDES", "Adapter selected despite Adapter both ends", 19, false, false, 3},
    {"READ_TOO_SHORT", "Read too short", 20, true, false, 3},
    {"ADAPTER_5P_AND_3P", "Adapter at 5\\' and 3\\'", 21, true, false, 3},
    {"TSO_5P_AND_3P", "TSO at 5\\' and 3\\'", 22, true, true, 3},
    {"TSO_5P_AND_3P_FAILED", "TSO at 5\\' and 3\\' failed", 23, true, true, 3},
    {"BC_FOUND", "Barcode found", 24, true, true, 5},
    {"BC_FOUND_NO_SECONDARY_MATCH", "Barcode found no secondary at <= ED + 2", 25, true, true, 5},
    {"BC_FOUND_ED0", "Barcode found ED= 0", 26, true, true, 25},
    {"BC_FOUND_ED1", "Barcode found ED= 1", 27, true, true, 25},
    {"BC_FOUND_ED2", "Barcode found ED= 2", 28, true, true, 25},
    {"BC_FOUND_ED3", "Barcode found ED= 3", 29, true, true, 25},
    {"BC_ED_DIFF_ABOVE2", "Barcode secondary match ED diff > 2", 30, true, true, 25},
    {"BC_ED_DIFF1", "Barcode secondary match ED diff 1", 31, true, true, 25},
    {"BC_ED_DIFF2", "Barcode secondary match ED diff 2", 32, true, true, 25},
    {"BC_OFFSET0", "Barcode offset from predicted pos=0", 33, true, true, 25},
    {"BC_OFFSET1", "Barcode offset from predicted pos=+/-1", 34, true, true, 25},
    {"BC_OFFSET2", "Barcode offset from predicted pos=+/-2", 35, true, true, 25},
};
enum { F_ALL = 0, F_CHIM_SPLIT = 1, F_AFTER_SPLIT_ALL = 3, F_READS_AFTER_SPLIT = 4, F_PASSED_TOTAL = 5, F_FAILED = 6, F_MEAN_P = 7, F_MEAN_F = 8 };
}  // namespace

extern "C" uint64_t smi_record_flags(const smi_scan_result *scan, const smi_bc_result *bc, int from_split, int multi_chimeric) {
    if (multi_chimeric) return (1ull << 2) | (1ull << 5);  // never scanned (Parser.java:L92): MULTI_CHIMERIC_READS_DISCARDED | FAILED
    uint64_t f = scan ? (uint64_t)scan->flags : 0;
    if (from_split) f |= 1ull << 3;
    if (bc && bc->found == 1 && scan && scan->found) {
        f |= 1ull << 24;
        if (bc->ed >= 0 && bc->ed <= 3) f |= 1ull << (26 + bc->ed);
        const long long d = (long long)bc->ed_sec - (long long)bc->ed;
        if (d > 2) f |= 1ull << 25;
        f |= d == 1 ? 1ull << 31 : d == 2 ? 1ull << 32 : 1ull << 30;
        const int o = bc->offset < 0 ? -bc->offset : bc->offset;
        if (o <= 2) f |= 1ull << (33 + o);
    }
    // ReadFlags$Flags.finalizeFlag
    const bool fwd = f & (1ull << 8), rev = f & (1ull << 9);
    if (!fwd && !rev)
        f |= 1ull << 5;
    else
        f |= 1ull << 4;
    if ((rev && (f & (1ull << 18))) || (fwd && (f & (1ull << 17)))) f |= 1ull << 10;
    if ((f & (1ull << 5)) && (f & (1ull << 22))) f |= 1ull << 23;
    return f;
}

extern "C" int smi_scan_stats_add(smi_scan_stats *st, const smi_pass2_decisions *dec) {
    if (!st || !dec || (dec->n_records_out && (!dec->scan || !dec->bc || !dec->frag_offsets))) {
        set_error("smi_scan_stats_add: null argument");
        return SMI_ERR_INVALID;
    }
    for (size_t i = 0; i < dec->n_records_out; i++) {
        const uint32_t src = dec->frag_src ? (dec->frag_src[i] >> 2) : (uint32_t)i;
        const smi_chimera_result *ch = dec->chim ? dec->chim + src : nullptr;
        const bool multi = ch && (ch->flags & SMI_CHIM_MULTI), split = ch && ch->n_split > 0 && !multi;
        const uint64_t f = smi_record_flags(dec->scan + i, dec->bc + i, split, multi);
        const uint64_t len = dec->frag_offsets[i + 1] - dec->frag_offsets[i];
        for (int k = 0; k < SMI_N_READ_FLAGS; k++)  // ReadFlags.addForCounting
            if (kFlags[k].bit < 0 || (f & (1ull << kFlags[k].bit))) st->counts[k]++;
        ((f & (1ull << 4)) ? st->sum_len_passed : st->sum_len_failed) += len;
        if (split && (!dec->frag_src || (dec->frag_src[i] & 3u) == 0)) st->n_reads_split++;  // once per split read (ChimeraFindernew.java:L288)
    }
    return SMI_OK;
}

extern "C" int smi_scan_stats_merge(smi_scan_stats *dst, const smi_scan_stats *src) {  // ReadFlags.mergeStats (the `mergestats` sub-command's sum)
    if (!dst || !src) {
        set_error("smi_scan_stats_merge: null argument");
        return SMI_ERR_INVALID;
    }
    for (int k = 0; k < SMI_N_READ_FLAGS; k++) dst->counts[k] += src->counts[k];
    dst->sum_len_passed += src->sum_len_passed;
    dst->sum_len_failed += src->sum_len_failed;
    dst->n_reads_split += src->n_reads_split;
    return SMI_OK;
}

// ReadFlags.print: "=======  Scan Stats =======" and one line `description TAB count TAB percent TAB of <reference>` per printed flag
extern "C" int smi_scan_stats_tsv(const smi_scan_stats *st, char *out, size_t cap, size_t *n_out) {
    if (!st || !n_out) {
        set_error("smi_scan_stats_tsv: null argument");
        return SMI_ERR_INVALID;
    }
    uint64_t c[SMI_N_READ_FLAGS];
    for (int k = 0; k < SMI_N_READ_FLAGS; k++) c[k] = st->counts[k];
    // generateStatDataForPrinting L283-286 (int arithmetic of the reference)
    c[F_ALL] = c[F_AFTER_SPLIT_ALL] - (c[F_READS_AFTER_SPLIT] - st->n_reads_split);
    c[F_CHIM_SPLIT] = st->n_reads_split;
    c[F_MEAN_P] = c[F_PASSED_TOTAL] ? st->sum_len_passed / c[F_PASSED_TOTAL] : 0;  // (the reference divides by zero here: ArithmeticException)
    c[F_MEAN_F] = c[F_FAILED] ? st->sum_len_failed / c[F_FAILED] : 0;
    auto grouped = [](unsigned long long v) {
        std::string d = std::to_string(v), r;
        for (size_t i = 0; i < d.size(); i++) {
            if (i && (d.size() - i) % 3 == 0) r += ',';
            r += d[i];
        }
        return r;
    };
    auto percent = [](float ratio) {  // DecimalFormat("###.0 %"): x 100, one fraction digit, HALF_EVEN, no integer digit when it is zero
        if (ratio != ratio) return std::string("NaN");
        double t = (double)ratio * 100.0 * 10.0;
        if (t > 1e18) return std::string("\xE2\x88\x9E %");
        unsigned long long q = (unsigned long long)t;
        const double frac = t - (double)q;
        if (frac > 0.5 || (frac == 0.5 && (q & 1))) q++;
        std::string r = q / 10 ? std::to_string(q / 10) : std::string();
        r += '.';
        r += (char)('0' + (int)(q % 10));
        return r + " %";
    };
    std::string text = "=======  Scan Stats =======\n\n";
    for (int k = 0; k < SMI_N_READ_FLAGS; k++) {
        const FlagDef &f = kFlags[k];
        if (!f.print || (f.only_nonzero && c[k] == 0)) continue;
        text += f.description;
        text += '\t' + grouped(c[k]) + '\t';
        if (f.ref >= 0) {
            text += percent((float)(int)c[k] / (float)(int)c[f.ref]);
            text += std::string("\tof ") + kFlags[f.ref].description;
        } else
            text += '\t';
        text += '\n';
    }
    *n_out = text.size();
    if (out) {
        if (text.size() > cap) {
            set_error("smi_scan_stats_tsv: output buffer too small");
            return SMI_ERR_INVALID;
        }
        std::memcpy(out, text.data(), text.size());
    }
    return SMI_OK;
}
