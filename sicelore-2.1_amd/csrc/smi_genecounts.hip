// smi_genecounts.hip -- <out>.genecounts.tsv and <out>.UMIdepths.tsv of assignumis (host only).
//
// Replaces GeneCounts (FJ!umifinder/scanstats/GeneCounts.java:L58-652): updateGeneCounts per written record
// (UmiFinderWorker$BamWriters.lambda$writeSams$2 L453-454), printCountTable / printUmisPerCellTable at the end of the run
// (UmiFinderWorker.java:L188-189), mergeGeneCounts.  The reference nests maps gene -> cell -> UMI -> UMIcounts (and region -> cell ->
// UMI -> UMIcounts); here one flat table per kind, keyed (gene | region, cell, UMI), aggregated when a table is printed.
//
// UMIcounts keeps one int: bits 0-11 the number of records, bits 12-15 "records that were a further alignment of their read"
// (GeneCounts.java:L606-652).  Its arithmetic is reproduced as written, including what reads like slips: getExcludingDuplicates is
// `data & (4095 - duplicates)` (an AND, not a difference), increment() stores `data & 7` as the duplicate count, add() combines two
// counters with ANDs and a shift by `12 + (other & 0xf000)` (taken modulo 32 by the JVM).
//
// Where the reference's order is the iteration order of a ConcurrentHashMap filled from a parallel stream (cells or genes with equal
// totals) the rows here are by ascending key (cells: 2-bit code, genes: name); a read with several gene names counts for the first
// one (the reference draws one with ThreadLocalRandom, L439).
#include <algorithm>
#include <cstring>
#include <iterator>
#include <string>
#include <unordered_map>
#include <vector>

#include "smi_internal.h"

using namespace smi;

namespace {

struct Key {
    uint64_t group;  // gene index or region number
    uint64_t cell, umi;
    bool operator==(const Key &o) const { return group == o.group && cell == o.cell && umi == o.umi; }
};
struct KeyHash {
    size_t operator()(const Key &k) const {
        uint64_t h = k.group * 0x9E3779B97F4A7C15ull;
        h = (h ^ (h >> 29)) + k.cell * 0xBF58476D1CE4E5B9ull;
        h = (h ^ (h >> 31)) + k.umi * 0x94D049BB133111EBull;
        return (size_t)(h ^ (h >> 32));
    }
};
using Table = std::unordered_map<Key, int32_t, KeyHash>;

// GeneCounts$UMIcounts
inline int32_t excluding_duplicates(int32_t d) { return d & (4095 - ((int32_t)((uint32_t)(d & 61440) >> 12))); }  // L617
inline void increment(int32_t &d, bool nth) {                                                                       // L638-641
    d = (int32_t)((uint32_t)d + 1u);
    if (nth) d = (d & 4095) | ((d & 7) << 12);
}
inline void add_counts(int32_t &d, int32_t c) {                                                                     // L650
    const uint32_t sh = (uint32_t)(12 + (c & 61440)) & 31u;
    const uint32_t inner = ((uint32_t)(d & 61440) >> sh) >> 12;
    d = (d & (int32_t)(4095u + (uint32_t)c)) & (int32_t)((4095u + inner) << 12);
}

}  // namespace

struct smi_gene_counts {
    std::vector<std::string> gene_names;
    std::unordered_map<std::string, uint32_t> gene_index;
    Table genes, regions;
    int64_t records_with_gene = 0, records_skipped_clipping = 0;
};

namespace {

// NucleicAcidTwoBitPerBase.longTwoBitToString (TB!nuc/encoding/TwoBit/NucleicAcidTwoBitPerBase.java:L337-342)
std::string two_bit_string(uint64_t v, int len) {
    std::string s((size_t)len, 'A');
    for (int i = len - 1; i >= 0; i--, v >>= 2) s[(size_t)i] = "AGCT"[v & 3];
    return s;
}

struct CellSum {
    uint64_t cell;
    int32_t n;
};

// getUMIperCellCounts(type) L98-111 + the sort of generateUMIperCellDataForUMIDepthPlot L133 / printCountTable L305: per cell the number
// of (group, UMI) entries whose counter is not empty "excluding duplicates", cells with the larger number first
std::vector<CellSum> umis_per_cell(const Table &t) {
    std::unordered_map<uint64_t, int32_t> m;
    for (const auto &e : t) m[e.first.cell] += excluding_duplicates(e.second) != 0 ? 1 : 0;
    std::vector<CellSum> v;
    v.reserve(m.size());
    for (const auto &e : m) v.push_back({e.first, e.second});
    std::sort(v.begin(), v.end(), [](const CellSum &a, const CellSum &b) { return a.n != b.n ? a.n > b.n : a.cell < b.cell; });
    return v;
}

int give(const std::string &text, char *out, size_t cap, size_t *n_out, const char *who) {
    *n_out = text.size();
    if (out) {
        if (text.size() > cap) {
            set_error(std::string(who) + ": output buffer too small");
            return SMI_ERR_INVALID;
        }
        std::memcpy(out, text.data(), text.size());
    }
    return SMI_OK;
}

}  // namespace

extern "C" int smi_gene_counts_create(smi_gene_counts **out) {
    if (!out) {
        set_error("smi_gene_counts_create: null argument");
        return SMI_ERR_INVALID;
    }
    *out = new smi_gene_counts();
    return SMI_OK;
}

extern "C" int smi_gene_counts_free(smi_gene_counts *gc) {
    delete gc;
    return SMI_OK;
}

extern "C" int smi_gene_counts_add(smi_gene_counts *gc, size_t n, const char *const *gene, const int64_t *region, const uint64_t *cell_bc,
                                   const uint64_t *umi, const uint8_t *has_bc_umi, const uint16_t *flag, const uint8_t *mapq,
                                   const uint32_t *first_cigar, const uint32_t *last_cigar, const uint8_t *nth_record, int five_prime) {
    if (!gc || (n && (!region || !cell_bc || !umi || !has_bc_umi || !flag || !mapq || !first_cigar || !last_cigar || !nth_record))) {
        set_error("smi_gene_counts_add: null argument");
        return SMI_ERR_INVALID;
    }
    for (size_t i = 0; i < n; i++) {
        // L375-378: unmapped, secondary or supplementary, mapping quality 0
        if ((flag[i] & 0x4) || (flag[i] & 0x900) || mapq[i] == 0) continue;
        // L384-417: the CIGAR element on the barcode's side of the alignment (first element for a forward 5' read or a reversed 3' read,
        // last element otherwise) is a clip of more than 150 bases
        const bool reversed = (flag[i] & 0x10) != 0;
        const uint32_t ce = (five_prime ? !reversed : reversed) ? first_cigar[i] : last_cigar[i];
        if (first_cigar[i] != 0xFFFFFFFFu) {  // 0xFFFFFFFF: the record has no CIGAR
            const uint32_t op = ce & 15u, len = ce >> 4;
            if (len > 150 && (op == 5 || op == 4)) {  // H, S
                gc->records_skipped_clipping++;
                continue;
            }
        }
        const bool do_gene = gene && gene[i];
        const bool do_region = region[i] >= 0;
        if (!has_bc_umi[i]) continue;  // L424-427: no BC or no U8 attribute
        const bool nth = nth_record[i] != 0;
        if (do_gene) {
            gc->records_with_gene++;
            auto it = gc->gene_index.find(gene[i]);
            uint32_t g;
            if (it == gc->gene_index.end()) {
                g = (uint32_t)gc->gene_names.size();
                gc->gene_names.emplace_back(gene[i]);
                gc->gene_index.emplace(gc->gene_names.back(), g);
            } else
                g = it->second;
            increment(gc->genes[Key{g, cell_bc[i], umi[i]}], nth);
        }
        if (do_region) increment(gc->regions[Key{(uint64_t)region[i], cell_bc[i], umi[i]}], nth);
    }
    return SMI_OK;
}

extern "C" int smi_gene_counts_merge(smi_gene_counts *dst, const smi_gene_counts *src) {
    if (!dst || !src) {
        set_error("smi_gene_counts_merge: null argument");
        return SMI_ERR_INVALID;
    }
    // mergeGeneCounts L540-592: entries the first object lacks are taken over, counters of the same (gene, cell, UMI) go through add()
    for (const auto &e : src->genes) {
        const std::string &name = src->gene_names[(size_t)e.first.group];
        auto it = dst->gene_index.find(name);
        uint32_t g;
        if (it == dst->gene_index.end()) {
            g = (uint32_t)dst->gene_names.size();
            dst->gene_names.push_back(name);
            dst->gene_index.emplace(name, g);
        } else
            g = it->second;
        const Key k{g, e.first.cell, e.first.umi};
        auto f = dst->genes.find(k);
        if (f == dst->genes.end())
            dst->genes.emplace(k, e.second);
        else
            add_counts(f->second, e.second);
    }
    // regions L567-591: the test `retval.containsKey(regionID)` asks the GENE map (String keys) for a Long, which is never there, so a region
    // of a later object REPLACES the first object's data for that region number
    if (!src->regions.empty()) {
        std::unordered_map<uint64_t, char> replaced;
        for (const auto &e : src->regions) replaced.emplace(e.first.group, 1);
        for (auto it = dst->regions.begin(); it != dst->regions.end();)
            it = replaced.count(it->first.group) ? dst->regions.erase(it) : std::next(it);
        for (const auto &e : src->regions) dst->regions.emplace(e.first, e.second);
    }
    return SMI_OK;
}

// ---- shards of one run (assignumis split by chromosome over ranks / GPUs) ------------------------------------------------------
// dump / load: the object as bytes (it crosses process borders); merge_shard: the tables of a LATER shard folded into an earlier one so that
// the result is what one process would have counted over both.  That is not mergeGeneCounts (its add() is an AND of the counters, see
// above): counters of one (gene, cell, UMI) key that both shards hold -- the same gene name on chromosomes of two shards -- are ADDED, as
// consecutive increments do; if the later shard's counter carries the "further alignment" bits the order of the increments matters and
// the call reports the key in *n_order_dependent instead of guessing.  Region numbers must not collide (each shard numbers its regions
// from its own base).
extern "C" int smi_gene_counts_dump(const smi_gene_counts *gc, uint8_t *out, size_t cap, size_t *n_out) {
    if (!gc || !n_out) {
        set_error("smi_gene_counts_dump: null argument");
        return SMI_ERR_INVALID;
    }
    std::string b;
    auto put = [&](const void *p, size_t n) { b.append(static_cast<const char *>(p), n); };
    auto put64 = [&](uint64_t v) { put(&v, 8); };
    put("SMIGC001", 8);
    put64((uint64_t)gc->records_with_gene);
    put64((uint64_t)gc->records_skipped_clipping);
    put64(gc->gene_names.size());
    for (const auto &nm : gc->gene_names) {
        put64(nm.size());
        put(nm.data(), nm.size());
    }
    for (const Table *t : {&gc->genes, &gc->regions}) {
        put64(t->size());
        for (const auto &e : *t) {
            put64(e.first.group);
            put64(e.first.cell);
            put64(e.first.umi);
            put64((uint64_t)(uint32_t)e.second);
        }
    }
    *n_out = b.size();
    if (out) {
        if (b.size() > cap) {
            set_error("smi_gene_counts_dump: output buffer too small");
            return SMI_ERR_INVALID;
        }
        std::memcpy(out, b.data(), b.size());
    }
    return SMI_OK;
}

extern "C" int smi_gene_counts_load(const uint8_t *data, size_t n, smi_gene_counts **out) {
    if (!data || !out) {
        set_error("smi_gene_counts_load: null argument");
        return SMI_ERR_INVALID;
    }
    size_t at = 0;
    bool ok = n >= 8 && !std::memcmp(data, "SMIGC001", 8);
    at = 8;
    auto get64 = [&](uint64_t &v) {
        if (!ok || at + 8 > n) {
            ok = false;
            v = 0;
            return;
        }
        std::memcpy(&v, data + at, 8);
        at += 8;
    };
    smi_gene_counts *gc = new smi_gene_counts();
    uint64_t v = 0, cnt = 0;
    get64(v);
    gc->records_with_gene = (int64_t)v;
    get64(v);
    gc->records_skipped_clipping = (int64_t)v;
    get64(cnt);
    for (uint64_t i = 0; ok && i < cnt; i++) {
        uint64_t len = 0;
        get64(len);
        if (!ok || len > n - at) {
            ok = false;
            break;
        }
        gc->gene_names.emplace_back(reinterpret_cast<const char *>(data + at), (size_t)len);
        gc->gene_index.emplace(gc->gene_names.back(), (uint32_t)i);
        at += (size_t)len;
    }
    for (Table *t : {&gc->genes, &gc->regions}) {
        get64(cnt);
        for (uint64_t i = 0; ok && i < cnt; i++) {
            Key k{0, 0, 0};
            uint64_t c = 0;
            get64(k.group);
            get64(k.cell);
            get64(k.umi);
            get64(c);
            if (ok && t == &gc->genes && k.group >= gc->gene_names.size()) ok = false;
            if (ok) t->emplace(k, (int32_t)(uint32_t)c);
        }
    }
    if (!ok || at != n) {
        delete gc;
        set_error("smi_gene_counts_load: not a dump of smi_gene_counts_dump (or truncated)");
        return SMI_ERR_INVALID;
    }
    *out = gc;
    return SMI_OK;
}

extern "C" int smi_gene_counts_merge_shard(smi_gene_counts *dst, const smi_gene_counts *later, size_t *n_order_dependent) {
    if (!dst || !later) {
        set_error("smi_gene_counts_merge_shard: null argument");
        return SMI_ERR_INVALID;
    }
    size_t bad = 0;
    for (const auto &e : later->genes) {
        const std::string &name = later->gene_names[(size_t)e.first.group];
        auto it = dst->gene_index.find(name);
        uint32_t g;
        if (it == dst->gene_index.end()) {
            g = (uint32_t)dst->gene_names.size();
            dst->gene_names.push_back(name);
            dst->gene_index.emplace(name, g);
        } else
            g = it->second;
        const Key k{g, e.first.cell, e.first.umi};
        auto f = dst->genes.find(k);
        if (f == dst->genes.end())
            dst->genes.emplace(k, e.second);
        else if ((e.second & 61440) == 0 && (((f->second & 4095) + (e.second & 4095)) & ~4095) == 0)
            f->second = (int32_t)((uint32_t)f->second + (uint32_t)e.second);  // plain increments commute: the count goes up by the later shard's
        else
            bad++;
    }
    for (const auto &e : later->regions) {
        if (dst->regions.count(e.first)) {
            set_error("smi_gene_counts_merge_shard: both shards hold region " + std::to_string(e.first.group) + " (give every shard its own region base)");
            return SMI_ERR_INVALID;
        }
    }
    for (const auto &e : later->regions) dst->regions.emplace(e.first, e.second);
    dst->records_with_gene += later->records_with_gene;
    dst->records_skipped_clipping += later->records_skipped_clipping;
    if (n_order_dependent) *n_order_dependent = bad;
    return SMI_OK;
}

extern "C" int smi_gene_counts_info(const smi_gene_counts *gc, int64_t *records_with_gene, int64_t *records_skipped_clipping, size_t *n_genes,
                                    size_t *n_gene_entries, size_t *n_region_entries) {
    if (!gc) {
        set_error("smi_gene_counts_info: null argument");
        return SMI_ERR_INVALID;
    }
    if (records_with_gene) *records_with_gene = gc->records_with_gene;
    if (records_skipped_clipping) *records_skipped_clipping = gc->records_skipped_clipping;
    if (n_genes) *n_genes = gc->gene_names.size();
    if (n_gene_entries) *n_gene_entries = gc->genes.size();
    if (n_region_entries) *n_region_entries = gc->regions.size();
    return SMI_OK;
}

// printCountTable L307-357
extern "C" int smi_gene_counts_tsv(const smi_gene_counts *gc, int bc_length, char *out, size_t cap, size_t *n_out) {
    if (!gc || !n_out || bc_length < 1 || bc_length > 32) {
        set_error("smi_gene_counts_tsv: bad argument");
        return SMI_ERR_INVALID;
    }
    const std::vector<CellSum> cells = umis_per_cell(gc->genes);
    std::unordered_map<uint64_t, uint32_t> column;
    column.reserve(cells.size() * 2);
    for (size_t c = 0; c < cells.size(); c++) column.emplace(cells[c].cell, (uint32_t)c);
    const size_t n_genes = gc->gene_names.size(), n_cells = cells.size();
    // per gene and cell the number of UMI entries (umis.entrySet().size() L346: every entry, empty counters included); per gene their sum
    std::vector<std::vector<std::pair<uint32_t, int32_t>>> rows(n_genes);  // (column, count), filled through a per-gene map
    std::vector<int64_t> total(n_genes, 0);
    {
        std::vector<std::unordered_map<uint32_t, int32_t>> acc(n_genes);
        for (const auto &e : gc->genes) acc[(size_t)e.first.group][column.at(e.first.cell)]++;
        for (size_t g = 0; g < n_genes; g++) {
            rows[g].assign(acc[g].begin(), acc[g].end());
            for (const auto &p : rows[g]) total[g] += p.second;
        }
    }
    std::vector<uint32_t> order;
    for (size_t g = 0; g < n_genes; g++)
        if (!rows[g].empty()) order.push_back((uint32_t)g);  // a gene name known from a merge only has entries, too; kept for safety
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return total[a] != total[b] ? total[a] > total[b] : gc->gene_names[a] < gc->gene_names[b];
    });
    std::string text;
    text.reserve((n_cells + 1) * (size_t)(bc_length + 1) + order.size() * (2 * n_cells + 16));
    text += '\t';
    for (size_t c = 0; c < n_cells; c++) {
        if (c) text += '\t';
        text += two_bit_string(cells[c].cell, bc_length);
    }
    text += '\n';
    std::vector<int32_t> line(n_cells);
    for (uint32_t g : order) {
        std::fill(line.begin(), line.end(), 0);
        for (const auto &p : rows[g]) line[p.first] = p.second;
        text += gc->gene_names[g];
        text += '\t';
        for (size_t c = 0; c < n_cells; c++) {
            if (c) text += '\t';
            text += std::to_string(line[c]);
        }
        text += '\n';
    }
    return give(text, out, cap, n_out, "smi_gene_counts_tsv");
}

// printUmisPerCellTable L256-284: rank, UMIs of the cell at that rank by genomic region, UMIs of the cell at that rank by gene
extern "C" int smi_umi_depths_tsv(const smi_gene_counts *gc, char *out, size_t cap, size_t *n_out) {
    if (!gc || !n_out) {
        set_error("smi_umi_depths_tsv: null argument");
        return SMI_ERR_INVALID;
    }
    const std::vector<CellSum> by_region = umis_per_cell(gc->regions), by_gene = umis_per_cell(gc->genes);
    std::string text = "Cell\tnUMIs based on genomic regions\tnUMIs based on genes\n";
    for (size_t i = 0; i < by_region.size(); i++) {
        text += std::to_string(i + 1) + '\t' + std::to_string(by_region[i].n);
        if (i < by_gene.size()) text += '\t' + std::to_string(by_gene[i].n);
        text += '\n';
    }
    return give(text, out, cap, n_out, "smi_umi_depths_tsv");
}
