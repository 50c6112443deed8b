// smi_bamwrite.hip -- the writer half of `assignumis` on the host's threads: one batch of records -> the bytes of <out>.bam and
// <out>_umifound_.bam (uncompressed BAM records; BGZF is smi_bgzf_deflate[_device]'s).
//
// Replaces, per record of a batch (UmiFinderWorker$BamWriters.writeSams, FJ!umifinder/UmiFinderWorker.java:L408-495):
//   * the order: a stable sort of the batch by htsjdk's SAMRecordCoordinateComparator (L421; SAMRecordCoordinateComparator.java:L48-105),
//   * the tags: ReadScanResult.writeSamFlags + writeBCSamFlags (FJ!nanoporereadscanner/readerwriter/ReadScanResult.java:L205-237, L254-279,
//     called from OneNanoporeSeqAnalyzer.call L95, L146) from the scan data in the read name (FastqRecordExt.getScanDatFromReadName,
//     FastqRecordExt.java:L395-496), GennameTagger's XF / GE / GS, ClusterOneBase.setSamflagsAndStatsForClustered (ClusterOneBase.java:
//     L145-164) or the U7 of UmiFinderWorker.lambda$new$0 (L248-255) with the U7 -> U8 + UZ fill of lambda$writeSams$2 (L442-447),
//   * the attribute list as htsjdk keeps it: ordered by binary tag from the moment it is decoded (BinaryTagCodec.readTags L271-305 builds it
//     through SAMBinaryTagAndValue.insert L207-228), a repeated tag keeping its last value, integers written in the smallest type
//     (BinaryTagCodec.getIntegerType L153-180), setAttribute(tag, null) removing,
//   * which records are written: those with a cell barcode; <out>_umifound_.bam holds the ones whose UMI comes from clustering; -w cuts the
//     read name at its first '_' (L431-432),
//   * GeneCounts.updateGeneCounts for every written record that ends up with U8 (L453-454), in write order.
// The Python mirror of the same rules (assignumis.py: record_tag_sets / apply_tag_sets / _coordinate_key) is what the reference-executed
// fixtures pin (ref_exec_samtags / auxorder / bamorder); tests hold this file to it byte for byte.
#include <algorithm>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_set>
#include <vector>

#include <cctype>

#include "smi_internal.h"

using namespace smi;

namespace {

// FastqRecordExt.lambda$getScanDatFromReadName$4 (L397-408): text behind the first `tag` up to the next '_'
bool extract(const char *sub, size_t n, const char *tag, const char **val, size_t *len) {
    const size_t tl = std::strlen(tag);
    const char *hit = (const char *)memmem(sub, n, tag, tl);
    if (!hit) return false;
    const char *a = hit + tl, *end = sub + n;
    const char *b = (const char *)std::memchr(a, '_', (size_t)(end - a));
    *val = a;
    *len = (size_t)((b ? b : end) - a);
    return true;
}

bool to_long(const char *v, size_t n, int base, long *out) {
    if (n == 0 || n > 18) return false;
    char buf[24];
    std::memcpy(buf, v, n);
    buf[n] = 0;
    char *end = nullptr;
    const long x = std::strtol(buf, &end, base);
    if (end != buf + n) return false;
    *out = x;
    return true;
}

struct ScanName {
    bool present = false, reverse = false;
    long ae = 0, ps = 0, pe = 0, tso = 0;
    bool has_ps = false, has_pe = false, has_tso = false;
    bool has_bc = false;  // the barcode fields were taken (ed present and within the limit)
    const char *seq = nullptr;
    size_t seq_len = 0;
    long ed = 0, ed_sec = 0, start = 0, end = 0, rank = 0, read_id = 0;
    bool has_ed_sec = false, has_start = false, has_end = false, has_rank = false;
};

// the whole of getScanDatFromReadName that the tags need; false = the reference throws (no AE=, a number that does not parse)
bool parse_scan_name(const char *name, size_t n, int bc_edit_limit, ScanName &d, std::string &err) {
    d = ScanName();
    const char *mark = (const char *)memmem(name, n, "_REV_", 5);
    d.reverse = mark != nullptr;
    if (!mark) mark = (const char *)memmem(name, n, "_FWD_", 5);
    if (!mark) return true;  // Optional.absent()
    const char *sub = mark + 4;
    const size_t sn = (size_t)(name + n - sub);
    const char *v;
    size_t l;
    auto number = [&](const char *tag, long *out, bool *has) {
        if (!extract(sub, sn, tag, &v, &l)) return true;
        if (!to_long(v, l, 10, out)) {
            err = std::string("a number does not parse in a read name (") + tag + ")";
            return false;
        }
        *has = true;
        return true;
    };
    bool has_ae = false;
    if (!number("AE=", &d.ae, &has_ae)) return false;
    if (!has_ae) {
        err = "adapter position (AE=) not found in a read name (AdapterInfoNotFoundInReadException)";
        return false;
    }
    d.present = true;
    if (!number("PS=", &d.ps, &d.has_ps) || !number("PE=", &d.pe, &d.has_pe) || !number("T=", &d.tso, &d.has_tso)) return false;
    bool has_ed = false;
    if (!number("ed=", &d.ed, &has_ed)) return false;
    if (has_ed && (bc_edit_limit < 0 || d.ed <= bc_edit_limit)) {
        d.has_bc = true;
        if (extract(sub, sn, "bc=", &v, &l)) {
            d.seq = v;
            d.seq_len = l;
        }
        if (!number("ed_sec=", &d.ed_sec, &d.has_ed_sec) || !number("bcStart=", &d.start, &d.has_start) || !number("bcEnd=", &d.end, &d.has_end) ||
            !number("rk=", &d.rank, &d.has_rank))
            return false;
    }
    // the read id: what follows the last '_' (NumberToAndFromAscii.convertString, base 36; L492-494)
    size_t last = sn;
    while (last > 0 && sub[last - 1] != '_') last--;
    if (last < sn) {
        size_t e = last;
        while (e < sn && sub[e] != ' ') e++;
        if (!to_long(sub + last, e - last, 36, &d.read_id)) {
            err = "the read id behind the last '_' of a read name is not a base-36 number";
            return false;
        }
    }
    return true;
}

// ---- attribute list -----------------------------------------------------------------------------------------------------
struct Field {
    uint16_t key;  // (second char << 8) | first char
    std::string raw;
};
inline uint16_t tag_key(const char *t) { return (uint16_t)(((uint8_t)t[1] << 8) | (uint8_t)t[0]); }

std::string aux_int(const char *tag, long long v) {  // BinaryTagCodec.getIntegerType L153-180
    std::string r(tag, 2);
    auto put = [&](char ty, int bytes) {
        r += ty;
        for (int i = 0; i < bytes; i++) r += (char)((uint64_t)v >> (8 * i));
    };
    if (v >= -128 && v <= 127)
        put('c', 1);
    else if (v >= 0 && v <= 255)
        put('C', 1);
    else if (v >= -32768 && v <= 32767)
        put('s', 2);
    else if (v >= 0 && v <= 65535)
        put('S', 2);
    else if (v >= -2147483648ll && v <= 2147483647ll)
        put('i', 4);
    else
        put('I', 4);
    return r;
}
std::string aux_str(const char *tag, const char *s, size_t n) {
    std::string r(tag, 2);
    r += 'Z';
    r.append(s, n);
    r += '\0';
    return r;
}
inline std::string aux_str(const char *tag, const std::string &s) { return aux_str(tag, s.data(), s.size()); }

struct Attributes {
    std::vector<Field> f;
    void set(std::string raw) {
        const uint16_t k = tag_key(raw.data());
        for (Field &x : f)
            if (x.key == k) {
                x.raw = std::move(raw);
                return;
            }
        f.push_back({k, std::move(raw)});
    }
    void remove(const char *tag) {
        const uint16_t k = tag_key(tag);
        for (size_t i = 0; i < f.size(); i++)
            if (f[i].key == k) {
                f.erase(f.begin() + (long)i);
                return;
            }
    }
    const Field *find(const char *tag) const {
        const uint16_t k = tag_key(tag);
        for (const Field &x : f)
            if (x.key == k) return &x;
        return nullptr;
    }
};

// the input's attribute bytes as htsjdk holds them after decoding (integers boxed -> smallest type on writing; H -> byte array)
bool read_aux(const uint8_t *aux, size_t n, Attributes &a, std::string &err) {
    size_t p = 0;
    while (p < n) {
        if (p + 3 > n) {
            err = "truncated attribute in a BAM record";
            return false;
        }
        const char tag[2] = {(char)aux[p], (char)aux[p + 1]};
        const uint8_t ty = aux[p + 2];
        size_t q;
        auto fixed = [&](size_t bytes) { return p + 3 + bytes; };
        switch (ty) {
        case 'A': q = fixed(1); break;
        case 'c': case 'C': q = fixed(1); break;
        case 's': case 'S': q = fixed(2); break;
        case 'i': case 'I': case 'f': q = fixed(4); break;
        case 'Z': case 'H': {
            q = p + 3;
            while (q < n && aux[q]) q++;
            q++;
            break;
        }
        case 'B': {
            if (p + 8 > n) {
                err = "truncated array attribute in a BAM record";
                return false;
            }
            const uint8_t sub = aux[p + 3];
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
            if (!es) {
                err = "unknown BAM array element type";
                return false;
            }
            uint32_t cnt;
            std::memcpy(&cnt, aux + p + 4, 4);
            q = p + 8 + es * (size_t)cnt;
            break;
        }
        default:
            err = "unknown BAM aux type";
            return false;
        }
        if (q > n) {
            err = "truncated attribute in a BAM record";
            return false;
        }
        long long v = 0;
        bool is_int = true;
        switch (ty) {
        case 'c': v = (int8_t)aux[p + 3]; break;
        case 'C': v = aux[p + 3]; break;
        case 's': { int16_t t; std::memcpy(&t, aux + p + 3, 2); v = t; break; }
        case 'S': { uint16_t t; std::memcpy(&t, aux + p + 3, 2); v = t; break; }
        case 'i': { int32_t t; std::memcpy(&t, aux + p + 3, 4); v = t; break; }
        case 'I': { uint32_t t; std::memcpy(&t, aux + p + 3, 4); v = t; break; }
        default: is_int = false;
        }
        if (is_int)
            a.set(aux_int(tag, v));
        else if (ty == 'H') {
            std::string r(tag, 2);
            const size_t hl = q - 1 - (p + 3);
            if (hl & 1) {
                err = "odd hex string attribute in a BAM record";
                return false;
            }
            r += "Bc";
            const uint32_t cnt = (uint32_t)(hl / 2);
            r.append((const char *)&cnt, 4);
            auto hex = [](uint8_t c) { return c <= '9' ? c - '0' : (c | 32) - 'a' + 10; };
            for (size_t i = 0; i < hl; i += 2) r += (char)((hex(aux[p + 3 + i]) << 4) | hex(aux[p + 4 + i]));
            a.set(std::move(r));
        } else
            a.set(std::string((const char *)aux + p, q - p));
        p = q;
    }
    return true;
}

// NucleicAcidTwoBitPerBase(String).getSequence() (TB!nuc/encoding/TwoBit/NucleicAcidTwoBitPerBase.java:L183-187, L448-450)
uint64_t two_bit_code(const char *s, size_t n) {
    uint64_t v = 0;
    for (size_t i = 0; i < n; i++) {
        int64_t c;
        switch (s[i]) {
        case 'A': case 'a': c = 0; break;
        case 'G': case 'g': c = 1; break;
        case 'C': case 'c': c = 2; break;
        case 'T': case 't': c = 3; break;
        default: c = -2;
        }
        v = (v << 2) | (uint64_t)c;
    }
    return v;
}

struct CountRow {
    std::string gene;
    bool has_gene = false;
    int64_t region;
    uint64_t cell, umi;
    uint8_t has_bc_umi, mapq, nth;
    uint16_t flag;
    uint32_t first_cigar, last_cigar;
};

struct Piece {
    std::string bc, umi;
    std::vector<CountRow> rows;
    std::string err;
};

}  // namespace

extern "C" int smi_bam_write_default_config(smi_bam_write_config *cfg) {
    if (!cfg) {
        set_error("smi_bam_write_default_config: null argument");
        return SMI_ERR_INVALID;
    }
    std::memset(cfg, 0, sizeof *cfg);
    cfg->bc_edit_limit = -1;
    cfg->n_threads = 4;
    cfg->gene_tag[0] = 'G';
    cfg->gene_tag[1] = 'E';
    return SMI_OK;
}

extern "C" int smi_bam_write_batch(const uint8_t *bam, size_t n_bam, const smi_bam_record *recs, const int32_t *batch, int32_t n_batch,
                                   const smi_umi_tag *tags, const char *gene, const uint32_t *gene_off, const smi_bam_write_config *cfg,
                                   uint8_t *out_bc, size_t cap_bc, size_t *n_bc, uint8_t *out_umi, size_t cap_umi, size_t *n_umi,
                                   int32_t *order_out, smi_gene_counts *gc, const int64_t *region, const uint8_t *nth_record) {
    if (!bam || !recs || (!batch && n_batch) || !tags || !cfg || !n_bc || !n_umi || n_batch < 0 || (gene && !gene_off) ||
        (gc && (!region || !nth_record))) {
        set_error("smi_bam_write_batch: bad argument");
        return SMI_ERR_INVALID;
    }
    // -g: two letters (UmiFinderMain.java:L241; all zero = a caller that zeroed the structure itself: GE)
    char gene_tag[3] = {cfg->gene_tag[0] ? cfg->gene_tag[0] : 'G', cfg->gene_tag[0] ? cfg->gene_tag[1] : 'E', 0};
    if (!std::isalpha((unsigned char)gene_tag[0]) || !std::isalpha((unsigned char)gene_tag[1]) || cfg->gene_tag[2]) {
        set_error("smi_bam_write_batch: the gene name attribute (-g) should have two letters");
        return SMI_ERR_INVALID;
    }
    for (int32_t k = 0; k < n_batch; k++) {
        const smi_bam_record &r = recs[batch[k]];
        if (r.aux_off + r.aux_len > n_bam || r.name_off + r.l_read_name > n_bam || r.cigar_off + 4ull * r.n_cigar > n_bam ||
            r.name_off != r.rec_off + 36 || r.aux_off < r.name_off + r.l_read_name + 4ull * r.n_cigar) {  // the layout smi_bam_index_records reports
            set_error("smi_bam_write_batch: a record index entry points outside the BAM buffer");
            return SMI_ERR_INVALID;
        }
    }
    // ---- the order of the batch: SAMRecordCoordinateComparator, stable
    std::vector<int32_t> order(batch, batch + n_batch);
    auto name_of = [&](int32_t i, size_t *len) {
        *len = recs[i].l_read_name ? recs[i].l_read_name - 1u : 0u;
        return (const char *)bam + recs[i].name_off;
    };
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        const smi_bam_record &x = recs[a], &y = recs[b];
        const int64_t rx = x.ref_id >= 0 ? x.ref_id : (1ll << 30), ry = y.ref_id >= 0 ? y.ref_id : (1ll << 30);
        if (rx != ry) return rx < ry;
        const int64_t px = x.ref_id >= 0 ? x.pos : 0, py = y.ref_id >= 0 ? y.pos : 0;  // two records without a reference: no position compare
        if (px != py) return px < py;
        const bool sx = (x.flag & 16) != 0, sy = (y.flag & 16) != 0;
        if (sx != sy) return !sx;  // forward first
        size_t lx, ly;
        const char *nx = name_of(a, &lx), *ny = name_of(b, &ly);
        const int c = std::memcmp(nx, ny, std::min(lx, ly));  // String.compareTo: first differing character, then the length
        if (c) return c < 0;
        if (lx != ly) return lx < ly;
        if (x.flag != y.flag) return x.flag < y.flag;
        if (x.mapq != y.mapq) return x.mapq < y.mapq;
        if (x.next_ref_id != y.next_ref_id) return x.next_ref_id < y.next_ref_id;  // plain integers: -1 first
        if (x.next_pos != y.next_pos) return x.next_pos < y.next_pos;
        return x.tlen < y.tlen;
    });
    if (order_out) std::memcpy(order_out, order.data(), sizeof(int32_t) * (size_t)n_batch);

    const int nt = std::max(1, std::min<int>(cfg->n_threads, std::max(1, n_batch / 256)));
    std::vector<Piece> pieces((size_t)nt);
    auto work = [&](int t) {
        Piece &pc = pieces[(size_t)t];
        const size_t lo = (size_t)n_batch * (size_t)t / (size_t)nt, hi = (size_t)n_batch * (size_t)(t + 1) / (size_t)nt;
        Attributes at;
        ScanName d;
        size_t bound = 64;
        for (size_t k = lo; k < hi; k++) bound += (size_t)recs[order[k]].rec_len + 420;
        pc.bc.reserve(bound);  // no re-allocation while the piece grows
        pc.umi.reserve(bound);
        for (size_t k = lo; k < hi; k++) {
            const int32_t i = order[k];
            const smi_bam_record &r = recs[i];
            size_t nl;
            const char *name = name_of(i, &nl);
            if (!parse_scan_name(name, nl, cfg->bc_edit_limit, d, pc.err)) return;
            if (!d.present || !d.has_bc || !d.seq) continue;  // no cell barcode: not written (L435)
            at.f.clear();
            if (!read_aux(bam + r.aux_off, r.aux_len, at, pc.err)) return;
            // ---- writeSamFlags L205-237
            if (d.has_pe) {
                at.set(aux_int("PE", d.pe));
                if (d.has_ps)
                    at.set(aux_int("PS", d.ps));
                else
                    at.remove("PS");
            }
            at.set(aux_int("AE", d.ae));
            if (d.reverse) at.set(aux_str("RE", "", 0));
            if (d.has_tso) at.set(aux_int("TE", d.tso));
            const std::string start = std::to_string(d.start), end = std::to_string(d.end), rank = std::to_string(d.rank);
            at.set(aux_str("BU", d.seq, d.seq_len));
            if (d.has_start) at.set(aux_str("BV", start));
            if (d.has_end) at.set(aux_str("BE", end));
            at.set(aux_int("BW", d.ed));
            if (d.has_ed_sec) at.set(aux_str("BX", d.ed_sec == 2147483647 ? std::string("N.A.") : std::to_string(d.ed_sec)));
            at.set(aux_str("SX", std::to_string(d.read_id)));
            if (d.has_rank) at.set(aux_str("BH", rank));
            // ---- writeBCSamFlags(sam, flags, false, false) L254-279
            at.set(aux_str("BC", d.seq, d.seq_len));
            if (d.has_start) at.set(aux_str("BB", start));
            if (d.has_end) at.set(aux_str("BF", end));
            at.set(aux_int("B1", d.ed));
            if (d.has_ed_sec) at.set(aux_str("B2", std::to_string(d.ed_sec)));
            at.set(aux_str("BZ", d.seq, d.seq_len));
            // ---- GennameTagger.annotateGene (OneNanoporeSeqAnalyzer L98): XF, then GE + GS or their removal; an empty XF = it threw, nothing touched
            if (gene) {
                const uint32_t *o = gene_off + 3 * (size_t)i;
                const char *ge = gene + o[0], *gs = gene + o[1], *xf = gene + o[2];
                const size_t ge_n = o[1] - o[0], gs_n = o[2] - o[1], xf_n = o[3] - o[2];
                if (xf_n) {
                    at.set(aux_str("XF", xf, xf_n));
                    if (ge_n && gs_n) {
                        at.set(aux_str(gene_tag, ge, ge_n));
                        at.set(aux_str("GS", gs, gs_n));
                    } else {
                        at.remove(gene_tag);
                        at.remove("GS");
                    }
                }
            }
            // ---- the UMI: from clustering (ClusterOneBase L145-164), or the read's own 12 bases (lambda$new$0 L248-255; U8 := U7 + UZ
            //      unless DONT_ASSIGN_UMI, lambda$writeSams$2 L442-447)
            const smi_umi_tag &u = tags[i];
            const bool clustered = (u.flags & SMI_UMI_CLUSTERED) != 0;
            if (clustered) {
                at.set(aux_str("U8", u.u8, strnlen(u.u8, 12)));
                at.set(aux_str("U7", u.u7, strnlen(u.u7, 12)));
                at.set(aux_str("UC", "", 0));
                at.set(aux_str("U1", std::to_string((int)u.u1)));
                if (u.u2 >= 0) at.set(aux_str("U2", std::to_string((int)u.u2)));
            } else if (u.flags & SMI_UMI_HAS_U7) {
                at.set(aux_str("U7", u.u7, strnlen(u.u7, 12)));
                if (!(u.flags & SMI_UMI_SKIPPED)) {
                    at.set(aux_str("U8", u.u7, strnlen(u.u7, 12)));
                    at.set(aux_str("UZ", "", 0));
                }
            }
            std::sort(at.f.begin(), at.f.end(), [](const Field &a, const Field &b) { return a.key < b.key; });
            // ---- the record: fixed part (read name cut with -w), CIGAR / SEQ / QUAL as they are, the attribute list
            const uint8_t *body = bam + r.rec_off + 4;
            const size_t fixed_n = (size_t)(r.aux_off - (r.rec_off + 4));
            std::string rec;
            size_t name_cut = r.l_read_name;
            if (cfg->truncate_read_name) {
                size_t c = 0;
                while (c < nl && name[c] != '_') c++;
                name_cut = c + 1;
            }
            size_t total = fixed_n - r.l_read_name + name_cut;
            for (const Field &x : at.f) total += x.raw.size();
            rec.reserve(total + 4);
            const uint32_t bs = (uint32_t)total;
            rec.append((const char *)&bs, 4);
            rec.append((const char *)body, 32);
            if (cfg->truncate_read_name) {
                rec[4 + 8] = (char)name_cut;  // l_read_name
                rec.append(name, name_cut - 1);
                rec += '\0';
            } else
                rec.append((const char *)body + 32, r.l_read_name);
            rec.append((const char *)body + 32 + r.l_read_name, fixed_n - 32 - r.l_read_name);
            for (const Field &x : at.f) rec += x.raw;
            pc.bc += rec;
            if (clustered) pc.umi += rec;
            // ---- GeneCounts.updateGeneCounts (L453-454): records that carry U8 now
            if (gc) {
                const Field *fu = at.find("U8"), *fb = at.find("BC"), *fg = at.find(gene_tag);
                auto z = [](const Field *f, const char **s, size_t *n) {
                    if (!f || f->raw[2] != 'Z') return false;
                    *s = f->raw.data() + 3;
                    *n = f->raw.size() - 4;
                    return true;
                };
                const char *s;
                size_t n;
                CountRow row;
                row.region = region[i];
                row.flag = r.flag;
                row.mapq = r.mapq;
                row.nth = nth_record[i];
                const bool hu = z(fu, &s, &n);
                row.umi = hu ? two_bit_code(s, n) : 0;
                const bool hb = z(fb, &s, &n);
                row.cell = hb ? two_bit_code(s, n) : 0;
                row.has_bc_umi = hu && hb;
                if (z(fg, &s, &n)) {  // geneNames = GE.split(","): trailing empty strings dropped, "" -> [""]; the first name counts
                    if (n == 0) {
                        row.has_gene = true;
                    } else {
                        size_t e = n;
                        while (e > 0 && s[e - 1] == ',') e--;
                        if (e > 0) {
                            size_t c = 0;
                            while (c < e && s[c] != ',') c++;
                            row.gene.assign(s, c);
                            row.has_gene = true;
                        }
                    }
                }
                if (r.n_cigar) {
                    std::memcpy(&row.first_cigar, bam + r.cigar_off, 4);
                    std::memcpy(&row.last_cigar, bam + r.cigar_off + 4ull * (r.n_cigar - 1u), 4);
                } else {
                    row.first_cigar = 0xFFFFFFFFu;
                    row.last_cigar = 0;
                }
                if (hu) pc.rows.push_back(std::move(row));
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &t : pool) t.join();
    size_t tb = 0, tu = 0, tr = 0;
    for (const Piece &pc : pieces) {
        if (!pc.err.empty()) {
            set_error("smi_bam_write_batch: " + pc.err);
            return SMI_ERR_INVALID;
        }
        tb += pc.bc.size();
        tu += pc.umi.size();
        tr += pc.rows.size();
    }
    *n_bc = tb;
    *n_umi = tu;
    if ((out_bc && tb > cap_bc) || (out_umi && tu > cap_umi)) {
        set_error("smi_bam_write_batch: output buffer too small");
        return SMI_ERR_INVALID;
    }
    {  // every thread copies its own piece to its place (the pages of a fresh output buffer are touched in parallel, too)
        std::vector<size_t> ab((size_t)nt + 1, 0), au((size_t)nt + 1, 0);
        for (int t = 0; t < nt; t++) {
            ab[(size_t)t + 1] = ab[(size_t)t] + pieces[(size_t)t].bc.size();
            au[(size_t)t + 1] = au[(size_t)t] + pieces[(size_t)t].umi.size();
        }
        auto place = [&](int t) {
            const Piece &pc = pieces[(size_t)t];
            if (out_bc) std::memcpy(out_bc + ab[(size_t)t], pc.bc.data(), pc.bc.size());
            if (out_umi) std::memcpy(out_umi + au[(size_t)t], pc.umi.data(), pc.umi.size());
        };
        std::vector<std::thread> copiers;
        for (int t = 1; t < nt; t++) copiers.emplace_back(place, t);
        place(0);
        for (auto &t : copiers) t.join();
    }
    if (gc && tr && out_bc) {  // counted once, with the bytes (a size query does not count)
        std::vector<const char *> g(tr);
        std::vector<int64_t> reg(tr);
        std::vector<uint64_t> cell(tr), umi(tr);
        std::vector<uint8_t> has(tr), mq(tr), nth(tr);
        std::vector<uint16_t> fl(tr);
        std::vector<uint32_t> c0(tr), c1(tr);
        size_t k = 0;
        for (const Piece &pc : pieces)
            for (const CountRow &row : pc.rows) {
                g[k] = row.has_gene ? row.gene.c_str() : nullptr;
                reg[k] = row.region;
                cell[k] = row.cell;
                umi[k] = row.umi;
                has[k] = row.has_bc_umi;
                mq[k] = row.mapq;
                nth[k] = row.nth;
                fl[k] = row.flag;
                c0[k] = row.first_cigar;
                c1[k] = row.last_cigar;
                k++;
            }
        return smi_gene_counts_add(gc, tr, g.data(), reg.data(), cell.data(), umi.data(), has.data(), fl.data(), mq.data(), c0.data(), c1.data(),
                                   nth.data(), cfg->five_prime);
    }
    return SMI_OK;
}

// names and CIGARs of the records idx[0 .. n) back to back, in the layout smi_assignumis_chunk takes (name_off / cigar_off: n + 1 entries);
// out buffers NULL: sizes only (*n_name_bytes, *n_cigar_ops)
extern "C" int smi_bam_chunk_inputs(const uint8_t *bam, size_t n_bam, const smi_bam_record *recs, const int32_t *idx, int32_t n, char *names,
                                    uint32_t *name_off, uint32_t *cigars, uint32_t *cigar_off, uint16_t *flags, int32_t *pos0,
                                    size_t *n_name_bytes, size_t *n_cigar_ops) {
    if (!bam || !recs || (!idx && n) || n < 0 || !n_name_bytes || !n_cigar_ops) {
        set_error("smi_bam_chunk_inputs: bad argument");
        return SMI_ERR_INVALID;
    }
    size_t nb = 0, nc = 0;
    for (int32_t k = 0; k < n; k++) {
        const smi_bam_record &r = recs[idx[k]];
        if (r.name_off + r.l_read_name > n_bam || r.cigar_off + 4ull * r.n_cigar > n_bam) {
            set_error("smi_bam_chunk_inputs: a record index entry points outside the BAM buffer");
            return SMI_ERR_INVALID;
        }
        const size_t nl = r.l_read_name ? r.l_read_name - 1u : 0u;
        if (names) {
            std::memcpy(names + nb, bam + r.name_off, nl);
            name_off[k] = (uint32_t)nb;
            std::memcpy(cigars + nc, bam + r.cigar_off, 4ull * r.n_cigar);
            cigar_off[k] = (uint32_t)nc;
            flags[k] = r.flag;
            pos0[k] = r.pos;
        }
        nb += nl;
        nc += r.n_cigar;
    }
    if (names) {
        name_off[n] = (uint32_t)nb;
        cigar_off[n] = (uint32_t)nc;
    }
    *n_name_bytes = nb;
    *n_cigar_ops = nc;
    return SMI_OK;
}

// nth[i] = 1 when a record with the same read name comes earlier in recs (OneNanoporeSeqAnalyzer.call L74-80: the read's statistics object
// exists already, so this is a further record of the read -- isNthRecordForRead)
extern "C" int smi_bam_name_seen(const uint8_t *bam, size_t n_bam, const smi_bam_record *recs, int32_t n, uint8_t *nth) {
    if (!bam || !recs || !nth || n < 0) {
        set_error("smi_bam_name_seen: bad argument");
        return SMI_ERR_INVALID;
    }
    std::unordered_set<std::string_view> seen;
    seen.reserve((size_t)n * 2);
    for (int32_t i = 0; i < n; i++) {
        const smi_bam_record &r = recs[i];
        if (r.name_off + r.l_read_name > n_bam) {
            set_error("smi_bam_name_seen: a record index entry points outside the BAM buffer");
            return SMI_ERR_INVALID;
        }
        const std::string_view nm((const char *)bam + r.name_off, r.l_read_name ? r.l_read_name - 1u : 0u);
        nth[i] = seen.insert(nm).second ? 0 : 1;
    }
    return SMI_OK;
}

// the same over a stream of segments: the set of read names seen so far lives in a handle; records [from, n) of this segment are looked up and
// added (records in front of `from` were handed in with an earlier segment)
struct smi_name_set {
    std::unordered_set<uint64_t> hashes;  // the reference keys its per-read statistics by a 64-bit hash of the name as well
                                          // (AllReadsScanStats.getLongHashFromString, OneNanoporeResult.java:L192)
};
namespace {
inline uint64_t name_hash(const uint8_t *p, size_t n) {  // eight bytes per step: multiply, fold the high half down, next word
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, p + i, 8);
        h = (h ^ w) * 0xD6E8FEB86659FD93ull;
        h ^= h >> 29;
    }
    uint64_t w = 0;
    if (i < n) std::memcpy(&w, p + i, n - i);
    h = (h ^ w) * 0xD6E8FEB86659FD93ull;
    h ^= h >> 32;
    h *= 0xCA5A826395121157ull;
    return h ^ (h >> 29);
}
}  // namespace
extern "C" int smi_name_set_create(smi_name_set **out) {
    if (!out) {
        set_error("smi_name_set_create: null argument");
        return SMI_ERR_INVALID;
    }
    *out = new smi_name_set();
    return SMI_OK;
}
extern "C" int smi_name_set_free(smi_name_set *s) {
    delete s;
    return SMI_OK;
}
extern "C" int smi_name_set_seen(smi_name_set *set, const uint8_t *bam, size_t n_bam, const smi_bam_record *recs, int32_t from, int32_t n, uint8_t *nth) {
    if (!set || !bam || !recs || !nth || from < 0 || n < from) {
        set_error("smi_name_set_seen: bad argument");
        return SMI_ERR_INVALID;
    }
    set->hashes.reserve(set->hashes.size() + (size_t)(n - from));
    for (int32_t i = from; i < n; i++) {
        const smi_bam_record &r = recs[i];
        if (r.name_off + r.l_read_name > n_bam) {
            set_error("smi_name_set_seen: a record index entry points outside the BAM buffer");
            return SMI_ERR_INVALID;
        }
        nth[i] = set->hashes.insert(name_hash(bam + r.name_off, r.l_read_name ? r.l_read_name - 1u : 0u)).second ? 0 : 1;
    }
    return SMI_OK;
}
