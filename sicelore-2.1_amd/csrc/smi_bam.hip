// smi_bam.hip -- BAM ingest for `assignumis` (host C++): BGZF inflate and the BAM record index.
//
// Replaces what the reference gets from htsjdk in BamReader.open / run (FJ!umifinder/bamreaders/BamReader.java:L82-158:
// SamReaderFactory.open + SAMRecordIterator): the container (SAM spec 4.1 BGZF: concatenated gzip members with a 'BC'
// extra field carrying the block size) and the record layout (SAM spec 4.2).  htsjdk 2.x is a jar dependency that is
// absent from /root/reference; the formats are restated from the published specification.  Blocks are independent, so
// they are inflated on n_threads host threads (zlib); records are length-prefixed, so their index is one sequential
// pass (it runs at memory speed; nothing here belongs on the device).
#include <dlfcn.h>
#include <zlib.h>  // types and constants only: libz is NOT linked -- smi_bgzf_deflate (the host-side BGZF writer with zlib's levels, a legacy path) looks it up with
                   // dlopen on its first call, so that loading libsicelore_mi.so (a JNI System.loadLibrary) needs no libz; everything else inflates and
                   // deflates with the library's own code (smi_inflate_host.hip, smi_deflate.hip)

#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "smi_internal.h"

using namespace smi;

namespace {

inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct Block {
    size_t off;       // of the gzip member in the input
    uint32_t bsize;   // total member size
    uint32_t xlen;
    uint32_t isize;   // uncompressed size
    size_t out_off;
};

// returns SMI_OK, or SMI_ERR_INVALID with the error text set; *consumed = bytes of complete blocks
int scan_blocks(const uint8_t *in, size_t n_in, std::vector<Block> &blocks, size_t *total, size_t *consumed) {
    size_t off = 0, out = 0;
    while (off < n_in) {
        if (n_in - off < 18) break;  // incomplete header: the caller may append more bytes
        const uint8_t *h = in + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) {
            set_error("smi_bgzf: not a BGZF block (gzip magic / FEXTRA missing) at offset " + std::to_string(off));
            return SMI_ERR_INVALID;
        }
        const uint32_t xlen = rd16(h + 10);
        if (n_in - off < 12 + (size_t)xlen) break;
        // the 'BC' subfield may sit anywhere in the extra field
        uint32_t bsize = 0;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *sf = h + 12 + x;
            const uint32_t slen = rd16(sf + 2);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (uint32_t)rd16(sf + 4) + 1;
            x += 4 + slen;
        }
        if (bsize == 0 || bsize < 12 + xlen + 8) {
            set_error("smi_bgzf: block without a valid BC subfield at offset " + std::to_string(off));
            return SMI_ERR_INVALID;
        }
        if (n_in - off < bsize) break;  // incomplete block
        Block b{off, bsize, xlen, rd32(h + bsize - 4), out};
        blocks.push_back(b);
        out += b.isize;
        off += bsize;
    }
    *total = out;
    *consumed = off;
    return SMI_OK;
}

}  // namespace

extern "C" int smi_bgzf_uncompressed_size(const uint8_t *in, size_t n_in, size_t *n_out, size_t *n_blocks, size_t *consumed) {
    if ((!in && n_in) || !n_out) {
        set_error("smi_bgzf_uncompressed_size: null argument");
        return SMI_ERR_INVALID;
    }
    std::vector<Block> blocks;
    size_t total = 0, used = 0;
    if (int rc = scan_blocks(in, n_in, blocks, &total, &used)) return rc;
    *n_out = total;
    if (n_blocks) *n_blocks = blocks.size();
    if (consumed) *consumed = used;
    return SMI_OK;
}

extern "C" int smi_bgzf_inflate(const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *n_out, size_t *consumed,
                                int n_threads) {
    if ((!in && n_in) || (!out && cap_out) || !n_out) {
        set_error("smi_bgzf_inflate: null argument");
        return SMI_ERR_INVALID;
    }
    std::vector<Block> blocks;
    size_t total = 0, used = 0;
    if (int rc = scan_blocks(in, n_in, blocks, &total, &used)) return rc;
    if (total > cap_out) {
        set_error("smi_bgzf_inflate: output buffer too small (" + std::to_string(total) + " bytes needed)");
        return SMI_ERR_INVALID;
    }
    std::atomic<size_t> next{0};
    std::atomic<int> bad{-1};
    auto work = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= blocks.size() || bad.load() >= 0) break;
            const Block &b = blocks[k];
            const uint8_t *payload = in + b.off + 12 + b.xlen;
            const size_t n_payload = b.bsize - 12 - b.xlen - 8;
            const bool ok = host_inflate_exact(payload, n_payload, out + b.out_off, b.isize) == 0 &&
                            host_crc32(0, out + b.out_off, b.isize) == rd32(in + b.off + b.bsize - 8);
            if (!ok) {
                bad = (int)k;
                break;
            }
        }
    };
    const int nt = std::max(1, std::min<int>(n_threads, (int)blocks.size()));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (bad.load() >= 0) {
        set_error("smi_bgzf_inflate: corrupt block " + std::to_string(bad.load()) + " (inflate / length / CRC32)");
        return SMI_ERR_INVALID;
    }
    *n_out = total;
    if (consumed) *consumed = used;
    return SMI_OK;
}

// BAM header (SAM spec 4.2): magic "BAM\1", l_text, text, n_ref, then per reference l_name, name (NUL-terminated), l_ref
extern "C" int smi_bam_header(const uint8_t *bam, size_t n, uint64_t *text_off, uint32_t *text_len, int32_t *n_ref,
                              uint64_t *ref_name_off, uint32_t *ref_name_len, int32_t *ref_len, size_t cap_ref,
                              uint64_t *records_off) {
    if (!bam || !text_off || !text_len || !n_ref || !records_off) {
        set_error("smi_bam_header: null argument");
        return SMI_ERR_INVALID;
    }
    if (n < 12 || std::memcmp(bam, "BAM\1", 4) != 0) {
        set_error("smi_bam_header: not a BAM stream (magic)");
        return SMI_ERR_INVALID;
    }
    const uint32_t l_text = rd32(bam + 4);
    if ((size_t)l_text + 12 > n) {
        set_error("smi_bam_header: truncated header text");
        return SMI_ERR_INVALID;
    }
    *text_off = 8;
    *text_len = l_text;
    size_t p = 8 + (size_t)l_text;
    const int32_t nr = (int32_t)rd32(bam + p);
    p += 4;
    if (nr < 0) {
        set_error("smi_bam_header: negative reference count");
        return SMI_ERR_INVALID;
    }
    *n_ref = nr;
    for (int32_t r = 0; r < nr; r++) {
        if (p + 4 > n) {
            set_error("smi_bam_header: truncated reference list");
            return SMI_ERR_INVALID;
        }
        const uint32_t l_name = rd32(bam + p);
        if (p + 4 + (size_t)l_name + 4 > n || l_name == 0) {
            set_error("smi_bam_header: truncated reference list");
            return SMI_ERR_INVALID;
        }
        if ((size_t)r < cap_ref) {
            if (ref_name_off) ref_name_off[r] = p + 4;
            if (ref_name_len) ref_name_len[r] = l_name - 1;  // without the NUL
            if (ref_len) ref_len[r] = (int32_t)rd32(bam + p + 4 + l_name);
        }
        p += 4 + (size_t)l_name + 4;
    }
    *records_off = p;
    return SMI_OK;
}

// Index of the alignment records from `start` on; stops in front of an incomplete record (*end_off = its offset, so the
// caller can carry the tail over to the next buffer).  All offsets are into `bam`.
extern "C" int smi_bam_index_records(const uint8_t *bam, size_t n, uint64_t start, smi_bam_record *recs, size_t cap,
                                     size_t *n_recs, uint64_t *end_off) {
    if (!bam || !n_recs || !end_off || (!recs && cap)) {
        set_error("smi_bam_index_records: null argument");
        return SMI_ERR_INVALID;
    }
    size_t p = start, k = 0;
    while (p + 4 <= n) {
        const uint32_t block_size = rd32(bam + p);
        if (p + 4 + (size_t)block_size > n) break;
        if (block_size < 32) {
            set_error("smi_bam_index_records: record shorter than its fixed part at offset " + std::to_string(p));
            return SMI_ERR_INVALID;
        }
        if (k >= cap) break;
        const uint8_t *r = bam + p + 4;
        smi_bam_record &o = recs[k];
        o.ref_id = (int32_t)rd32(r);
        o.pos = (int32_t)rd32(r + 4);
        o.l_read_name = r[8];
        o.mapq = r[9];
        o.n_cigar = rd16(r + 12);
        o.flag = rd16(r + 14);
        o.l_seq = (int32_t)rd32(r + 16);
        o.next_ref_id = (int32_t)rd32(r + 20);
        o.next_pos = (int32_t)rd32(r + 24);
        o.tlen = (int32_t)rd32(r + 28);
        o.rec_off = p;
        o.rec_len = block_size + 4;
        const uint64_t name_off = p + 36;
        const uint64_t cigar_off = name_off + o.l_read_name;
        const uint64_t seq_off = cigar_off + 4ull * o.n_cigar;
        const uint64_t qual_off = seq_off + ((uint64_t)(o.l_seq < 0 ? 0 : o.l_seq) + 1) / 2;
        const uint64_t aux_off = qual_off + (uint64_t)(o.l_seq < 0 ? 0 : o.l_seq);
        if (o.l_seq < 0 || o.l_read_name == 0 || aux_off > p + 4 + block_size) {
            set_error("smi_bam_index_records: inconsistent record at offset " + std::to_string(p));
            return SMI_ERR_INVALID;
        }
        o.name_off = name_off;
        o.cigar_off = cigar_off;
        o.seq_off = seq_off;
        o.qual_off = qual_off;
        o.aux_off = aux_off;
        o.aux_len = (uint32_t)(p + 4 + block_size - aux_off);
        k++;
        p += 4 + (size_t)block_size;
    }
    *n_recs = k;
    *end_off = p;
    return SMI_OK;
}

// Plain gzip (RFC 1952, possibly several members, as `cat a.gz b.gz` or bgzip produce): the FASTQ inputs of `scanfastq`
// (FastqFileReader opens *.gz through htsjdk's FastqReader / GZIPInputStream, FJ!nanoporereadscanner/readerwriter/
// FastqFileReader.java:L138-150).  One stream is inherently serial; files are independent, so the caller inflates several
// at once.  Two-call protocol: out == NULL returns the inflated size in *n_out.
extern "C" int smi_gz_inflate_into(const uint8_t *in, size_t n_in, size_t *in_pos, uint8_t *out, size_t cap_out, size_t *out_pos) {
    if ((!in && n_in) || !in_pos || !out_pos || (!out && cap_out) || *in_pos > n_in || *out_pos > cap_out) {
        set_error("smi_gz_inflate_into: bad argument");
        return SMI_ERR_INVALID;
    }
    return host_gunzip(in, n_in, in_pos, out, cap_out, out_pos);
}

extern "C" int smi_gz_inflate(const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *n_out) {
    if ((!in && n_in) || !n_out) {
        set_error("smi_gz_inflate: null argument");
        return SMI_ERR_INVALID;
    }
    if (out) {  // the library's own decoder (smi_inflate_host.hip)
        size_t in_pos = 0, out_pos = 0;
        const int rc = host_gunzip(in, n_in, &in_pos, out, cap_out, &out_pos);
        if (rc == 1) {
            set_error("smi_gz_inflate: output buffer too small");
            return SMI_ERR_INVALID;
        }
        *n_out = out_pos;
        return rc;
    }
    // size query (out == NULL): the same decoder into a scratch buffer that is doubled until the stream fits (round 6: no zlib here any more)
    std::vector<uint8_t> sink(std::max<size_t>(4 * n_in + (1u << 16), 1u << 20));
    for (;;) {
        size_t in_pos = 0, out_pos = 0;
        const int rc = host_gunzip(in, n_in, &in_pos, sink.data(), sink.size(), &out_pos);
        if (rc == 1) {  // output buffer too small
            sink.resize(sink.size() * 2);
            continue;
        }
        if (rc != SMI_OK) return rc;
        *n_out = out_pos;
        return SMI_OK;
    }
}

// BGZF writer: `in` cut into blocks of block_bytes (<= 0xFF00) uncompressed bytes, each deflated on its own (zlib level
// `level`), plus the 28-byte empty EOF block.  Replaces htsjdk's BlockCompressedOutputStream under BAMFileWriter
// (UmiFinderWorker$OneBamWriter); the compressed bytes depend on the deflate implementation and are not comparable with the
// reference's, the inflated stream is.  Two-call protocol: out == NULL returns an upper bound of the size in *n_out.
namespace {
struct ZlibApi {
    int (*deflateInit2_)(z_streamp, int, int, int, int, int, const char *, int) = nullptr;
    int (*deflate)(z_streamp, int) = nullptr;
    int (*deflateReset)(z_streamp) = nullptr;
    int (*deflateEnd)(z_streamp) = nullptr;
    bool ok = false;
};
const ZlibApi &zlib_api() {
    static const ZlibApi api = [] {
        ZlibApi a;
        void *h = dlopen("libz.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libz.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.deflateInit2_ = reinterpret_cast<decltype(a.deflateInit2_)>(dlsym(h, "deflateInit2_"));
        a.deflate = reinterpret_cast<decltype(a.deflate)>(dlsym(h, "deflate"));
        a.deflateReset = reinterpret_cast<decltype(a.deflateReset)>(dlsym(h, "deflateReset"));
        a.deflateEnd = reinterpret_cast<decltype(a.deflateEnd)>(dlsym(h, "deflateEnd"));
        a.ok = a.deflateInit2_ && a.deflate && a.deflateReset && a.deflateEnd;
        return a;
    }();
    return api;
}
}  // namespace

extern "C" int smi_bgzf_deflate(const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *n_out, int level,
                                int block_bytes, int n_threads) {
    if ((!in && n_in) || !n_out || block_bytes < 1 || block_bytes > 0xFF00 || level < 0 || level > 9) {
        set_error("smi_bgzf_deflate: bad argument");
        return SMI_ERR_INVALID;
    }
    const ZlibApi &Z = zlib_api();
    if (out && !Z.ok) {
        set_error("smi_bgzf_deflate: libz.so.1 not found (this host-side writer is the only entry point that uses zlib; smi_bgzf_deflate_device needs none)");
        return SMI_ERR_STATE;
    }
    const size_t n_blocks = (n_in + (size_t)block_bytes - 1) / (size_t)block_bytes;
    const size_t slot = (size_t)block_bytes + 26 + 1024;  // deflate never grows a block of <= 0xFF00 bytes past this
    if (!out) {
        *n_out = n_blocks * slot + 28;
        return SMI_OK;
    }
    std::vector<std::vector<uint8_t>> blk(n_blocks);
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&]() {
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (Z.deflateInit2_(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY, ZLIB_VERSION, (int)sizeof(z_stream)) != Z_OK) {
            bad = 1;
            return;
        }
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= n_blocks) break;
            const size_t o = k * (size_t)block_bytes, n = std::min<size_t>((size_t)block_bytes, n_in - o);
            std::vector<uint8_t> &b = blk[k];
            b.resize(slot);
            Z.deflateReset(&zs);
            zs.next_in = const_cast<Bytef *>(in + o);
            zs.avail_in = (uInt)n;
            zs.next_out = b.data() + 18;
            zs.avail_out = (uInt)(slot - 26);
            if (Z.deflate(&zs, Z_FINISH) != Z_STREAM_END || 18 + zs.total_out + 8 > 0x10000) {
                bad = 1;
                break;
            }
            const uint32_t bsize = (uint32_t)(18 + zs.total_out + 8);
            const uint8_t head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)((bsize - 1) & 0xFF),
                                      (uint8_t)((bsize - 1) >> 8)};
            std::memcpy(b.data(), head, 18);
            const uint32_t crc = host_crc32(0, in + o, n);
            uint8_t *t = b.data() + 18 + zs.total_out;
            for (int i = 0; i < 4; i++) t[i] = (uint8_t)(crc >> (8 * i));
            for (int i = 0; i < 4; i++) t[4 + i] = (uint8_t)((uint32_t)n >> (8 * i));
            b.resize(bsize);
        }
        Z.deflateEnd(&zs);
    };
    const int nt = std::max(1, std::min<int>(n_threads, (int)std::max<size_t>(n_blocks, 1)));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (bad.load()) {
        set_error("smi_bgzf_deflate: deflate failed");
        return SMI_ERR_INVALID;
    }
    static const uint8_t kEof[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    size_t total = 28;
    for (const auto &b : blk) total += b.size();
    if (total > cap_out) {
        set_error("smi_bgzf_deflate: output buffer too small");
        return SMI_ERR_INVALID;
    }
    size_t o = 0;
    for (const auto &b : blk) {
        std::memcpy(out + o, b.data(), b.size());
        o += b.size();
    }
    std::memcpy(out + o, kEof, 28);
    *n_out = total;
    return SMI_OK;
}
