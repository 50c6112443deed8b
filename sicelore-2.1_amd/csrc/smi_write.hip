// smi_write.hip -- K-WRITE: the FASTQ records of pass 2, assembled on the device (gfx950).
//
// Reference units (bytecode, see DESIGN.md for the citation form):
//   FastqRecordExt.getRecordForWriting            FJ!nanoporereadscanner/readerwriter/FastqRecordExt.java:L209-311
//   FastqRecordExt.getStrandedSeq / lambdas       ...FastqRecordExt.java:L62-70,L115-122 (reverse complement, reversed qualities)
//   FastqWriterThreadPool$FastQoneFileThread.run  ...FastqWriterThreadPool.java:L300-306 (passed / failed stream, read ids)
//   ChimeraFindernew fragment names               FJ!nanoporereadscanner/analyzers/ChimeraFindernew.java:L309,L323
//   htsjdk BasicFastqWriter.write                 '@' name LF bases LF '+' qualityHeader LF qualities LF ("null" for a null string)
//
// MI355X mapping.  Byte work bound by HBM: a record is read once (bases, qualities, the name line) and written once.
//   K-WLEN   (thread = record)   length of the name (the formatter of smi_name.h run against a counting sink), of the
//                                sequence / quality range that is written, stream (passed / failed)
//   3 scans  (hipcub)            byte offset of every record in its stream, ordinal among the passed records (read id)
//   K-WNAME  (thread = record)   fragment tag + suffix formatted straight into the output stream
//   K-WRITE  (wave = record)     the 64 lanes copy the name token, bases (reverse complement through an LDS table), '+'
//                                line and qualities: one load per byte from a selected address, aligned dword stores
#include <hipcub/hipcub.hpp>

#include "smi_internal.h"
#include "smi_name.h"

namespace smi {

constexpr int kSuffixCap = 1024;  // longest fragment tag + suffix accepted (a real one is < 250 bytes)

struct WriteArgs {
    const uint8_t *text;
    const uint64_t *line_start;
    const uint8_t *reads, *quals;
    const uint64_t *bstart, *qstart;  // non-null: record i's bases / qualities begin at text[bstart[i]] / text[qstart[i]] (reads, quals unused)
    const uint64_t *offsets;
    const uint32_t *frag_src;
    const smi_chimera_result *chim;
    const smi_scan_result *scan;
    const smi_bc_result *bc;
    const int32_t *rank;
    size_t n;
    uint32_t first_read_id;
    int five_prime, trim;
};

// what a record consists of, derived the same way by K-WLEN and K-WRITE
struct RecPlan {
    size_t src;       // input record the name and '+' line come from
    int frag;         // fragment number, -1: not split
    uint64_t name_beg, name_tok_len;  // readName.split(" ")[0] in the text
    bool name_had_blank;
    uint64_t qh_beg, qh_len;          // text behind '+'
    uint64_t base;    // offset of the (fragment's) raw read in reads / quals
    int len;          // its length
    bool passed, rev, forced_failed;
    int cut_beg, cut_len;  // range of the stranded (passed) or raw (failed) sequence that is written, 0-based
};

__device__ __forceinline__ uint64_t line_end(const WriteArgs &A, uint64_t L) {
    uint64_t e = A.line_start[L + 1] - 1;
    if (e > A.line_start[L] && A.text[e - 1] == '\r') e--;
    return e;
}

__device__ __forceinline__ RecPlan plan_record(const WriteArgs &A, size_t i) {
    RecPlan R;
    const uint32_t fs = A.frag_src ? A.frag_src[i] : (uint32_t)(i << 2);
    R.src = A.frag_src ? (size_t)(fs >> 2) : i;
    const smi_chimera_result *ch = A.chim ? A.chim + R.src : nullptr;
    R.frag = (ch && ch->n_split) ? (int)(fs & 3u) : -1;
    R.forced_failed = ch && (ch->flags & SMI_CHIM_MULTI);  // MULTI_CHIMERIC_READS_DISCARDED | FAILED: never scanned (Parser.java:L92)
    const uint64_t l0 = A.line_start[4 * R.src], e0 = line_end(A, 4 * R.src);
    R.name_beg = l0 + 1;
    uint64_t t = R.name_beg;
    while (t < e0 && A.text[t] != ' ') t++;
    R.name_tok_len = t - R.name_beg;
    R.name_had_blank = t < e0;
    const uint64_t l2 = A.line_start[4 * R.src + 2];
    R.qh_beg = l2 + 1;
    R.qh_len = line_end(A, 4 * R.src + 2) - R.qh_beg;
    R.base = A.offsets[i];
    R.len = (int)(A.offsets[i + 1] - A.offsets[i]);
    const smi_scan_result &sc = A.scan[i];
    R.passed = !R.forced_failed && (sc.flags & (SMI_F_PASSED_FWD | SMI_F_PASSED_REV));
    R.rev = R.passed && (sc.flags & SMI_F_PASSED_REV);
    R.cut_beg = 0;
    R.cut_len = R.len;
    const smi_bc_result &b = A.bc[i];
    if (R.passed && A.trim && b.found == 1) {
        // partOfSeqToWrite (L210-217): from the TSO end (5': 30 bases behind the barcode start) to the polyA start
        const int bc_start = sc.adapter_end + 1 + b.offset;
        const int beg = A.five_prime ? bc_start + 30 : (sc.tso_end != 0 ? sc.tso_end : 1);
        const int end = sc.polya_end != 0 ? sc.polya_start : R.len;
        if (beg < end) {  // substring(beg - 1, end) (L303-304); a range outside the read throws in the reference
            R.cut_beg = min(max(beg - 1, 0), R.len);
            R.cut_len = max(min(end, R.len) - R.cut_beg, 0);
        }
    }
    return R;
}

__device__ __forceinline__ const char *split_tag(int reason) {  // ChimeraFindernew$SplitPosition$SplitReason tags
    switch (reason) {
    case SMI_SPLIT_FWD_ADAPTER: return "FA";
    case SMI_SPLIT_RA_FA: return "RA_FA";
    case SMI_SPLIT_RA_FT: return "RA_FT";
    case SMI_SPLIT_RT_FA: return "RT_FA";
    case SMI_SPLIT_RT_FT: return "RT_FT";
    default: return "RA";
    }
}

// what follows the name token: fragment tag + suffix; returns the status of append_name_suffix
__device__ __forceinline__ int format_record_suffix(const WriteArgs &A, const RecPlan &R, size_t i, uint32_t read_id, NameSink &s,
                                                    bool *quals_set) {
    if (R.frag >= 0 && R.name_had_blank) {
        // readName.replaceFirst(" ", "_" + tag + "sp" + (k + 1) + " "): fragments before a cut carry that cut's tag, the
        // last fragment the tag of the cut it starts at
        const smi_chimera_result &ch = A.chim[R.src];
        const int cut = R.frag < ch.n_split ? R.frag : ch.n_split - 1;
        s.put('_');
        s.puts(split_tag(ch.reason[cut]));
        s.puts("sp");
        s.put_int(R.frag + 1);
    }
    if (R.forced_failed) {
        s.puts("_FAILED ");
        *quals_set = true;
        return NAME_OK;
    }
    const uint8_t *rd = A.bstart ? A.text + A.bstart[i] : A.reads + R.base, *ql = A.bstart ? A.text + A.qstart[i] : A.quals + R.base;
    const smi_bc_result *b = A.bc[i].found == 1 ? A.bc + i : nullptr;
    return append_name_suffix(
        s, A.scan[i], b, A.rank ? A.rank[i] : 0, read_id, A.five_prime != 0, R.len, [&](int k) { return (char)rd[k]; },
        [&](int k) { return (char)ql[k]; }, quals_set);
}

// '@' name LF bases LF '+' header LF qualities LF
__device__ __forceinline__ uint64_t record_bytes(const RecPlan &R, int name_len, bool quals_set) {
    return 1ull + name_len + 1 + R.cut_len + 1 + 1 + R.qh_len + 1 + (quals_set ? (uint64_t)R.cut_len : 4ull) + 1;
}

__global__ void k_write_len(WriteArgs A, uint64_t *__restrict__ len_passed, uint64_t *__restrict__ len_failed,
                            uint64_t *__restrict__ cnt_passed, uint32_t *__restrict__ sfx_len, uint8_t *__restrict__ is_passed,
                            uint32_t *__restrict__ err) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    const RecPlan R = plan_record(A, i);
    NameSink s{nullptr, 0, 0};  // counts only
    bool quals_set = true;
    // the base-36 id changes the length, and it is only known after the scan of the passed flags: K-WLEN therefore
    // measures the suffix with id 0 and k_write_idlen adds the missing digits
    const int st = format_record_suffix(A, R, i, 0u, s, &quals_set);
    if (st == NAME_RANGE) atomicOr(err, SMI_WR_NAME_RANGE);
    if (s.n + 8 > kSuffixCap) atomicOr(err, SMI_WR_NAME_TOO_LONG);
    const uint64_t bytes = record_bytes(R, (int)R.name_tok_len + s.n, quals_set);
    len_passed[i] = R.passed ? bytes : 0;
    len_failed[i] = R.passed ? 0 : bytes;
    cnt_passed[i] = R.passed ? 1 : 0;
    sfx_len[i] = (uint32_t)s.n | (quals_set ? 0x80000000u : 0u);
    is_passed[i] = R.passed ? 1 : 0;
}

// width of Integer.toString(id, 36) minus the one digit K-WLEN counted for id 0
__device__ __forceinline__ int base36_extra(uint32_t v) {
    int k = 0;
    while (v >= 36u) {
        v /= 36u;
        k++;
    }
    return k;
}

// second step of the length computation: the read id of a passed record = first id + its ordinal among the passed ones
__global__ void k_write_idlen(const WriteArgs A, const uint64_t *__restrict__ ord_passed, uint64_t *__restrict__ len_passed,
                              uint32_t *__restrict__ sfx_len) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= A.n || len_passed[i] == 0) return;
    const smi_scan_result &sc = A.scan[i];
    // only names that carry the suffix carry the id (append_name_suffix)
    const int begin = A.five_prime ? sc.adapter_end - 3 : sc.adapter_end - 41;
    if (!sc.found || begin < 0) return;
    const int extra = base36_extra(A.first_read_id + (uint32_t)ord_passed[i]);
    len_passed[i] += (uint64_t)extra;
    sfx_len[i] += (uint32_t)extra;
}

// K-WNAME (thread = record): the suffix goes straight to its place in the output stream -- serial per record, parallel
// over records (inside K-WRITE the other 63 lanes of the record's wave would wait for it)
constexpr int kNameBlock = 128;   // threads (= records) per block of K-WNAME
constexpr int kNameStage = 260;   // bytes of LDS per record: 65 words, so equal byte offsets of neighbouring records fall into different banks
__global__ __launch_bounds__(kNameBlock) void k_write_name(WriteArgs A, const uint64_t *__restrict__ off_passed,
                                                           const uint64_t *__restrict__ off_failed, const uint64_t *__restrict__ ord_passed,
                                                           const uint32_t *__restrict__ sfx_len, uint8_t *__restrict__ out_passed,
                                                           size_t cap_passed, uint8_t *__restrict__ out_failed, size_t cap_failed) {
    // The formatter is serial per record and emits single bytes: into global memory that is one 1-byte transaction per lane and character.
    // Every thread therefore formats into its own row of LDS, then each wave copies the rows of its 64 records out, a row at a time with
    // lane = byte, so that a store instruction covers 64 consecutive bytes of the stream.
    __shared__ char stage[kNameBlock][kNameStage];
    __shared__ uint64_t dst[kNameBlock];
    __shared__ int cnt[kNameBlock];
    const int t = threadIdx.x;
    const size_t i = blockIdx.x * (size_t)kNameBlock + t;
    int n_copy = 0;
    uint64_t d = 0;
    if (i < A.n) {
        const RecPlan R = plan_record(A, i);
        const int n_sfx = (int)(sfx_len[i] & 0x7FFFFFFFu);
        const uint64_t pos = (R.passed ? off_passed[i] : off_failed[i]) + 1 + R.name_tok_len;
        if (pos + n_sfx <= (R.passed ? cap_passed : cap_failed)) {  // otherwise K-WRITE reports the overflow
            char *out = reinterpret_cast<char *>(R.passed ? out_passed : out_failed) + pos;
            const bool staged = n_sfx <= kNameStage;
            NameSink s{staged ? stage[t] : out, 0, n_sfx};
            bool quals_set = true;
            format_record_suffix(A, R, i, A.first_read_id + (uint32_t)ord_passed[i], s, &quals_set);
            if (staged) {
                n_copy = n_sfx;
                d = (uint64_t)(uintptr_t)out;
            }
        }
    }
    cnt[t] = n_copy;
    dst[t] = d;
    __syncthreads();
    const int lane = t & 63, w0 = t & ~63;
    for (int r = w0; r < w0 + 64; r++) {
        const int n = cnt[r];
        char *o = reinterpret_cast<char *>((uintptr_t)dst[r]);
        for (int k = lane; k < n; k += 64) o[k] = stage[r][k];
    }
}

// K-WRITE: one wave per record copies everything but the suffix.  Every output byte is ONE load from a computed address
// (text, reads, qualities or the literal table in LDS, through generic pointers) chosen with selects, so the four loads
// of a lane are in flight together; four bytes per lane go out as one aligned dword.
__global__ __launch_bounds__(256) void k_write(WriteArgs A, const uint64_t *__restrict__ off_passed,
                                               const uint64_t *__restrict__ off_failed, const uint32_t *__restrict__ sfx_len,
                                               uint8_t *__restrict__ out_passed, size_t cap_passed, uint8_t *__restrict__ out_failed,
                                               size_t cap_failed, uint64_t *__restrict__ rec_off, uint32_t *__restrict__ err) {
    __shared__ char rc_lut[512];  // [0..255] identity, [256..511] FastqRecordExt.REVERSE_COMPLEMENT
    __shared__ char literals[8];  // "@\n+null"
    rc_lut[threadIdx.x] = (char)threadIdx.x;
    rc_lut[256 + threadIdx.x] = rc_char((unsigned char)threadIdx.x);
    if (threadIdx.x < 7) literals[threadIdx.x] = "@\n+null"[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const size_t i = blockIdx.x * (size_t)4 + (threadIdx.x >> 6);
    if (i >= A.n) return;
    // ---- plan (wave-uniform values; the token end is searched 64 characters at a time) ---------------------------------
    const uint32_t fs = A.frag_src ? A.frag_src[i] : 0u;
    const size_t src = A.frag_src ? (size_t)(fs >> 2) : i;
    const bool forced_failed = A.chim && (A.chim[src].flags & SMI_CHIM_MULTI);
    const uint64_t name_beg = A.line_start[4 * src] + 1, e0 = line_end(A, 4 * src);
    uint64_t t = name_beg;
    for (;; t += 64) {
        const bool stop = t + lane >= e0 || A.text[t + lane] == ' ';
        const unsigned long long m = __ballot(stop);
        if (m) {
            t += __builtin_ctzll(m);
            break;
        }
    }
    const uint64_t tok_len = t - name_beg;
    const uint64_t qh_beg = A.line_start[4 * src + 2] + 1, qh_len = line_end(A, 4 * src + 2) - qh_beg;
    const uint64_t base = A.offsets[i];
    const int len = (int)(A.offsets[i + 1] - base);
    const smi_scan_result &sc = A.scan[i];
    const bool passed = !forced_failed && (sc.flags & (SMI_F_PASSED_FWD | SMI_F_PASSED_REV));
    const bool rev = passed && (sc.flags & SMI_F_PASSED_REV);
    int cut_beg = 0, cut_len = len;
    const smi_bc_result &b = A.bc[i];
    if (passed && A.trim && b.found == 1) {  // same rule as plan_record
        const int bc_start = sc.adapter_end + 1 + b.offset;
        const int beg = A.five_prime ? bc_start + 30 : (sc.tso_end != 0 ? sc.tso_end : 1);
        const int end = sc.polya_end != 0 ? sc.polya_start : len;
        if (beg < end) {
            cut_beg = min(max(beg - 1, 0), len);
            cut_len = max(min(end, len) - cut_beg, 0);
        }
    }
    const uint32_t sl = sfx_len[i];
    const bool qset = sl >> 31;
    const uint64_t n_sfx = sl & 0x7FFFFFFFu;
    const uint64_t off = passed ? off_passed[i] : off_failed[i];
    const uint64_t qlen = qset ? (uint64_t)cut_len : 4ull;
    // '@' token suffix LF bases LF '+' header LF qualities LF
    const uint64_t p_sfx = 1ull + tok_len, p_seq = p_sfx + n_sfx + 1, p_plus = p_seq + cut_len + 1, p_qh = p_plus + 1,
                   p_q = p_qh + qh_len + 1, bytes = p_q + qlen + 1;
    if (lane == 0) rec_off[i] = off;
    uint8_t *out = passed ? out_passed : out_failed;
    if (off + bytes > (passed ? cap_passed : cap_failed)) {
        if (lane == 0) atomicOr(err, SMI_WR_OVERFLOW);
        return;
    }
    const uint8_t *rd = A.bstart ? A.text + A.bstart[i] : A.reads + base, *ql = A.bstart ? A.text + A.qstart[i] : A.quals + base;
    const uint8_t *tok = A.text + name_beg, *qh = A.text + qh_beg;
    const uint8_t *lit = reinterpret_cast<const uint8_t *>(literals);
    const uint8_t *lut = reinterpret_cast<const uint8_t *>(rc_lut);
    // bases / qualities: position k of the written range is raw index cut_beg + k, or len - 1 - cut_beg - k when reversed
    const int64_t step = rev ? -1 : 1;
    const uint8_t *seq0 = rd + (rev ? len - 1 - cut_beg : cut_beg);
    const uint8_t *qual0 = qset ? ql + (rev ? len - 1 - cut_beg : cut_beg) : lit + 3;
    const int64_t qstep = qset ? step : 1;
    const uint32_t rc_sel = rev ? 256u : 0u;
    auto source = [&](uint64_t j, uint32_t &table) -> const uint8_t * {
        const uint8_t *p = lit + 1;  // LF: the byte that closes each of the four lines
        p = j == 0 ? lit : p;
        p = (j >= 1 && j < p_sfx) ? tok + (j - 1) : p;
        const bool in_seq = j >= p_seq && j + 1 < p_plus;
        p = in_seq ? seq0 + step * (int64_t)(j - p_seq) : p;
        table = in_seq ? rc_sel : 0u;
        p = j == p_plus ? lit + 2 : p;
        p = (j >= p_qh && j + 1 < p_q) ? qh + (j - p_qh) : p;
        p = (j >= p_q && j - p_q < qlen) ? qual0 + qstep * (int64_t)(j - p_q) : p;
        return p;
    };
    // dword-aligned output positions g = (off & ~3) + 4 * (64 * it + lane); dwords that are not wholly this wave's (the two
    // ends of the record, shared with its neighbours, and the ends of the suffix K-WNAME writes) go out as bytes
    const uint64_t g0 = off & ~3ull, end = off + bytes, s0 = off + p_sfx, s1 = s0 + n_sfx;
    // the long runs -- four bytes that all lie in the bases or in the qualities -- take one (unaligned) dword load per lane
    // when that holds for the whole wave; everything else goes through `source`
    const uint64_t seq_lo = off + p_seq, seq_hi = off + p_plus - 1, q_lo = off + p_q, q_hi = q_lo + (qset ? qlen : 0);
    for (uint64_t g = g0 + 4ull * lane; g < end; g += 256) {
        const bool all_seq = g >= seq_lo && g + 4 <= seq_hi, all_q = g >= q_lo && g + 4 <= q_hi;
        if (!__ballot(!(all_seq || all_q))) {
            const uint64_t k = all_seq ? g - seq_lo : g - q_lo;
            const uint8_t *b0 = all_seq ? seq0 : qual0;
            uint32_t w;
            if (rev) {
                __builtin_memcpy(&w, b0 - (int64_t)k - 3, 4);  // source bytes k+3 .. k of the mirrored run
                w = __builtin_bswap32(w);
                if (all_seq)
                    w = lut[256 + (w & 0xFF)] | (lut[256 + ((w >> 8) & 0xFF)] << 8) | (lut[256 + ((w >> 16) & 0xFF)] << 16) |
                        (lut[256 + (w >> 24)] << 24);
            } else
                __builtin_memcpy(&w, b0 + k, 4);
            *reinterpret_cast<uint32_t *>(out + g) = w;
            continue;
        }
        uint32_t c[4], tb[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t a = g + k;
            ok[k] = a >= off && a < end && !(a >= s0 && a < s1);
            const uint8_t *p = source(ok[k] ? a - off : 0, tb[k]);
            c[k] = *p;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) c[k] = lut[tb[k] + c[k]];
        if (ok[0] && ok[1] && ok[2] && ok[3])
            *reinterpret_cast<uint32_t *>(out + g) = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
        else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (ok[k]) out[g + k] = (uint8_t)c[k];
        }
    }
}

}  // namespace smi

using namespace smi;

static int write_core(smi_ctx *ctx, const uint8_t *d_text, const uint64_t *d_line_start, const uint8_t *d_reads, const uint8_t *d_quals,
                      const uint64_t *d_bstart, const uint64_t *d_qstart, const uint64_t *d_offsets, const uint32_t *d_frag_src,
                      const smi_chimera_result *d_chim, const smi_scan_result *d_scan, const smi_bc_result *d_bc, const int32_t *d_rank,
                      size_t n_out, uint32_t first_read_id, const smi_write_config *cfg, uint8_t *d_passed, size_t cap_passed,
                      uint8_t *d_failed, size_t cap_failed, uint64_t *d_rec_off, uint8_t *d_is_passed, uint64_t *totals, uint32_t *errors,
                      void *stream) {
    if (!ctx || !cfg || !totals || !errors) {
        set_error("smi_fastq_write_device: null argument");
        return SMI_ERR_INVALID;
    }
    totals[0] = totals[1] = totals[2] = 0;
    *errors = 0;
    if (n_out == 0) return SMI_OK;
    if (!d_text || !d_line_start || (d_bstart ? !d_qstart : (!d_reads || !d_quals)) || !d_offsets || !d_scan || !d_bc || !d_passed || !d_failed || !d_rec_off ||
        !d_is_passed || ((d_frag_src == nullptr) != (d_chim == nullptr))) {
        set_error("smi_fastq_write_device: null argument");
        return SMI_ERR_INVALID;
    }
    SMI_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scratch: 6 arrays of n_out + 1 u64 (3 lengths, 3 scanned), the error word, hipcub temp storage
    size_t tmp_bytes = 0;
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (int)(n_out + 1), s));
    const size_t arr = (n_out + 1) * sizeof(uint64_t);
    const size_t sfx_bytes = ((n_out * sizeof(uint32_t) + 255) / 256) * 256;
    const size_t need = 6 * arr + sfx_bytes + 256 + tmp_bytes;
    if (ctx->scan_tmp_bytes < need) {
        SMI_HIP(hipStreamSynchronize(s));
        if (ctx->scan_tmp) SMI_HIP(hipFree(ctx->scan_tmp));
        ctx->scan_tmp = nullptr;
        ctx->scan_tmp_bytes = 0;
        SMI_HIP(hipMalloc(&ctx->scan_tmp, need));
        ctx->scan_tmp_bytes = need;
    }
    uint8_t *base = static_cast<uint8_t *>(ctx->scan_tmp);
    uint64_t *lenp = (uint64_t *)base, *lenf = (uint64_t *)(base + arr), *cntp = (uint64_t *)(base + 2 * arr);
    uint64_t *offp = (uint64_t *)(base + 3 * arr), *offf = (uint64_t *)(base + 4 * arr), *ordp = (uint64_t *)(base + 5 * arr);
    uint32_t *sfx = (uint32_t *)(base + 6 * arr);
    uint32_t *d_err = (uint32_t *)(base + 6 * arr + sfx_bytes);
    void *cub_tmp = base + 6 * arr + sfx_bytes + 256;
    SMI_HIP(hipMemsetAsync(d_err, 0, 4, s));
    // the extra last element makes the exclusive scans deliver the totals
    SMI_HIP(hipMemsetAsync(lenp + n_out, 0, 8, s));
    SMI_HIP(hipMemsetAsync(lenf + n_out, 0, 8, s));
    SMI_HIP(hipMemsetAsync(cntp + n_out, 0, 8, s));
    WriteArgs A{d_text, d_line_start, d_reads, d_quals, d_bstart, d_qstart, d_offsets, d_frag_src, d_chim, d_scan, d_bc, d_rank,
                n_out,  first_read_id, cfg->five_prime, cfg->trim_fastq};
    const unsigned g = (unsigned)((n_out + 255) / 256);
    hipLaunchKernelGGL(k_write_len, dim3(g), dim3(256), 0, s, A, lenp, lenf, cntp, sfx, d_is_passed, d_err);
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(cub_tmp, tmp_bytes, cntp, ordp, (int)(n_out + 1), s));
    hipLaunchKernelGGL(k_write_idlen, dim3(g), dim3(256), 0, s, A, ordp, lenp, sfx);
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(cub_tmp, tmp_bytes, lenp, offp, (int)(n_out + 1), s));
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(cub_tmp, tmp_bytes, lenf, offf, (int)(n_out + 1), s));
    // K-WNAME (serial per record, latency-bound) and K-WRITE (wave per record, bandwidth-bound) write disjoint bytes and only read what is
    // finished by now: they run side by side, K-WNAME on the context's side stream
    if (!ctx->side_stream) {
        SMI_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        SMI_HIP(hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
        SMI_HIP(hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming));
    }
    SMI_HIP(hipEventRecord(ctx->side_fork, s));
    SMI_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
    hipLaunchKernelGGL(k_write_name, dim3((unsigned)((n_out + kNameBlock - 1) / kNameBlock)), dim3(kNameBlock), 0, ctx->side_stream, A, offp, offf, ordp, sfx, d_passed, cap_passed, d_failed,
                       cap_failed);
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipEventRecord(ctx->side_join, ctx->side_stream));
    hipLaunchKernelGGL(k_write, dim3((unsigned)((n_out + 3) / 4)), dim3(256), 0, s, A, offp, offf, sfx, d_passed, cap_passed,
                       d_failed, cap_failed, d_rec_off, d_err);
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipStreamWaitEvent(s, ctx->side_join, 0));
    uint64_t h[3];
    SMI_HIP(hipMemcpyAsync(&h[0], offp + n_out, 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(&h[1], offf + n_out, 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(&h[2], ordp + n_out, 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(errors, d_err, 4, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    totals[0] = h[0];
    totals[1] = h[1];
    totals[2] = h[2];
    if (*errors) {
        set_error("smi_fastq_write_device: see *errors (SMI_WR_*)");
        return SMI_ERR_INVALID;
    }
    return SMI_OK;
}

extern "C" int smi_fastq_write_device(smi_ctx *ctx, const uint8_t *d_text, const uint64_t *d_line_start, const uint8_t *d_reads,
                                      const uint8_t *d_quals, const uint64_t *d_offsets, const uint32_t *d_frag_src,
                                      const smi_chimera_result *d_chim, const smi_scan_result *d_scan, const smi_bc_result *d_bc,
                                      const int32_t *d_rank, size_t n_out, uint32_t first_read_id, const smi_write_config *cfg,
                                      uint8_t *d_passed, size_t cap_passed, uint8_t *d_failed, size_t cap_failed,
                                      uint64_t *d_rec_off, uint8_t *d_is_passed, uint64_t *totals, uint32_t *errors, void *stream) {
    if (!d_reads || !d_quals) {
        if (n_out) {
            set_error("smi_fastq_write_device: null argument");
            return SMI_ERR_INVALID;
        }
    }
    return write_core(ctx, d_text, d_line_start, d_reads, d_quals, nullptr, nullptr, d_offsets, d_frag_src, d_chim, d_scan, d_bc, d_rank, n_out,
                      first_read_id, cfg, d_passed, cap_passed, d_failed, cap_failed, d_rec_off, d_is_passed, totals, errors, stream);
}

// the same writer taking bases and qualities where the FASTQ text has them: d_base_start / d_qual_start from smi_frag_text_starts_device
extern "C" int smi_fastq_write_text_device(smi_ctx *ctx, const uint8_t *d_text, const uint64_t *d_line_start, const uint64_t *d_base_start,
                                           const uint64_t *d_qual_start, const uint64_t *d_offsets, const uint32_t *d_frag_src,
                                           const smi_chimera_result *d_chim, const smi_scan_result *d_scan, const smi_bc_result *d_bc,
                                           const int32_t *d_rank, size_t n_out, uint32_t first_read_id, const smi_write_config *cfg,
                                           uint8_t *d_passed, size_t cap_passed, uint8_t *d_failed, size_t cap_failed,
                                           uint64_t *d_rec_off, uint8_t *d_is_passed, uint64_t *totals, uint32_t *errors, void *stream) {
    if (n_out && (!d_base_start || !d_qual_start)) {
        set_error("smi_fastq_write_text_device: null argument");
        return SMI_ERR_INVALID;
    }
    return write_core(ctx, d_text, d_line_start, nullptr, nullptr, d_base_start, d_qual_start, d_offsets, d_frag_src, d_chim, d_scan, d_bc,
                      d_rank, n_out, first_read_id, cfg, d_passed, cap_passed, d_failed, cap_failed, d_rec_off, d_is_passed, totals, errors,
                      stream);
}
