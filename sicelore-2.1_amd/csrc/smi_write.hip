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

// What a record consists of: derived once, by K-WLEN, and handed to K-WNAME and K-WRITE as one 64-byte row (a wave of K-WRITE used to
// walk frag_src -> line table -> name line -> offsets / scan / bc itself: five dependent round trips before its first copy).
struct __attribute__((aligned(16))) RecPlan {
    uint64_t name_beg;   // readName.split(" ")[0] in the text
    uint64_t qh_beg;     // text behind '+'
    uint64_t rd, ql;     // the (fragment's) raw bases / qualities: positions in `text` (bstart given) or in reads / quals
    uint32_t name_tok_len, qh_len;
    int32_t len;         // raw length
    int32_t cut_beg, cut_len;  // range of the stranded (passed) or raw (failed) sequence that is written, 0-based
    uint32_t src;        // input record the name and '+' line come from
    int32_t frag;        // fragment number, -1: not split
    uint32_t flags;      // kPassed | kRev | kForcedFailed | kHadBlank
};
static_assert(sizeof(RecPlan) == 64, "one plan row = four 16-byte loads");
enum : uint32_t { kPassed = 1u, kRev = 2u, kForcedFailed = 4u, kHadBlank = 8u };

__device__ __forceinline__ uint64_t line_end(const WriteArgs &A, uint64_t L) {
    uint64_t e = A.line_start[L + 1] - 1;
    if (e > A.line_start[L] && A.text[e - 1] == '\r') e--;
    return e;
}

// first blank of text[beg, end) (or end): 16 characters per step while the loads stay in front of `safe` (a position inside the same
// record), SWAR zero-byte test on x ^ "    "
__device__ __forceinline__ uint64_t find_blank(const uint8_t *text, uint64_t beg, uint64_t end, uint64_t safe) {
    uint64_t t = beg;
    while (t < end && t + 16 <= safe) {
        uint32_t w[4];
        __builtin_memcpy(w, text + t, 16);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t x = w[k] ^ 0x20202020u;
            const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;
            if (z) {
                const uint64_t hit = t + 4 * k + (__builtin_ctz(z) >> 3);
                return hit < end ? hit : end;
            }
        }
        t += 16;
    }
    while (t < end && text[t] != ' ') t++;
    return t < end ? t : end;
}

__device__ __forceinline__ RecPlan plan_record(const WriteArgs &A, size_t i) {
    RecPlan R;
    const uint32_t fs = A.frag_src ? A.frag_src[i] : (uint32_t)(i << 2);
    R.src = A.frag_src ? (fs >> 2) : (uint32_t)i;
    const smi_chimera_result *ch = A.chim ? A.chim + R.src : nullptr;
    R.frag = (ch && ch->n_split) ? (int)(fs & 3u) : -1;
    const bool forced_failed = ch && (ch->flags & SMI_CHIM_MULTI);  // MULTI_CHIMERIC_READS_DISCARDED | FAILED: never scanned (Parser.java:L92)
    const uint64_t l0 = A.line_start[4 * (size_t)R.src], e0 = line_end(A, 4 * (size_t)R.src);
    const uint64_t l2 = A.line_start[4 * (size_t)R.src + 2];
    R.name_beg = l0 + 1;
    const uint64_t t = find_blank(A.text, R.name_beg, e0, l2);
    R.name_tok_len = (uint32_t)(t - R.name_beg);
    R.qh_beg = l2 + 1;
    R.qh_len = (uint32_t)(line_end(A, 4 * (size_t)R.src + 2) - R.qh_beg);
    const uint64_t base = A.offsets[i];
    R.rd = A.bstart ? A.bstart[i] : base;
    R.ql = A.bstart ? A.qstart[i] : base;
    R.len = (int)(A.offsets[i + 1] - base);
    const smi_scan_result &sc = A.scan[i];
    const bool passed = !forced_failed && (sc.flags & (SMI_F_PASSED_FWD | SMI_F_PASSED_REV));
    R.flags = (passed ? kPassed : 0u) | ((passed && (sc.flags & SMI_F_PASSED_REV)) ? kRev : 0u) | (forced_failed ? kForcedFailed : 0u) |
              (t < e0 ? kHadBlank : 0u);
    R.cut_beg = 0;
    R.cut_len = R.len;
    const smi_bc_result &b = A.bc[i];
    if (passed && A.trim && b.found == 1) {
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

// The characters behind X= and Q= (name_window of smi_name.h) in registers, in stranded order: a few wide loads that are in flight
// together, instead of one dependent byte load per character from inside the formatter.
struct WindowRegs {
    uint32_t w[kNameWindowMax / 4];
    __device__ __forceinline__ uint8_t at(int k) const { return (uint8_t)(w[k >> 2] >> (8 * (k & 3))); }  // k is a constant after unrolling
};
__device__ __forceinline__ WindowRegs load_window(const uint8_t *raw, const NameWindow &nw) {
    constexpr int ND = kNameWindowMax / 4;
    uint32_t v[ND];
    const uint8_t *p = raw + nw.lo;
#pragma unroll
    for (int d = 0; d < ND; d++) {
        v[d] = 0;
        if (4 * d + 4 <= nw.n_chars)
            __builtin_memcpy(&v[d], p + 4 * d, 4);
        else
            for (int k = 4 * d; k < nw.n_chars; k++) v[d] |= (uint32_t)p[k] << (8 * (k - 4 * d));
    }
    WindowRegs r;
    if (!nw.rev) {
#pragma unroll
        for (int d = 0; d < ND; d++) r.w[d] = v[d];
        return r;
    }
    // stranded character k = raw character n_chars - 1 - k: reverse all 44 bytes, then drop the 44 - n_chars bytes that came to the front
    uint32_t t[ND + 1];
#pragma unroll
    for (int d = 0; d < ND; d++) t[d] = __builtin_bswap32(v[ND - 1 - d]);
    t[ND] = 0;
    const int drop = kNameWindowMax - nw.n_chars;  // 0 (3') or 1 (5')
#pragma unroll
    for (int d = 0; d < ND; d++) r.w[d] = drop == 0 ? t[d] : (drop == 1 ? __builtin_amdgcn_alignbyte(t[d + 1], t[d], 1) : 0u);
    return r;
}

// reverse complement of four bases at once.  (c >> 1) & 7 separates A C G T N (0 1 3 2 7), so one v_perm_b32 against an 8-byte table
// complements a dword and a second one against the identity table proves that all four characters were one of the five; anything
// else (lower case, IUPAC codes, which FastqRecordExt.REVERSE_COMPLEMENT also maps, and the characters it maps to 0) takes the table in LDS.
__device__ __forceinline__ bool rc4(uint32_t w, uint32_t &out) {
    const uint32_t idx = (w >> 1) & 0x07070707u;
    // byte j of {hi:lo} = the character with index j:   0 'A'  1 'C'  2 'T'  3 'G'  4 -  5 -  6 -  7 'N'
    const uint32_t id_lo = 'A' | ('C' << 8) | ('T' << 16) | ((uint32_t)'G' << 24), id_hi = (uint32_t)'N' << 24;
    const uint32_t rc_lo = 'T' | ('G' << 8) | ('A' << 16) | ((uint32_t)'C' << 24), rc_hi = (uint32_t)'N' << 24;
    out = __builtin_amdgcn_perm(rc_hi, rc_lo, idx);
    return __builtin_amdgcn_perm(id_hi, id_lo, idx) == w;
}

// the complement of a dword that holds something else than A C G T N (or pad bytes): FastqRecordExt.REVERSE_COMPLEMENT's switch per byte, kept out
// of line -- inlined per character of the X= window (43 times 30 cases) it was a good part of K-WNAME's 44,000 instructions
__device__ __noinline__ uint32_t rc_dword_switch(uint32_t w) {
    return (uint32_t)(uint8_t)rc_char((unsigned char)(w & 0xFF)) | ((uint32_t)(uint8_t)rc_char((unsigned char)((w >> 8) & 0xFF)) << 8) |
           ((uint32_t)(uint8_t)rc_char((unsigned char)((w >> 16) & 0xFF)) << 16) | ((uint32_t)(uint8_t)rc_char((unsigned char)(w >> 24)) << 24);
}

// The X= / Q= windows as the formatter's bulk hooks (smi_name.h has_bulk): the window sits in eleven registers in stranded order; X= is the
// window without its first character, complemented -- a dword at a time -- for a read that passed on the reverse strand; Q= sums all of it.
struct DevSeqWindow {
    static constexpr bool kBulk = true;
    WindowRegs w;
    __device__ __forceinline__ char operator()(int) const { return 0; }  // (never used: bulk_x below)
    template <class Sink>
    __device__ __forceinline__ void bulk_x(Sink &s, const NameWindow &nw) const {
        constexpr int ND = kNameWindowMax / 4;
        uint32_t c[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) {
            c[d] = w.w[d];
            if (nw.rev) {
                uint32_t o;
                c[d] = (4 * d + 4 <= nw.n_chars && rc4(c[d], o)) ? o : rc_dword_switch(c[d]);
            }
        }
#pragma unroll
        for (int k = 1; k < kNameWindowMax; k++)
            if (k < nw.n_chars) s.put((char)(uint8_t)(c[k >> 2] >> (8 * (k & 3))));
    }
};
struct DevQualWindow {
    static constexpr bool kBulk = true;
    WindowRegs w;
    __device__ __forceinline__ char operator()(int) const { return 0; }
    __device__ __forceinline__ int bulk_sum(const NameWindow &nw) const {  // (bytes behind the window are 0)
        uint32_t acc = 0;
#pragma unroll
        for (int d = 0; d < kNameWindowMax / 4; d++) acc = __builtin_amdgcn_sad_u8(w.w[d], 0u, acc);
        return (int)acc - 33 * nw.n_chars;
    }
};

// what follows the name token: fragment tag + suffix; returns the status of append_name_suffix
__device__ __forceinline__ int format_record_suffix(const WriteArgs &A, const RecPlan &R, size_t i, uint32_t read_id, NameSink &s,
                                                    bool *quals_set) {
    if (R.frag >= 0 && (R.flags & kHadBlank)) {
        // readName.replaceFirst(" ", "_" + tag + "sp" + (k + 1) + " "): fragments before a cut carry that cut's tag, the
        // last fragment the tag of the cut it starts at
        const smi_chimera_result &ch = A.chim[R.src];
        const int cut = R.frag < ch.n_split ? R.frag : ch.n_split - 1;
        s.put('_');
        s.puts(split_tag(cut == 0 ? ch.reason[0] : ch.reason[1]));  // (an indexed member of a struct held in registers would go through scratch)
        s.puts("sp");
        s.put_i32(R.frag + 1);
    }
    if (R.flags & kForcedFailed) {
        s.puts("_FAILED ");
        *quals_set = true;
        return NAME_OK;
    }
    const smi_scan_result sc = A.scan[i];
    const smi_bc_result bcr = A.bc[i];
    const NameWindow nw = name_window(sc, A.five_prime != 0, R.len);
    DevSeqWindow ws;
    DevQualWindow wq;
#pragma unroll
    for (int d = 0; d < kNameWindowMax / 4; d++) ws.w.w[d] = wq.w.w[d] = 0;
    if (nw.has && nw.n_chars >= kNameWindowMax - 1) {  // the two shipped layouts: 44 (3') and 43 (5') characters
        ws.w = load_window((A.bstart ? A.text : A.reads) + R.rd, nw);
        wq.w = load_window((A.bstart ? A.text : A.quals) + R.ql, nw);
    }
    return append_name_suffix(s, sc, &bcr /* has_bc = found == 1, tested inside */, A.rank ? A.rank[i] : 0, read_id, A.five_prime != 0, R.len, ws, wq, quals_set);
}

// '@' name LF bases LF '+' header LF qualities LF
__device__ __forceinline__ uint64_t record_bytes(const RecPlan &R, int name_len, bool quals_set) {
    return 1ull + name_len + 1 + R.cut_len + 1 + 1 + R.qh_len + 1 + (quals_set ? (uint64_t)R.cut_len : 4ull) + 1;
}

__global__ void k_write_len(WriteArgs A, RecPlan *__restrict__ plan, uint64_t *__restrict__ len_passed, uint64_t *__restrict__ len_failed,
                            uint64_t *__restrict__ cnt_passed, uint32_t *__restrict__ sfx_len, uint8_t *__restrict__ is_passed,
                            uint32_t *__restrict__ err) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    const RecPlan R = plan_record(A, i);
    plan[i] = R;
    const bool passed = R.flags & kPassed;
    NameSink s{nullptr, 0, 0};  // counts only
    bool quals_set = true;
    // the base-36 id changes the length, and it is only known after the scan of the passed flags: K-WLEN therefore
    // measures the suffix with id 0 and k_write_idlen adds the missing digits
    const int st = format_record_suffix(A, R, i, 0u, s, &quals_set);
    if (st == NAME_RANGE) atomicOr(err, SMI_WR_NAME_RANGE);
    if (s.n + 8 > kSuffixCap) atomicOr(err, SMI_WR_NAME_TOO_LONG);
    const uint64_t bytes = record_bytes(R, (int)R.name_tok_len + s.n, quals_set);
    len_passed[i] = passed ? bytes : 0;
    len_failed[i] = passed ? 0 : bytes;
    cnt_passed[i] = passed ? 1 : 0;
    sfx_len[i] = (uint32_t)s.n | (quals_set ? 0x80000000u : 0u);
    is_passed[i] = passed ? 1 : 0;
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
__global__ __launch_bounds__(kNameBlock) void k_write_name(WriteArgs A, const RecPlan *__restrict__ plan, const uint64_t *__restrict__ off_passed,
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
        const RecPlan R = plan[i];
        const bool passed = R.flags & kPassed;
        const int n_sfx = (int)(sfx_len[i] & 0x7FFFFFFFu);
        const uint64_t pos = (passed ? off_passed[i] : off_failed[i]) + 1 + R.name_tok_len;
        if (pos + n_sfx <= (passed ? cap_passed : cap_failed)) {  // otherwise K-WRITE reports the overflow
            char *out = reinterpret_cast<char *>(passed ? out_passed : out_failed) + pos;
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

// K-WRITE: one wave per record copies everything but the suffix.  The two long runs of a record -- bases and qualities -- go out in
// 16-byte pieces per lane: an unaligned 16-byte load from the text (mirrored, and complemented with v_perm_b32, for a read that passed
// on the reverse strand) and an aligned 16-byte store, all loads of a record in flight before its first store.  What is left -- '@',
// the name token, the line ends, the '+' line and the few bytes of each run in front of / behind its 16-byte aligned part -- is copied
// a byte per lane from an address chosen with selects.
__global__ __launch_bounds__(256) void k_write(WriteArgs A, const RecPlan *__restrict__ plan, const uint64_t *__restrict__ off_passed,
                                               const uint64_t *__restrict__ off_failed, const uint32_t *__restrict__ sfx_len,
                                               uint8_t *__restrict__ out_passed, size_t cap_passed, uint8_t *__restrict__ out_failed,
                                               size_t cap_failed, uint64_t *__restrict__ rec_off, uint32_t *__restrict__ err) {
    // (No table in LDS: a block is four records, and filling 512 bytes of LDS behind a barrier before the first load of a record was a fixed cost per
    // four records.  The complement of a dword is two v_perm_b32; a character outside A C G T N takes FastqRecordExt.REVERSE_COMPLEMENT's switch.)
    const int lane = threadIdx.x & 63;
    const size_t i = blockIdx.x * (size_t)4 + (threadIdx.x >> 6);
    if (i >= A.n) return;
    // the same 64 bytes for every lane of the wave: through v_readfirstlane into scalar registers, and with them everything derived
    // from the plan (positions, lengths, segment bounds).  (A persistent form of this kernel -- waves that loop over records and request
    // the next plan row early -- needs 99 VGPRs = 4 waves per SIMD and measured 1.17 ms against 1.08 ms for this one at 8 waves.)
    RecPlan R;
    {
        uint32_t v[16];
        __builtin_memcpy(v, plan + i, 64);
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = __builtin_amdgcn_readfirstlane(v[k]);
        __builtin_memcpy(&R, v, 64);
    }
    const uint32_t sl = __builtin_amdgcn_readfirstlane(sfx_len[i]);
    const uint64_t off_v = (R.flags & kPassed) ? off_passed[i] : off_failed[i];
    const uint64_t off = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(off_v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)off_v);
    const bool passed = R.flags & kPassed, rev = R.flags & kRev;
    const uint64_t tok_len = R.name_tok_len, qh_len = R.qh_len;
    const int len = R.len, cut_beg = R.cut_beg, cut_len = R.cut_len;
    const bool qset = sl >> 31;
    const uint64_t n_sfx = sl & 0x7FFFFFFFu;
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
    const uint8_t *rd = (A.bstart ? A.text : A.reads) + R.rd, *ql = (A.bstart ? A.text : A.quals) + R.ql;
    const uint8_t *tok = A.text + R.name_beg, *qh = A.text + R.qh_beg;
    static __device__ const uint8_t kLiterals[8] = {'@', '\n', '+', 'n', 'u', 'l', 'l', 0};
    const uint8_t *lit = kLiterals;
    // bases / qualities: position k of the written range is raw index cut_beg + k, or len - 1 - cut_beg - k when reversed
    const int64_t step = rev ? -1 : 1;
    const uint8_t *seq0 = rd + (rev ? len - 1 - cut_beg : cut_beg);
    const uint8_t *qual0 = qset ? ql + (rev ? len - 1 - cut_beg : cut_beg) : lit + 3;
    const int64_t qstep = qset ? step : 1;
    const uint32_t rc_sel = rev ? 256u : 0u;
    // ---- the 16-byte aligned inside of the two runs --------------------------------------------------------------------------------
    // output addresses [lo, hi) of a run; its aligned part is [a0, a1) (empty: a0 = a1 = lo)
    const uint64_t obase = (uint64_t)(uintptr_t)out;
    const uint64_t seq_lo = off + p_seq, seq_hi = seq_lo + (uint64_t)cut_len, q_lo = off + p_q, q_hi = q_lo + (qset ? qlen : 0);
    auto aligned_part = [&](uint64_t lo, uint64_t hi, uint64_t &a0, uint64_t &a1) {
        a0 = ((obase + lo + 15) & ~15ull) - obase;
        a1 = ((obase + hi) & ~15ull) - obase;
        if (a0 >= a1) a0 = a1 = lo;
    };
    uint64_t sa0, sa1, qa0, qa1;
    aligned_part(seq_lo, seq_hi, sa0, sa1);
    aligned_part(q_lo, q_hi, qa0, qa1);
    // run positions k .. k+15 in two steps: the load as the text has the bytes (forwards, or the 16 bytes that end at position k of a reversed
    // run), and -- after every load of the turn has been issued -- the mirroring and the complement.  (As one step the complement's validity test
    // sat between the loads: a record on the reverse strand waited for each of its sequence pieces before it asked for the next.)
    auto load16 = [&](const uint8_t *b0, uint64_t k, uint32_t (&v)[4]) {
        if (!rev)
            __builtin_memcpy(v, b0 + k, 16);
        else
            __builtin_memcpy(v, b0 - (int64_t)k - 15, 16);  // positions k+15 .. k
    };
    auto finish16 = [&](bool is_seq, uint32_t (&w)[4]) {
        if (!rev) return;
        uint32_t v[4] = {w[0], w[1], w[2], w[3]};
#pragma unroll
        for (int d = 0; d < 4; d++) w[d] = __builtin_bswap32(v[3 - d]);
        if (is_seq) {
            uint32_t c[4];
            bool ok = true;
#pragma unroll
            for (int d = 0; d < 4; d++) ok = rc4(w[d], c[d]) && ok;
            if (ok) {
#pragma unroll
                for (int d = 0; d < 4; d++) w[d] = c[d];
            } else {
#pragma unroll
                for (int d = 0; d < 4; d++)
                    w[d] = (uint32_t)(uint8_t)rc_char((unsigned char)(w[d] & 0xFF)) | ((uint32_t)(uint8_t)rc_char((unsigned char)((w[d] >> 8) & 0xFF)) << 8) |
                           ((uint32_t)(uint8_t)rc_char((unsigned char)((w[d] >> 16) & 0xFF)) << 16) | ((uint32_t)(uint8_t)rc_char((unsigned char)(w[d] >> 24)) << 24);
            }
        }
    };
    // ---- everything else, a byte per lane: '@' + token | LF + front of the bases | their end + LF + '+' line + front of the
    // qualities | their end + LF (the suffix between the first two is K-WNAME's) -------------------------------------------------
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
    const uint64_t end = off + bytes, s0 = off + p_sfx, s1 = s0 + n_sfx;  // [s0, s1): the suffix K-WNAME writes
    const uint64_t n0 = s0 - off, n1 = sa0 - s1, n2 = qa0 - sa1, n3 = end - qa1, n_loose = n0 + n1 + n2 + n3;
    auto loose_addr = [&](uint64_t u) -> uint64_t {  // u-th byte outside the aligned parts and the suffix
        uint64_t a = off + u;
        a = u >= n0 ? s1 + (u - n0) : a;
        a = u >= n0 + n1 ? sa1 + (u - n0 - n1) : a;
        a = u >= n0 + n1 + n2 ? qa1 + (u - n0 - n1 - n2) : a;
        return a;
    };
    // One turn = per lane two 16-byte pieces of each run and three loose bytes: every load of the turn is issued before its first store,
    // so a record of up to 2 KiB per run costs one round trip to memory after the plan row.
    constexpr int kLoose = 3;
    for (uint64_t turn = 0;; turn++) {
        const uint64_t g = 16ull * lane + 2048 * turn;
        const uint64_t gs0 = sa0 + g, gs1 = gs0 + 1024, gq0 = qa0 + g, gq1 = gq0 + 1024;
        const bool bs0 = gs0 < sa1, bs1 = gs1 < sa1, bq0 = gq0 < qa1, bq1 = gq1 < qa1;
        const uint64_t u0 = (uint64_t)lane + 64ull * kLoose * turn;
        if (!__ballot(bs0 || bq0 || u0 < n_loose)) break;
        uint32_t ws0[4], ws1[4], wq0[4], wq1[4];
        if (bs0) load16(seq0, gs0 - seq_lo, ws0);
        if (bs1) load16(seq0, gs1 - seq_lo, ws1);
        if (bq0) load16(qual0, gq0 - q_lo, wq0);
        if (bq1) load16(qual0, gq1 - q_lo, wq1);
        uint64_t la[kLoose];
        uint32_t lc[kLoose], ltb[kLoose];
#pragma unroll
        for (int j = 0; j < kLoose; j++) {
            const uint64_t u = u0 + 64ull * j;
            la[j] = u < n_loose ? loose_addr(u) : ~0ull;
            const uint8_t *p = source(la[j] != ~0ull ? la[j] - off : 0, ltb[j]);
            lc[j] = *p;
        }
        if (bs0) finish16(true, ws0);
        if (bs1) finish16(true, ws1);
        if (bq0) finish16(false, wq0);
        if (bq1) finish16(false, wq1);
        if (bs0) __builtin_memcpy(__builtin_assume_aligned(out + gs0, 16), ws0, 16);
        if (bs1) __builtin_memcpy(__builtin_assume_aligned(out + gs1, 16), ws1, 16);
        if (bq0) __builtin_memcpy(__builtin_assume_aligned(out + gq0, 16), wq0, 16);
        if (bq1) __builtin_memcpy(__builtin_assume_aligned(out + gq1, 16), wq1, 16);
#pragma unroll
        for (int j = 0; j < kLoose; j++)
            if (la[j] != ~0ull) {
                uint32_t ch = lc[j];
                if (ltb[j]) {  // a base of a reversed record
                    uint32_t c4;
                    ch = rc4(ch * 0x01010101u, c4) ? (c4 & 0xFFu) : (uint32_t)(uint8_t)rc_char((unsigned char)ch);
                }
                out[la[j]] = (uint8_t)ch;
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
    // scratch: 6 arrays of n_out + 1 u64 (3 lengths, 3 scanned), the suffix lengths, the error word, the plan rows, hipcub temp storage
    size_t tmp_bytes = 0;
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (int)(n_out + 1), s));
    const size_t arr = (n_out + 1) * sizeof(uint64_t);
    const size_t sfx_bytes = ((n_out * sizeof(uint32_t) + 255) / 256) * 256;
    const size_t plan_bytes = n_out * sizeof(RecPlan);
    const size_t tmp_off = ((6 * arr + sfx_bytes + 256 + 255) / 256) * 256 + plan_bytes;
    const size_t need = tmp_off + tmp_bytes;
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
    RecPlan *plan = (RecPlan *)(base + tmp_off - plan_bytes);  // 256-byte aligned
    void *cub_tmp = base + tmp_off;
    SMI_HIP(hipMemsetAsync(d_err, 0, 4, s));
    // the extra last element makes the exclusive scans deliver the totals
    SMI_HIP(hipMemsetAsync(lenp + n_out, 0, 8, s));
    SMI_HIP(hipMemsetAsync(lenf + n_out, 0, 8, s));
    SMI_HIP(hipMemsetAsync(cntp + n_out, 0, 8, s));
    WriteArgs A{d_text, d_line_start, d_reads, d_quals, d_bstart, d_qstart, d_offsets, d_frag_src, d_chim, d_scan, d_bc, d_rank,
                n_out,  first_read_id, cfg->five_prime, cfg->trim_fastq};
    const unsigned g = (unsigned)((n_out + 255) / 256);
    hipLaunchKernelGGL(k_write_len, dim3(g), dim3(256), 0, s, A, plan, lenp, lenf, cntp, sfx, d_is_passed, d_err);
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
    SideStreamGuard side_guard{ctx->side_stream, true};  // (an error return before the join drains the side stream: smi_internal.h)
    hipLaunchKernelGGL(k_write_name, dim3((unsigned)((n_out + kNameBlock - 1) / kNameBlock)), dim3(kNameBlock), 0, ctx->side_stream, A, plan, offp, offf, ordp, sfx, d_passed, cap_passed, d_failed,
                       cap_failed);
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipEventRecord(ctx->side_join, ctx->side_stream));
    hipLaunchKernelGGL(k_write, dim3((unsigned)((n_out + 3) / 4)), dim3(256), 0, s, A, plan, offp, offf, sfx, d_passed, cap_passed,
                       d_failed, cap_failed, d_rec_off, d_err);
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipStreamWaitEvent(s, ctx->side_join, 0));
    side_guard.armed = false;
    uint64_t h_stack[4];
    uint64_t *h = static_cast<uint64_t *>(pin_words(ctx));  // (four words: page-locked, or the stack if there is none)
    if (!h) h = h_stack;
    SMI_HIP(hipMemcpyAsync(&h[0], offp + n_out, 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(&h[1], offf + n_out, 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(&h[2], ordp + n_out, 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(&h[3], d_err, 4, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    totals[0] = h[0];
    totals[1] = h[1];
    totals[2] = h[2];
    *errors = (uint32_t)h[3];
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
