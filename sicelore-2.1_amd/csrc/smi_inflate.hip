// smi_inflate.hip -- K-INFLATE: the *.fastq.gz inputs of scanfastq inflated on the device (FastqFileReader.java:L138-150 reads them through a
// GZIPInputStream per file; README.md:155 -- the reference parallelises over input files).  A deflate stream is one serial bit stream, so the
// parallelism is the reference's: one wavefront per file, and inside a file the 64 lanes of the wave decode SPECULATIVELY: lane l decodes the
// token (literal, end-of-block, or length + distance with their extra bits) that would start at bit position p + l, all 64 at once through
// look-up tables in LDS; the true token starts are then the chain 0 -> 0 + bits(0) -> ... walked with scalar lane reads (about a dozen hops per
// 64 bits), the surviving tokens are numbered, their output lengths prefix-summed, and every OUTPUT BYTE of the step gets a lane: it finds its
// token by a binary search and is either the literal itself or a byte of the 32 KiB window (kept in LDS) at the match's distance; a byte whose
// source was produced in the same step waits a round.  Tables are rebuilt per dynamic block (canonical codes, 11-bit primary table for
// literals / lengths and 9-bit for distances, 16- / 64-entry second-level tables for longer codes).  Stored and fixed-Huffman blocks, gzip
// headers with optional fields and multi-member files are handled; CRC-32 and ISIZE of every member are checked afterwards (k_inflate_crc).
// Anything malformed or beyond the bounds given ends the stream with a status and the caller inflates that file with zlib.  Byte / bit work:
// no MFMA.
#include "smi_internal.h"

namespace smi {

namespace {

constexpr int kWin = 32768;
constexpr int kLlBits = 11, kDBits = 9;
constexpr int kLlSub = 1 << (15 - kLlBits), kDSub = 1 << (15 - kDBits);  // 16, 64 entries per second-level table
constexpr int kLlSubMax = 160, kDSubMax = 16;
constexpr int kInWords = 128;  // staged input: 4096 bits

enum { TK_LIT = 0, TK_MATCH = 1, TK_EOB = 2, TK_BAD = 3 };

struct InfLds {
    uint8_t window[kWin];
    uint32_t ll[1 << kLlBits];
    uint32_t lls[kLlSubMax * kLlSub];
    uint32_t dt[1 << kDBits];
    uint32_t dts[kDSubMax * kDSub];
    uint32_t inbuf[kInWords + 4];
    uint8_t lens[320];
    uint16_t code_of[320];
    uint32_t t_off[66];
    uint32_t t_info[66];
    uint32_t n_ll_sub, n_d_sub;
    int32_t hdr_status;          // block header parser -> wave
    uint32_t blk_type, blk_final, stored_len;
    uint64_t hdr_bitpos;         // bit position behind the block header
};

// A table entry says everything about its symbol, so that a token costs one look-up per code (two for a code longer than the primary index):
//   bits 0-3 code length (0: no such code), bits 4-7 extra bits, bits 8-9 kind (0 literal, 1 length or distance, 2 end of block,
//   3 "continue in second-level table `value`"), bits 16-30 value (literal byte, length base, distance base, table number)
enum { EK_LIT = 0, EK_BASE = 1, EK_EOB = 2, EK_SUB = 3 };
__device__ __forceinline__ uint32_t entry_of(uint32_t sym, uint32_t len, bool dist_table) {
    if (dist_table) {
        if (sym > 29u) return 0u;  // codes 30 and 31 of the fixed code never occur in a valid stream
        const uint32_t e = sym < 4u ? 0u : (sym >> 1) - 1u;
        const uint32_t base = sym < 4u ? 1u + sym : 1u + ((2u + (sym & 1u)) << e);
        return len | (e << 4) | (EK_BASE << 8) | (base << 16);
    }
    if (sym < 256u) return len | (EK_LIT << 8) | (sym << 16);
    if (sym == 256u) return len | (EK_EOB << 8);
    if (sym > 285u) return 0u;
    const uint32_t li = sym - 257u;
    if (li == 28u) return len | (EK_BASE << 8) | (258u << 16);
    const uint32_t e = li < 8u ? 0u : (li >> 2) - 1u;
    const uint32_t base = li < 8u ? 3u + li : 3u + ((4u + (li & 3u)) << e);
    return len | (e << 4) | (EK_BASE << 8) | (base << 16);
}

// serial bit reader over global memory (block headers, gzip headers: one lane)
struct BitReader {
    const uint8_t *p;
    uint64_t n_bits, at;
    __device__ bool have(uint32_t k) const { return at + k <= n_bits; }
    __device__ uint32_t peek(uint32_t k) const {  // k <= 24
        const uint64_t b = at >> 3;
        uint32_t v = (uint32_t)p[b] | ((uint32_t)p[b + 1] << 8) | ((uint32_t)p[b + 2] << 16) | ((uint32_t)p[b + 3] << 24);  // (the input is padded)
        return (v >> (at & 7)) & ((1u << k) - 1u);
    }
    __device__ uint32_t get(uint32_t k) {
        const uint32_t v = peek(k);
        at += k;
        return v;
    }
};

// canonical Huffman decode tables from code lengths lens[0 .. n): primary table of 2^P entries, second-level tables of 2^(15 - P).
// (entries: entry_of above; a second-level entry keeps the WHOLE code length).
// Returns false for an over-subscribed code or one that is incomplete in a way inflate does not accept.
template <int P>
__device__ bool build_table(InfLds &L, const uint8_t *lens, int n, uint32_t *prim, uint32_t *sub, int sub_max, uint32_t *n_sub_out, int lane, bool dist_table) {
    constexpr int SUB = 1 << (15 - P);
    for (int i = lane; i < (1 << P); i += 64) prim[i] = 0;
    for (int i = lane; i < sub_max * SUB; i += 64) sub[i] = 0;
    wave_sync();
    // codes, second-level table allocation, completeness: one lane (<= 288 symbols)
    int ok = 1;
    if (lane == 0) {
        uint32_t count[16];
        for (int b = 0; b < 16; b++) count[b] = 0;
        for (int s = 0; s < n; s++) count[lens[s]]++;
        count[0] = 0;
        uint32_t next[16], code = 0;
        int left = 1;
        for (int b = 1; b <= 15; b++) {
            left <<= 1;
            left -= (int)count[b];
            if (left < 0) ok = 0;  // over-subscribed
            code = (code + count[b - 1]) << 1;
            next[b] = code;
        }
        int n_codes = 0;
        for (int b = 1; b <= 15; b++) n_codes += (int)count[b];
        // incomplete codes: zlib accepts them only for a distance code with at most one code of one bit (or none at all)
        if (left > 0 && !((n_codes == 1 && count[1] == 1) || (dist_table && n_codes == 0))) ok = 0;
        uint32_t n_sub = 0;
        for (int s = 0; s < n && ok; s++) {
            const uint32_t len = lens[s];
            if (!len) {
                L.code_of[s] = 0;
                continue;
            }
            const uint32_t c = next[len]++;
            const uint32_t rev = __brev(c) >> (32u - len);
            L.code_of[s] = (uint16_t)rev;
            if (len > (uint32_t)P) {
                const uint32_t pre = rev & ((1u << P) - 1u);
                if (((prim[pre] >> 8) & 3u) != EK_SUB) {
                    if (n_sub >= (uint32_t)sub_max) {
                        ok = 0;
                        break;
                    }
                    prim[pre] = (uint32_t)P | (EK_SUB << 8) | (n_sub << 16);
                    n_sub++;
                }
            }
        }
        *n_sub_out = n_sub;
        L.hdr_status = ok ? 0 : -1;
    }
    wave_sync();
    if (L.hdr_status) return false;
    for (int s = lane; s < n; s += 64) {
        const uint32_t len = lens[s];
        if (!len) continue;
        const uint32_t rev = L.code_of[s];
        const uint32_t e = entry_of((uint32_t)s, len, dist_table);
        if (len <= (uint32_t)P) {
            for (uint32_t k = rev; k < (1u << P); k += 1u << len) prim[k] = e;
        } else {
            const uint32_t t = prim[rev & ((1u << P) - 1u)] >> 16;
            for (uint32_t k = rev >> P; k < (uint32_t)SUB; k += 1u << (len - P)) sub[t * SUB + k] = e;
        }
    }
    wave_sync();
    return true;
}

// a token as one word: bits 0-5 its length in bits (1 .. 48), bits 6-7 kind, bits 8-16 output length (0 .. 258); and its info word
// (literal byte, or length | distance << 9)
__device__ __forceinline__ void decode_token(const InfLds &L, uint64_t x, uint32_t &tok, uint32_t &info) {
    tok = 1u | (TK_BAD << 6);
    info = 0;
    uint32_t e = L.ll[x & ((1u << kLlBits) - 1u)];
    if (((e >> 8) & 3u) == EK_SUB && (e & 15u)) e = L.lls[(e >> 16) * kLlSub + ((x >> kLlBits) & (kLlSub - 1))];
    const uint32_t len = e & 15u, kind = (e >> 8) & 3u;
    if (!len || kind == EK_SUB) return;
    if (kind == EK_LIT) {
        tok = len | (TK_LIT << 6) | (1u << 8);
        info = e >> 16;
        return;
    }
    if (kind == EK_EOB) {
        tok = len | (TK_EOB << 6);
        return;
    }
    const uint32_t le = (e >> 4) & 15u;
    const uint32_t length = (e >> 16) + ((uint32_t)(x >> len) & ((1u << le) - 1u));
    const uint32_t used = len + le;
    const uint64_t y = x >> used;
    uint32_t d = L.dt[y & ((1u << kDBits) - 1u)];
    if (((d >> 8) & 3u) == EK_SUB && (d & 15u)) d = L.dts[(d >> 16) * kDSub + ((y >> kDBits) & (kDSub - 1))];
    const uint32_t dlen = d & 15u;
    if (!dlen || ((d >> 8) & 3u) != EK_BASE) return;
    const uint32_t de = (d >> 4) & 15u;
    const uint32_t dist = (d >> 16) + ((uint32_t)(y >> dlen) & ((1u << de) - 1u));
    tok = (used + dlen + de) | (TK_MATCH << 6) | (length << 8);
    info = length | (dist << 9);
}

}  // namespace

// status of a stream
enum { INF_OK = 0, INF_BAD_HEADER = 1, INF_BAD_BLOCK = 2, INF_BAD_CODE = 3, INF_BAD_DISTANCE = 4, INF_OUT_FULL = 5, INF_TRUNCATED = 6, INF_TOO_MANY_MEMBERS = 7, INF_CRC = 8 };

struct InfStream {   // = smi_inflate_stream
    uint64_t in_off, in_len, out_off, out_cap;
};
struct InfResult {   // = smi_inflate_result
    uint64_t out_len;
    uint32_t status, n_members;
};
struct InfMember {   // one gzip member of a stream
    uint64_t out_start, out_len;  // relative to the stream's output
    uint32_t crc, isize;
};

// One wavefront per stream.  in: the compressed bytes (stream i at in + S[i].in_off, 4-byte aligned; 1 KiB readable behind the last one).
__global__ __launch_bounds__(64) void k_inflate(const uint8_t *__restrict__ in, const InfStream *__restrict__ S, int n_streams, uint8_t *__restrict__ out,
                                                InfResult *__restrict__ R, InfMember *__restrict__ M, int max_members) {
    __shared__ InfLds L;
    const int lane = threadIdx.x;
    const int si = blockIdx.x;
    if (si >= n_streams) return;
    const uint8_t *src = in + S[si].in_off;
    const uint64_t n_in_bits = S[si].in_len * 8;
    uint8_t *dst = out + S[si].out_off;
    const uint64_t out_cap = S[si].out_cap;
    uint64_t out_pos = 0;     // bytes written so far (all members)
    uint64_t bitpos = 0;      // next unread bit of the input
    uint32_t status = INF_OK, n_members = 0;
    InfMember *mem = M + (size_t)si * max_members;
    bool more_members = true;
    while (more_members && status == INF_OK) {
        // ---- gzip header (one lane) -------------------------------------------------------------------------------------------------
        if (lane == 0) {
            int st = 0;
            uint64_t b = bitpos >> 3;
            const uint64_t n = S[si].in_len;
            if (b + 18 > n || src[b] != 0x1f || src[b + 1] != 0x8b || src[b + 2] != 8 || (src[b + 3] & 0xE0))
                st = INF_BAD_HEADER;
            else {
                const uint32_t flg = src[b + 3];
                b += 10;
                if (flg & 4) {
                    if (b + 2 > n)
                        st = INF_TRUNCATED;
                    else
                        b += 2 + ((uint32_t)src[b] | ((uint32_t)src[b + 1] << 8));
                }
                for (int f = 8; f <= 16 && !st; f <<= 1)
                    if (flg & f) {
                        while (b < n && src[b]) b++;
                        b++;
                    }
                if (flg & 2) b += 2;
                if (b + 8 > n) st = st ? st : INF_TRUNCATED;
            }
            L.hdr_status = st;
            L.hdr_bitpos = b * 8;
        }
        wave_sync();
        if (L.hdr_status) {
            status = (uint32_t)L.hdr_status;
            break;
        }
        bitpos = L.hdr_bitpos;
        const uint64_t member_out_start = out_pos;
        // ---- blocks -------------------------------------------------------------------------------------------------------------------
        bool last_block = false;
        while (!last_block && status == INF_OK) {
            // block header (one lane): type, and for a dynamic block the code lengths
            if (lane == 0) {
                BitReader br{src, n_in_bits, bitpos};
                int st = 0;
                if (!br.have(3))
                    st = INF_TRUNCATED;
                else {
                    L.blk_final = br.get(1);
                    L.blk_type = br.get(2);
                    if (L.blk_type == 0) {
                        br.at = (br.at + 7) & ~7ull;
                        if (!br.have(32))
                            st = INF_TRUNCATED;
                        else {
                            const uint32_t len = br.get(16), nlen = br.get(16);
                            if ((len ^ nlen) != 0xFFFFu) st = INF_BAD_BLOCK;
                            L.stored_len = len;
                            if ((br.at >> 3) + len > (n_in_bits >> 3)) st = st ? st : INF_TRUNCATED;
                        }
                    } else if (L.blk_type == 1) {
                        for (int s = 0; s < 144; s++) L.lens[s] = 8;
                        for (int s = 144; s < 256; s++) L.lens[s] = 9;
                        for (int s = 256; s < 280; s++) L.lens[s] = 7;
                        for (int s = 280; s < 288; s++) L.lens[s] = 8;
                        for (int s = 288; s < 320; s++) L.lens[s] = 5;  // the 30 (+2) distance codes of the fixed code
                    } else if (L.blk_type == 2) {
                        if (!br.have(14))
                            st = INF_TRUNCATED;
                        else {
                            const uint32_t hlit = br.get(5) + 257, hdist = br.get(5) + 1, hclen = br.get(4) + 4;
                            if (hlit > 286 || hdist > 30) st = INF_BAD_BLOCK;
                            uint8_t cl[19];
                            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                            for (int k = 0; k < 19; k++) cl[k] = 0;
                            if (!st && !br.have(hclen * 3)) st = INF_TRUNCATED;
                            for (uint32_t k = 0; k < hclen && !st; k++) cl[order[k]] = (uint8_t)br.get(3);
                            // the code-length code: at most 7 bits, decoded by trying the 19 symbols (canonical codes, bit-reversed)
                            uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nxt[8], code = 0;
                            uint8_t clc[19];
                            int left = 1;
                            for (int k = 0; k < 19; k++) cnt[cl[k]]++;
                            cnt[0] = 0;
                            for (int b = 1; b <= 7; b++) {
                                left = (left << 1) - (int)cnt[b];
                                code = (code + cnt[b - 1]) << 1;
                                nxt[b] = code;
                            }
                            if (!st && left != 0) st = INF_BAD_BLOCK;  // (inflate wants the code-length code complete)
                            for (int k = 0; k < 19; k++) clc[k] = cl[k] ? (uint8_t)(__brev(nxt[cl[k]]++) >> (32 - cl[k])) : 0;
                            uint32_t i = 0;
                            const uint32_t n_len = hlit + hdist;
                            while (i < n_len && !st) {
                                if (!br.have(7 + 7)) {
                                    // (a few bits may be missing at the very end of a stream; the padding behind the input makes peek safe)
                                    if (!br.have(1)) {
                                        st = INF_TRUNCATED;
                                        break;
                                    }
                                }
                                const uint32_t w = br.peek(7);
                                int sym = -1;
                                for (int k = 0; k < 19; k++)
                                    if (cl[k] && (w & ((1u << cl[k]) - 1u)) == clc[k]) {
                                        sym = k;
                                        break;
                                    }
                                if (sym < 0) {
                                    st = INF_BAD_BLOCK;
                                    break;
                                }
                                br.at += cl[sym];
                                if (sym < 16)
                                    L.lens[i++] = (uint8_t)sym;
                                else {
                                    uint32_t rep, val = 0;
                                    if (sym == 16) {
                                        if (i == 0) {
                                            st = INF_BAD_BLOCK;
                                            break;
                                        }
                                        val = L.lens[i - 1];
                                        rep = 3 + br.get(2);
                                    } else if (sym == 17)
                                        rep = 3 + br.get(3);
                                    else
                                        rep = 11 + br.get(7);
                                    if (i + rep > n_len) {
                                        st = INF_BAD_BLOCK;
                                        break;
                                    }
                                    while (rep--) L.lens[i++] = (uint8_t)val;
                                }
                            }
                            if (!st && L.lens[256] == 0) st = INF_BAD_BLOCK;  // no end-of-block code
                            if (!st) {
                                // lay the two alphabets out at fixed places: literal / length lengths at 0 .. 287, distances at 288 .. 319
                                uint8_t dl[32];
                                for (uint32_t k = 0; k < 32; k++) dl[k] = k < hdist ? L.lens[hlit + k] : 0;
                                for (uint32_t k = hlit; k < 288; k++) L.lens[k] = 0;
                                for (uint32_t k = 0; k < 32; k++) L.lens[288 + k] = dl[k];
                            }
                        }
                    } else
                        st = INF_BAD_BLOCK;
                }
                if (br.at > n_in_bits) st = st ? st : INF_TRUNCATED;
                L.hdr_status = st;
                L.hdr_bitpos = br.at;
            }
            wave_sync();
            if (L.hdr_status) {
                status = (uint32_t)L.hdr_status;
                break;
            }
            bitpos = L.hdr_bitpos;
            last_block = L.blk_final != 0;
            if (L.blk_type == 0) {
                // stored block: bytes as they are
                const uint32_t len = L.stored_len;
                if (out_pos + len > out_cap) {
                    status = INF_OUT_FULL;
                    break;
                }
                const uint8_t *p = src + (bitpos >> 3);
                for (uint32_t k = (uint32_t)lane; k < len; k += 64) {
                    const uint8_t v = p[k];
                    dst[out_pos + k] = v;
                    L.window[(out_pos + k) & (kWin - 1)] = v;
                }
                wave_sync();
                out_pos += len;
                bitpos += (uint64_t)len * 8;
                continue;
            }
            if (!build_table<kLlBits>(L, L.lens, 288, L.ll, L.lls, kLlSubMax, &L.n_ll_sub, lane, false) ||
                !build_table<kDBits>(L, L.lens + 288, 32, L.dt, L.dts, kDSubMax, &L.n_d_sub, lane, true)) {
                status = INF_BAD_CODE;
                break;
            }
            // ---- the block's tokens, 64 bit positions per step --------------------------------------------------------------------------
            uint64_t buf_base = ~0ull;  // bit position of inbuf[0]; reloaded when the step's window would leave the staged words
            bool block_done = false;
            while (!block_done) {
                if (bitpos >= n_in_bits) {
                    status = INF_TRUNCATED;
                    break;
                }
                if (buf_base == ~0ull || bitpos - buf_base + 64 + 64 + 32 > (uint64_t)kInWords * 32) {
                    buf_base = bitpos & ~31ull;
                    const uint32_t *w = reinterpret_cast<const uint32_t *>(src) + (buf_base >> 5);
                    L.inbuf[lane] = w[lane];
                    L.inbuf[lane + 64] = w[lane + 64];
                    wave_sync();
                }
                // my 64 bits from position bitpos + lane
                const uint32_t o = (uint32_t)(bitpos - buf_base) + (uint32_t)lane;
                const uint32_t wi = o >> 5, sh = o & 31u;
                const uint64_t lo = (uint64_t)L.inbuf[wi] | ((uint64_t)L.inbuf[wi + 1] << 32);
                const uint64_t x = sh ? (lo >> sh) | ((uint64_t)L.inbuf[wi + 2] << (64u - sh)) : lo;
                uint32_t tok, info;
                decode_token(L, x, tok, info);
                // The chain of true token starts.  Six scalar instructions per hop, written out (the compiler unrolled the C form sixty-four
                // times at forty instructions a hop, and a single wave issues one instruction per four cycles): read the step of the token
                // at `cur`, mark the lane, advance.  A token that ends the walk (end of block, bad code) carries a step of 64 or more.
                const uint32_t kind = (tok >> 6) & 3u;
                const uint32_t step = kind == TK_BAD ? 127u : kind == TK_EOB ? (tok & 63u) | 64u : tok & 63u;
                uint64_t valid;
                uint32_t cur, t_s;
                asm volatile(
                    "s_mov_b64 %[valid], 0\n\t"
                    "s_mov_b32 %[cur], 0\n"
                    "1:\n\t"
                    "s_nop 3\n\t"  // (a lane select written by the scalar unit needs four wait states before v_readlane reads it)
                    "v_readlane_b32 %[t], %[step], %[cur]\n\t"
                    "s_bitset1_b64 %[valid], %[cur]\n\t"
                    "s_add_u32 %[cur], %[cur], %[t]\n\t"
                    "s_cmp_lt_u32 %[cur], 64\n\t"
                    "s_cbranch_scc1 1b"
                    : [valid] "=&s"(valid), [cur] "=&s"(cur), [t] "=&s"(t_s)
                    : [step] "v"(step)
                    : "scc");
                int ended = 0;  // 1: end of block, 2: bad code
                {
                    const uint32_t last = 63u - (uint32_t)__builtin_clzll(valid);
                    const uint32_t tl = (uint32_t)__builtin_amdgcn_readlane((int)tok, (int)last);
                    const uint32_t kl = (tl >> 6) & 3u;
                    if (kl == TK_BAD)
                        ended = 2;
                    else if (kl == TK_EOB) {
                        ended = 1;
                        cur = last + (tl & 63u);
                    }
                }
                if (ended == 2) {
                    status = INF_BAD_CODE;
                    break;
                }
                if (bitpos + cur > n_in_bits) {
                    status = INF_TRUNCATED;
                    break;
                }
                // the tokens numbered and their output added up: rank among the marked lanes, inclusive scan of the output lengths
                const bool mine = (valid >> lane) & 1ull;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(valid >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)valid, 0u));
                const uint32_t n_tok = (uint32_t)__popcll(valid);
                const uint32_t olen = mine ? tok >> 8 : 0u;
                uint32_t inc = olen;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t y = __shfl_up(inc, d);
                    if (lane >= d) inc += y;
                }
                const uint32_t total_out = __shfl(inc, 63);
                if (mine) {
                    L.t_off[rank] = inc - olen;
                    L.t_info[rank] = info | (kind == TK_MATCH ? 0x80000000u : 0u);
                }
                if (lane == 0) L.t_off[n_tok] = total_out;
                wave_sync();
                if (out_pos + total_out > out_cap) {
                    status = INF_OUT_FULL;
                    break;
                }
                // every output byte of the step gets a lane
                bool bad_dist = false;
                for (uint32_t g = 0; g < total_out; g += 64) {
                    const uint32_t b = g + (uint32_t)lane;
                    const bool act = b < total_out;
                    uint32_t v = 0, src_rel = 0;  // src_rel: source byte of a match relative to the step's first output byte (when it lies inside the step)
                    bool done = !act, in_step = false;
                    uint64_t s_abs = 0;
                    if (act) {
                        uint32_t lo_t = 0, hi_t = n_tok;  // largest token with t_off <= b (tokens without output share their successor's offset)
                        while (hi_t - lo_t > 1) {
                            const uint32_t mid = (lo_t + hi_t) >> 1;
                            if (L.t_off[mid] <= b)
                                lo_t = mid;
                            else
                                hi_t = mid;
                        }
                        const uint32_t info = L.t_info[lo_t];
                        if (!(info & 0x80000000u)) {
                            v = info & 0xFFu;
                            done = true;
                        } else {
                            const uint32_t dist = (info >> 9) & 0xFFFFu, k = b - L.t_off[lo_t];
                            const uint64_t d0 = out_pos + L.t_off[lo_t];  // where the match's first byte goes
                            // (a distance may not reach in front of its own member: zlib and the host decoder refuse it as "too far back")
                            if (dist > d0 - member_out_start || dist > (uint32_t)kWin) {
                                bad_dist = true;
                                done = true;
                            } else {
                                s_abs = d0 - dist + (k < dist ? k : k % dist);
                                if (s_abs < out_pos) {
                                    v = L.window[s_abs & (kWin - 1)];
                                    done = true;
                                } else {
                                    in_step = true;
                                    src_rel = (uint32_t)(s_abs - out_pos);
                                }
                            }
                        }
                    }
                    if (__ballot(bad_dist)) break;
                    // bytes that are known go out; a byte whose source is a byte of this step waits until that one is out (sources lie before the
                    // match they belong to, so every round settles at least the first waiting byte)
                    uint64_t settled = 0;  // of this group of 64
                    bool written = false;
                    for (;;) {
                        if (act && done && !written) {
                            L.window[(out_pos + b) & (kWin - 1)] = (uint8_t)v;
                            dst[out_pos + b] = (uint8_t)v;
                            written = true;
                        }
                        wave_sync();
                        settled = __ballot(written || !act);
                        if (settled == ~0ull) break;
                        if (act && !done && in_step) {
                            const bool ready = src_rel < g || ((settled >> (src_rel - g)) & 1ull);
                            if (ready) {
                                v = L.window[s_abs & (kWin - 1)];
                                done = true;
                            }
                        }
                    }
                }
                if (__ballot(bad_dist)) {
                    status = INF_BAD_DISTANCE;
                    break;
                }
                out_pos += total_out;
                bitpos += cur;
                if (ended == 1) block_done = true;
            }
        }
        if (status != INF_OK) break;
        // ---- trailer: CRC-32 and ISIZE behind the last block, at a byte boundary -----------------------------------------------------------
        if (lane == 0) {
            int st = 0;
            const uint64_t b = (bitpos + 7) >> 3;
            if (b + 8 > S[si].in_len)
                st = INF_TRUNCATED;
            else if ((int)n_members >= max_members)
                st = INF_TOO_MANY_MEMBERS;
            else {
                InfMember m;
                m.out_start = member_out_start;
                m.out_len = out_pos - member_out_start;
                m.crc = (uint32_t)src[b] | ((uint32_t)src[b + 1] << 8) | ((uint32_t)src[b + 2] << 16) | ((uint32_t)src[b + 3] << 24);
                m.isize = (uint32_t)src[b + 4] | ((uint32_t)src[b + 5] << 8) | ((uint32_t)src[b + 6] << 16) | ((uint32_t)src[b + 7] << 24);
                if (m.isize != (uint32_t)m.out_len) st = INF_CRC;
                mem[n_members] = m;
            }
            L.hdr_status = st;
            L.hdr_bitpos = (b + 8) * 8;
        }
        wave_sync();
        if (L.hdr_status) {
            status = (uint32_t)L.hdr_status;
            break;
        }
        n_members++;
        bitpos = L.hdr_bitpos;
        more_members = (bitpos >> 3) < S[si].in_len;  // another member behind this one (trailing zero padding is not expected in these files)
    }
    if (lane == 0) {
        R[si].out_len = out_pos;
        R[si].status = status;
        R[si].n_members = n_members;
    }
}

namespace {
constexpr uint32_t kPoly = 0xEDB88320u;
__device__ __forceinline__ uint32_t gfm(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ kPoly : b >> 1;
    }
    return p;
}
__device__ uint32_t xpow8(uint64_t n) {  // x^(8 n) mod P
    uint32_t p = 1u << 31, q = 1u << 30;
    for (int k = 0; k < 3; k++) q = gfm(q, q);  // x^8
    for (; n; n >>= 1) {
        if (n & 1u) p = gfm(q, p);
        q = gfm(q, q);
    }
    return p;
}
}  // namespace

// CRC-32 of every member's output against its trailer: one workgroup per (member, 64 KiB piece); piece CRCs are combined by GF(2)
// multiplication as in K-DEFLATE (crc(A || B) = crc(A) x^(8 |B|) + crc(B)).  acc[member] must be zero before.
struct CrcPiece {
    uint64_t off, len, behind;  // absolute offset in `text`, bytes, bytes of the member behind the piece
    uint32_t member;
    uint32_t pad;
};
__global__ __launch_bounds__(256) void k_inflate_crc(const uint8_t *__restrict__ text, const CrcPiece *__restrict__ P, uint32_t *__restrict__ acc) {
    __shared__ uint32_t tab[256];
    __shared__ uint32_t part[4];
    const int tid = threadIdx.x;
    {
        uint32_t c = (uint32_t)tid;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ kPoly : c >> 1;
        tab[tid] = c;
    }
    __syncthreads();
    const CrcPiece pc = P[blockIdx.x];
    const uint64_t per = (pc.len + 255) / 256;
    const uint64_t a = min(pc.len, per * (uint64_t)tid), b = min(pc.len, a + per);
    uint32_t crc = 0xFFFFFFFFu;
    const uint8_t *p = text + pc.off;
    for (uint64_t i = a; i < b; i++) crc = tab[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    uint32_t v = b > a ? gfm(xpow8(pc.behind + (pc.len - b)), ~crc) : 0u;
#pragma unroll
    for (int o = 32; o; o >>= 1) v ^= __shfl_xor(v, o);
    if ((tid & 63) == 0) part[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) atomicXor(&acc[pc.member], part[0] ^ part[1] ^ part[2] ^ part[3]);
}

}  // namespace smi

using namespace smi;

// host entry: see sicelore_mi.h
extern "C" int smi_gz_inflate_device(smi_ctx *ctx, const uint8_t *d_in, const smi_inflate_stream *streams, int n_streams, uint8_t *d_out,
                                     smi_inflate_result *results, void *stream) {
    static_assert(sizeof(smi_inflate_stream) == sizeof(InfStream) && sizeof(smi_inflate_result) == sizeof(InfResult), "layout");
    if (!ctx || n_streams < 0 || (n_streams && (!d_in || !streams || !d_out || !results))) {
        set_error("smi_gz_inflate_device: null argument");
        return SMI_ERR_INVALID;
    }
    if (!n_streams) return SMI_OK;
    for (int i = 0; i < n_streams; i++)
        if (streams[i].in_off & 3u) {
            set_error("smi_gz_inflate_device: stream offsets must be multiples of 4 (and 1 KiB of readable bytes must follow the last stream)");
            return SMI_ERR_INVALID;
        }
    SMI_HIP(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    constexpr int kMaxMembers = 256;  // per file (a file of this pipeline has one member per 100 k-read chunk)
    InfStream *d_S = nullptr;
    InfResult *d_R = nullptr;
    InfMember *d_M = nullptr;
    SMI_HIP(hipMalloc((void **)&d_S, (size_t)n_streams * sizeof(InfStream)));
    SMI_HIP(hipMalloc((void **)&d_R, (size_t)n_streams * sizeof(InfResult)));
    SMI_HIP(hipMalloc((void **)&d_M, (size_t)n_streams * kMaxMembers * sizeof(InfMember)));
    auto cleanup = [&]() {
        (void)hipFree(d_S);
        (void)hipFree(d_R);
        (void)hipFree(d_M);
    };
    int rc = SMI_OK;
    std::vector<InfMember> members;
    std::vector<CrcPiece> pieces;
    std::vector<uint32_t> crc_acc;
    do {
        if (hipMemcpyAsync(d_S, streams, (size_t)n_streams * sizeof(InfStream), hipMemcpyHostToDevice, s) != hipSuccess) {
            rc = SMI_ERR_HIP;
            break;
        }
        hipLaunchKernelGGL(k_inflate, dim3((unsigned)n_streams), dim3(64), 0, s, d_in, d_S, n_streams, d_out, d_R, d_M, kMaxMembers);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(results, d_R, (size_t)n_streams * sizeof(InfResult), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            rc = SMI_ERR_HIP;
            break;
        }
        // CRC-32 of every member of the streams that came through
        members.resize((size_t)n_streams * kMaxMembers);
        std::vector<std::pair<int, uint32_t>> which;  // (stream, member)
        for (int i = 0; i < n_streams; i++) {
            if (results[i].status != INF_OK || !results[i].n_members) continue;
            if (hipMemcpy(members.data() + (size_t)i * kMaxMembers, d_M + (size_t)i * kMaxMembers, results[i].n_members * sizeof(InfMember),
                          hipMemcpyDeviceToHost) != hipSuccess) {
                rc = SMI_ERR_HIP;
                break;
            }
            for (uint32_t m = 0; m < results[i].n_members; m++) {
                const InfMember &mm = members[(size_t)i * kMaxMembers + m];
                const uint32_t id = (uint32_t)which.size();
                which.emplace_back(i, m);
                constexpr uint64_t kPiece = 1u << 20;
                for (uint64_t at = 0; at < mm.out_len; at += kPiece) {
                    const uint64_t len = std::min<uint64_t>(kPiece, mm.out_len - at);
                    pieces.push_back(CrcPiece{streams[i].out_off + mm.out_start + at, len, mm.out_len - at - len, id, 0});
                }
            }
        }
        if (rc != SMI_OK || which.empty()) break;
        crc_acc.assign(which.size(), 0);
        CrcPiece *d_P = nullptr;
        uint32_t *d_acc = nullptr;
        if (hipMalloc((void **)&d_P, std::max<size_t>(pieces.size(), 1) * sizeof(CrcPiece)) != hipSuccess ||
            hipMalloc((void **)&d_acc, which.size() * 4) != hipSuccess) {
            (void)hipFree(d_P);
            rc = SMI_ERR_HIP;
            break;
        }
        bool ok = hipMemsetAsync(d_acc, 0, which.size() * 4, s) == hipSuccess;
        if (ok && !pieces.empty()) {
            ok = hipMemcpyAsync(d_P, pieces.data(), pieces.size() * sizeof(CrcPiece), hipMemcpyHostToDevice, s) == hipSuccess;
            if (ok) hipLaunchKernelGGL(k_inflate_crc, dim3((unsigned)pieces.size()), dim3(256), 0, s, d_out, d_P, d_acc);
            ok = ok && hipGetLastError() == hipSuccess;
        }
        ok = ok && hipMemcpyAsync(crc_acc.data(), d_acc, which.size() * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        (void)hipFree(d_P);
        (void)hipFree(d_acc);
        if (!ok) {
            rc = SMI_ERR_HIP;
            break;
        }
        for (size_t k = 0; k < which.size(); k++) {
            const InfMember &mm = members[(size_t)which[k].first * kMaxMembers + which[k].second];
            if (crc_acc[k] != mm.crc) results[which[k].first].status = INF_CRC;
        }
    } while (false);
    cleanup();
    if (rc != SMI_OK) set_error("smi_gz_inflate_device: HIP error");
    return rc;
}
