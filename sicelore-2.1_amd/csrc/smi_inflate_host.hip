// smi_inflate_host.hip -- gzip / DEFLATE decoding on the host's cores (RFC 1951, RFC 1952).
//
// Replaces java.util.zip.GZIPInputStream under htsjdk's FastqReader (FJ!nanoporereadscanner/readerwriter/FastqFileReader.java:L138-150)
// for the *.fastq.gz inputs of scanfastq, where a few files are open at a time (K-INFLATE, smi_inflate.hip, takes a whole directory of
// them at once and pays from about a thousand files on).  FASTQ text is literal-heavy (base calls and quality strings find few LZ77
// matches), so the decoder is built around literals:
//   * a 64-bit bit buffer refilled without a branch (one unaligned 8-byte load per refill),
//   * an 11-bit primary literal/length table whose entries hold TWO literals when both codes fit in the 11 bits (quality symbols take
//     about five bits, bases two or three), written with one 2-byte store; up to three lookups per refill,
//   * lengths / end-of-block / long codes through the same entry word, distances through an 8-bit primary table; subtables for longer codes,
//   * matches copied eight bytes at a time when the distance allows,
//   * CRC-32 of the output by carry-less multiplication (PCLMULQDQ folding) where the CPU has it, slicing-by-8 otherwise.
// Every member's CRC-32 and ISIZE are checked.  The decoder needs the whole input and one contiguous output buffer.
#include <immintrin.h>

#include <cstring>
#include <mutex>
#include <string>

#include "smi_internal.h"

using namespace smi;

namespace {

// ---------------------------------------------------------------- CRC-32 (IEEE 802.3, reflected, as gzip uses it)
uint32_t g_crc_table[8][256];
std::once_flag g_crc_once;

void crc_tables() {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        g_crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++) g_crc_table[t][i] = (g_crc_table[t - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[t - 1][i] & 0xFF];
}

// crc = running value with the pre/post inversion already applied by the caller (i.e. the raw register)
uint32_t crc_slice8(uint32_t crc, const uint8_t *p, size_t n) {
    while (n && ((uintptr_t)p & 7)) {
        crc = (crc >> 8) ^ g_crc_table[0][(crc ^ *p++) & 0xFF];
        n--;
    }
    while (n >= 8) {
        uint64_t v;
        std::memcpy(&v, p, 8);
        v ^= crc;
        crc = g_crc_table[7][v & 0xFF] ^ g_crc_table[6][(v >> 8) & 0xFF] ^ g_crc_table[5][(v >> 16) & 0xFF] ^ g_crc_table[4][(v >> 24) & 0xFF] ^
              g_crc_table[3][(v >> 32) & 0xFF] ^ g_crc_table[2][(v >> 40) & 0xFF] ^ g_crc_table[1][(v >> 48) & 0xFF] ^ g_crc_table[0][v >> 56];
        p += 8;
        n -= 8;
    }
    while (n--) crc = (crc >> 8) ^ g_crc_table[0][(crc ^ *p++) & 0xFF];
    return crc;
}

// Folding by carry-less multiplication (V. Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", 2009):
// four 128-bit accumulators folded over 64 bytes per step, then 4 -> 1, 128 -> 64 -> 32 bits and a Barrett reduction.  Constants for the
// reflected polynomial 0xEDB88320: x^(4*128+32), x^(4*128-32), x^(128+32), x^(128-32), x^64 mod P (bit-reflected, shifted left by one), P', mu.
__attribute__((target("pclmul,sse4.1"))) uint32_t crc_pclmul(uint32_t crc, const uint8_t *p, size_t n) {
    if (n < 64) return crc_slice8(crc, p, n);
    const __m128i k1k2 = _mm_set_epi64x(0x00000001c6e41596ll, 0x0000000154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00000000ccaa009ell, 0x00000001751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0000000163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x00000001f7011641ll, 0x00000001db710641ll);
    const __m128i mask32 = _mm_set_epi32(0, 0, 0, -1);
#define ld(q) _mm_loadu_si128((const __m128i *)(q))
#define fold(acc, next) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc, k3k4, 0x11), _mm_clmulepi64_si128(acc, k3k4, 0x00)), next)
    __m128i x1 = _mm_xor_si128(ld(p), _mm_cvtsi32_si128((int)crc)), x2 = ld(p + 16), x3 = ld(p + 32), x4 = ld(p + 48);
    p += 64;
    n -= 64;
    while (n >= 64) {
        __m128i t1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), t2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        __m128i t3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), t4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k1k2, 0x11), t1), ld(p));
        x2 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x2, k1k2, 0x11), t2), ld(p + 16));
        x3 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x3, k1k2, 0x11), t3), ld(p + 32));
        x4 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x4, k1k2, 0x11), t4), ld(p + 48));
        p += 64;
        n -= 64;
    }
    x1 = fold(x1, x2);
    x1 = fold(x1, x3);
    x1 = fold(x1, x4);
    while (n >= 16) {
        x1 = fold(x1, ld(p));
        p += 16;
        n -= 16;
    }
    // 128 -> 64 bits (this also appends the 32 zero bits of the CRC definition)
    __m128i t = _mm_clmulepi64_si128(k3k4, x1, 0x01);  // k4 * x1.lo
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    // 64 -> 32 bits
    __m128i hi = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00), hi);
    // Barrett reduction
    __m128i keep = x1;
    x1 = _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10);
    x1 = _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x00);
    x1 = _mm_xor_si128(x1, keep);
    crc = (uint32_t)_mm_extract_epi32(x1, 1);
    return n ? crc_slice8(crc, p, n) : crc;
#undef ld
#undef fold
}

bool g_have_pclmul = false;

}  // namespace

namespace smi {

// CRC-32 of p[0 .. n) continued from `crc` (0 for a fresh one), the value zlib's crc32() returns
uint32_t host_crc32(uint32_t crc, const uint8_t *p, size_t n) {
    std::call_once(g_crc_once, [] {
        crc_tables();
        g_have_pclmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    });
    crc = ~crc;
    crc = g_have_pclmul ? crc_pclmul(crc, p, n) : crc_slice8(crc, p, n);
    return ~crc;
}

}  // namespace smi

namespace {

// ---------------------------------------------------------------- tables
constexpr int LP = 11, DP = 8;                      // primary table bits
constexpr uint32_t LIT = 1u << 31, TWO = 1u << 30, EXC = 1u << 29, EOB = 1u << 28;
constexpr int LIT_CAP = (1 << LP) + 2048, DIST_CAP = (1 << DP) + 1024;

// literal / length entry:  LIT: bits 0-7 code bits consumed, 8-15 first literal, 16-23 second literal (TWO)
//                          EXC | EOB: end of block, bits 0-7 code bits
//                          EXC: subtable, bits 0-7 = LP, 8-23 start index, 24-27 subtable bits
//                          else a length: bits 0-7 code bits + extra bits, 8-15 extra bits, 16-24 base length;  0 = no code
// distance entry:          bits 0-7 code bits + extra bits, 8-15 extra bits (0x80 | subtable bits for a subtable), 16-31 base distance / subtable start
// (the extra bits are read out of the buffer before ONE shift drops code and extra bits together: one shift less in the dependency chain)
struct Tables {
    uint32_t lit[LIT_CAP];
    uint32_t dist[DIST_CAP];
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t reverse_bits(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; i++) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// canonical Huffman decoding table from code lengths (RFC 1951 3.2.2), codes read least-significant bit first.
// entry_of(symbol, len) gives the entry with `len` in bits 0-7.  Returns false for an over-subscribed set, for an incomplete set unless
// allow_incomplete (distance codes: one code, or none at all), and when the subtables do not fit.
template <typename F>
bool build_table(const uint8_t *lens, int n_sym, int primary, uint32_t *table, int cap, bool is_dist, bool allow_incomplete, F entry_of) {
    int count[16] = {0};
    for (int s = 0; s < n_sym; s++) count[lens[s]]++;
    const int n_codes = n_sym - count[0];
    count[0] = 0;
    uint32_t first[16];
    uint32_t code = 0;
    int64_t left = 1;
    for (int l = 1; l <= 15; l++) {
        left = (left << 1) - count[l];
        if (left < 0) return false;  // over-subscribed
        code = (code + (uint32_t)count[l - 1]) << 1;
        first[l] = code;
    }
    // incomplete: zlib takes a single code of one bit (and, for distances, no code at all); anything else is an error
    if (left > 0 && !(allow_incomplete && ((n_codes == 1 && count[1] == 1) || (is_dist && n_codes == 0)))) return false;
    const uint32_t psize = 1u << primary;
    std::memset(table, 0, psize * sizeof(uint32_t));
    bool any_long = false;
    for (int l = primary + 1; l <= 15; l++) any_long |= count[l] != 0;
    uint8_t longest[1 << LP];  // longest code behind every primary slot
    uint32_t nx[16];
    if (any_long) {
        std::memset(longest, 0, psize);
        std::memcpy(nx, first, sizeof nx);
        for (int s = 0; s < n_sym; s++) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t rev = reverse_bits(nx[l]++, l);
            if (l > primary) {
                uint8_t &m = longest[rev & (psize - 1)];
                if (l > m) m = (uint8_t)l;
            }
        }
    }
    uint32_t used = psize;
    std::memcpy(nx, first, sizeof nx);
    for (int s = 0; s < n_sym; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t rev = reverse_bits(nx[l]++, l);
        const uint32_t e = entry_of(s, l);
        if (l <= primary) {
            for (uint32_t i = rev; i < psize; i += 1u << l) table[i] = e;
            continue;
        }
        const uint32_t slot = rev & (psize - 1);
        const int sub_bits = longest[slot] - primary;
        if (table[slot] == 0) {
            if (used + (1u << sub_bits) > (uint32_t)cap) return false;
            std::memset(table + used, 0, sizeof(uint32_t) << sub_bits);
            table[slot] = is_dist ? ((used << 16) | ((0x80u | (uint32_t)sub_bits) << 8) | (uint32_t)primary)
                                  : (EXC | ((uint32_t)sub_bits << 24) | (used << 8) | (uint32_t)primary);
            used += 1u << sub_bits;
        }
        const uint32_t start = is_dist ? (table[slot] >> 16) : ((table[slot] >> 8) & 0xFFFF);
        for (uint32_t i = rev >> primary; i < (1u << sub_bits); i += 1u << (l - primary)) table[start + i] = e;
    }
    return true;
}

inline uint32_t litlen_entry(int sym, int len) {
    if (sym < 256) return LIT | ((uint32_t)sym << 8) | (uint32_t)len;
    if (sym == 256) return EXC | EOB | (uint32_t)len;
    if (sym > 285) return 0;  // 286, 287: not valid in data (they take part in the code)
    return ((uint32_t)kLenBase[sym - 257] << 16) | ((uint32_t)kLenExtra[sym - 257] << 8) | (uint32_t)(len + kLenExtra[sym - 257]);
}
inline uint32_t dist_entry(int sym, int len) {
    if (sym > 29) return 0;
    return ((uint32_t)kDistBase[sym] << 16) | ((uint32_t)kDistExtra[sym] << 8) | (uint32_t)(len + kDistExtra[sym]);
}

// pairs of literals in the primary table: slot i starts with a literal of l1 < LP bits; when the slot of the bits behind it is a
// literal whose code fits in the remaining LP - l1 bits, both go into slot i
void pair_literals(uint32_t *lit) {
    uint32_t single[1 << LP];
    std::memcpy(single, lit, sizeof single);
    for (uint32_t i = 0; i < (1u << LP); i++) {
        const uint32_t e1 = single[i];
        if (!(e1 & LIT)) continue;
        const uint32_t l1 = e1 & 0xFF;
        if (l1 >= (uint32_t)LP) continue;
        const uint32_t e2 = single[i >> l1];
        if (!(e2 & LIT)) continue;
        const uint32_t l2 = e2 & 0xFF;
        if (l1 + l2 > (uint32_t)LP) continue;
        lit[i] = LIT | TWO | (((e2 >> 8) & 0xFF) << 16) | (e1 & 0xFF00) | (l1 + l2);
    }
}

bool build_litlen(const uint8_t *lens, int n, Tables &t) {
    if (!build_table(lens, n, LP, t.lit, LIT_CAP, false, true, litlen_entry)) return false;
    pair_literals(t.lit);
    return true;
}
bool build_dist(const uint8_t *lens, int n, Tables &t) { return build_table(lens, n, DP, t.dist, DIST_CAP, true, true, dist_entry); }

struct FixedTables {
    Tables t;
    FixedTables() {
        uint8_t l[288];
        for (int i = 0; i < 288; i++) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        build_litlen(l, 288, t);
        uint8_t d[32];
        std::memset(d, 5, sizeof d);
        build_table(d, 32, DP, t.dist, DIST_CAP, true, true, dist_entry);
    }
};

// ---------------------------------------------------------------- the decoder
enum { INF_OK = 0, INF_BAD = 1, INF_TRUNCATED = 2, INF_OUT_FULL = 3 };

struct Bits {
    const uint8_t *in, *in_end;
    uint64_t buf = 0;
    int cnt = 0;       // valid bits in buf
    int overrun = 0;   // zero bytes fed behind the end of the input
    inline void refill_fast() {  // needs in + 8 <= in_end
        uint64_t v;
        std::memcpy(&v, in, 8);
        buf |= v << cnt;
        in += (63 - cnt) >> 3;
        cnt |= 56;
    }
    inline void refill_safe() {
        while (cnt < 56) {
            uint64_t b = 0;
            if (in < in_end)
                b = *in++;
            else
                overrun++;
            buf |= b << cnt;
            cnt += 8;
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline void drop(int n) {
        buf >>= n;
        cnt -= n;
    }
};

int read_dynamic_header(Bits &b, Tables &t) {
    b.refill_safe();
    const int hlit = (int)b.peek(5) + 257;
    b.drop(5);
    const int hdist = (int)b.peek(5) + 1;
    b.drop(5);
    const int hclen = (int)b.peek(4) + 4;
    b.drop(4);
    if (hlit > 286 || hdist > 30) return INF_BAD;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; i++) {
        if (b.cnt < 3) b.refill_safe();
        cl[order[i]] = (uint8_t)b.peek(3);
        b.drop(3);
    }
    uint32_t pre[128];
    if (!build_table(cl, 19, 7, pre, 128, true, false, [](int s, int l) { return ((uint32_t)s << 16) | (uint32_t)l; })) return INF_BAD;
    uint8_t lens[286 + 30 + 138];
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
        b.refill_safe();
        const uint32_t e = pre[b.peek(7)];
        if (!(e & 0xFF)) return INF_BAD;
        b.drop((int)(e & 0xFF));
        const int s = (int)(e >> 16);
        if (s < 16) {
            lens[i++] = (uint8_t)s;
            continue;
        }
        int rep;
        uint8_t v = 0;
        if (s == 16) {
            if (!i) return INF_BAD;
            v = lens[i - 1];
            rep = 3 + (int)b.peek(2);
            b.drop(2);
        } else if (s == 17) {
            rep = 3 + (int)b.peek(3);
            b.drop(3);
        } else {
            rep = 11 + (int)b.peek(7);
            b.drop(7);
        }
        if (i + rep > total) return INF_BAD;
        std::memset(lens + i, v, (size_t)rep);
        i += rep;
    }
    if (b.overrun > 8) return INF_TRUNCATED;
    if (lens[256] == 0) return INF_BAD;  // no end-of-block code
    if (!build_litlen(lens, hlit, t)) return INF_BAD;
    if (!build_dist(lens + hlit, hdist, t)) return INF_BAD;
    return INF_OK;
}

inline void copy_match(uint8_t *dst, uint32_t dist, uint32_t len) {  // may write up to 7 bytes past dst + len
    const uint8_t *src = dst - dist;
    if (dist >= 8) {
        uint8_t *end = dst + len;
        do {
            uint64_t v;
            std::memcpy(&v, src, 8);
            std::memcpy(dst, &v, 8);
            src += 8;
            dst += 8;
        } while (dst < end);
    } else if (dist == 1) {
        std::memset(dst, *src, len);
    } else {
        for (uint32_t i = 0; i < len; i++) dst[i] = src[i];
    }
}

// one block's symbols.  Fast loop while there is slack on both sides, then symbol by symbol with every bound checked.
__attribute__((always_inline)) inline int decode_block_body(Bits &b, const Tables &t, uint8_t *out_start, uint8_t *&out_pos, uint8_t *out_end) {
    uint8_t *out = out_pos;
    const uint32_t *lt = t.lit, *dt = t.dist;
    for (;;) {
        // ---- fast: >= 16 input bytes and >= 280 output bytes of room.  The entry of the NEXT symbol is looked up before a match is copied
        //      (and after a run of literals), so that the table load overlaps the copy; `e` is carried from one turn to the next
        if (b.in_end - b.in >= 16 && out_end - out >= 280) {
            b.refill_fast();
            uint32_t e = lt[b.buf & ((1u << LP) - 1)];
            for (;;) {
                if (e & LIT) {
                    uint16_t w = (uint16_t)(e >> 8);
                    std::memcpy(out, &w, 2);
                    out += 1 + ((e >> 30) & 1);
                    b.drop((int)(e & 0xFF));
                    e = lt[b.buf & ((1u << LP) - 1)];
                    if (e & LIT) {
                        w = (uint16_t)(e >> 8);
                        std::memcpy(out, &w, 2);
                        out += 1 + ((e >> 30) & 1);
                        b.drop((int)(e & 0xFF));
                        e = lt[b.buf & ((1u << LP) - 1)];
                        if (e & LIT) {
                            w = (uint16_t)(e >> 8);
                            std::memcpy(out, &w, 2);
                            out += 1 + ((e >> 30) & 1);
                            b.drop((int)(e & 0xFF));
                            if (!(b.in_end - b.in >= 16 && out_end - out >= 280)) break;
                            b.refill_fast();
                            e = lt[b.buf & ((1u << LP) - 1)];
                            continue;
                        }
                    }
                }
                // not a literal: at least 56 - 22 = 34 bits are left
                uint32_t len, dist;
                if (e & EXC) {
                    if (e & EOB) {
                        b.drop((int)(e & 0xFF));
                        out_pos = out;
                        return INF_OK;
                    }
                    e = lt[((e >> 8) & 0xFFFF) + ((b.buf >> LP) & ((1u << ((e >> 24) & 15)) - 1))];
                    if (e & LIT) {
                        *out++ = (uint8_t)(e >> 8);
                        b.drop((int)(e & 0xFF));
                        if (!(b.in_end - b.in >= 16 && out_end - out >= 280)) break;
                        b.refill_fast();
                        e = lt[b.buf & ((1u << LP) - 1)];
                        continue;
                    }
                    if (e & EXC) {  // end of block behind a long code
                        b.drop((int)(e & 0xFF));
                        out_pos = out;
                        return INF_OK;
                    }
                }
                if (!(e & 0xFF)) return INF_BAD;
                {
                    const int tot = (int)(e & 0xFF), xb = (int)((e >> 8) & 0xFF);
                    len = (e >> 16) + ((uint32_t)(b.buf >> (tot - xb)) & ((1u << xb) - 1));
                    b.drop(tot);
                }
                if (b.cnt < 28) b.refill_fast();  // a distance takes up to 15 + 13 bits
                uint32_t d = dt[b.buf & ((1u << DP) - 1)];
                if (d & 0x8000) d = dt[(d >> 16) + ((b.buf >> DP) & ((1u << ((d >> 8) & 0x7F)) - 1))];
                if (!(d & 0xFF)) return INF_BAD;
                {
                    const int tot = (int)(d & 0xFF), db = (int)((d >> 8) & 0xFF);
                    dist = (d >> 16) + ((uint32_t)(b.buf >> (tot - db)) & ((1u << db) - 1));
                    b.drop(tot);
                }
                if (dist > (uint32_t)(out - out_start)) return INF_BAD;
                uint8_t *const dst0 = out;
                out += len;
                const bool more = b.in_end - b.in >= 16 && out_end - out >= 280;
                if (more) {  // the next symbol's entry, in flight while the match is copied
                    b.refill_fast();
                    e = lt[b.buf & ((1u << LP) - 1)];
                }
                if (dist >= 8) {  // the usual case: eight bytes at a time, the first eight at once
                    const uint8_t *src = dst0 - dist;
                    uint64_t v;
                    std::memcpy(&v, src, 8);
                    std::memcpy(dst0, &v, 8);
                    if (len > 8) {
                        uint8_t *dst = dst0 + 8, *end = dst0 + len;
                        src += 8;
                        do {
                            std::memcpy(&v, src, 8);
                            std::memcpy(dst, &v, 8);
                            src += 8;
                            dst += 8;
                        } while (dst < end);
                    }
                } else
                    copy_match(dst0, dist, len);
                if (!more) break;
            }
        }
        // ---- careful: one symbol
        b.refill_safe();
        uint32_t e = lt[b.buf & ((1u << LP) - 1)];
        if ((e & EXC) && !(e & EOB)) e = lt[((e >> 8) & 0xFFFF) + ((b.buf >> LP) & ((1u << ((e >> 24) & 15)) - 1))];
        if (e & LIT) {
            const int n = 1 + (int)((e >> 30) & 1);
            if (out_end - out < n) {
                out_pos = out;
                return INF_OUT_FULL;
            }
            out[0] = (uint8_t)(e >> 8);
            if (n == 2) out[1] = (uint8_t)(e >> 16);
            out += n;
            b.drop((int)(e & 0xFF));
        } else if (e & EXC) {
            b.drop((int)(e & 0xFF));
            out_pos = out;
            return b.overrun * 8 > b.cnt ? INF_TRUNCATED : INF_OK;
        } else {
            if (!(e & 0xFF)) return INF_BAD;
            const int tot = (int)(e & 0xFF), xb = (int)((e >> 8) & 0xFF);
            const uint32_t len = (e >> 16) + ((uint32_t)(b.buf >> (tot - xb)) & ((1u << xb) - 1));
            b.drop(tot);
            b.refill_safe();
            uint32_t d = dt[b.buf & ((1u << DP) - 1)];
            if (d & 0x8000) d = dt[(d >> 16) + ((b.buf >> DP) & ((1u << ((d >> 8) & 0x7F)) - 1))];
            if (!(d & 0xFF)) return INF_BAD;
            const int dtot = (int)(d & 0xFF), db = (int)((d >> 8) & 0xFF);
            const uint32_t dist = (d >> 16) + ((uint32_t)(b.buf >> (dtot - db)) & ((1u << db) - 1));
            b.drop(dtot);
            if (dist > (uint32_t)(out - out_start)) return INF_BAD;
            if ((size_t)(out_end - out) < len) {
                out_pos = out;
                return INF_OUT_FULL;
            }
            for (uint32_t i = 0; i < len; i++) out[i] = out[(ptrdiff_t)i - (ptrdiff_t)dist];
            out += len;
        }
        if (b.overrun * 8 > b.cnt) return INF_TRUNCATED;  // bits that were never in the input have been consumed
    }
}

int decode_block_plain(Bits &b, const Tables &t, uint8_t *out_start, uint8_t *&out_pos, uint8_t *out_end) {
    return decode_block_body(b, t, out_start, out_pos, out_end);
}
// the same compiled with BMI2 (shifts by a register count without the flags / CL dependency)
__attribute__((target("bmi2"))) int decode_block_bmi2(Bits &b, const Tables &t, uint8_t *out_start, uint8_t *&out_pos, uint8_t *out_end) {
    return decode_block_body(b, t, out_start, out_pos, out_end);
}
using DecodeFn = int (*)(Bits &, const Tables &, uint8_t *, uint8_t *&, uint8_t *);
DecodeFn pick_decoder() { return __builtin_cpu_supports("bmi2") ? decode_block_bmi2 : decode_block_plain; }

// a raw DEFLATE stream from in[0 .. n_in) into out[0 .. cap): *consumed = bytes of input the stream took, *produced = bytes written
int inflate_raw(const uint8_t *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed, size_t *produced) {
    static const FixedTables fixed;
    static const DecodeFn decode_block = pick_decoder();
    static thread_local Tables dyn;
    Bits b;
    b.in = in;
    b.in_end = in + n_in;
    uint8_t *pos = out, *end = out + cap;
    for (;;) {
        b.refill_safe();
        const uint32_t last = b.peek(1), type = (b.peek(3) >> 1);
        b.drop(3);
        if (type == 0) {
            b.drop(b.cnt & 7);  // to the byte boundary
            // give the whole bytes still in the buffer back
            const int back = b.cnt >> 3;
            const uint8_t *p = b.in - (back - b.overrun > 0 ? back - b.overrun : 0);
            if (b.overrun > back) return INF_TRUNCATED;
            b.buf = 0;
            b.cnt = 0;
            b.overrun = 0;
            if (b.in_end - p < 4) return INF_TRUNCATED;
            const uint32_t len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
            if ((len ^ nlen) != 0xFFFF) return INF_BAD;
            p += 4;
            if ((size_t)(b.in_end - p) < len) return INF_TRUNCATED;
            if ((size_t)(end - pos) < len) return INF_OUT_FULL;
            std::memcpy(pos, p, len);
            pos += len;
            b.in = p + len;
        } else if (type == 1 || type == 2) {
            const Tables *t = &fixed.t;
            if (type == 2) {
                const int rc = read_dynamic_header(b, dyn);
                if (rc) return rc;
                t = &dyn;
            }
            const int rc = decode_block(b, *t, out, pos, end);
            if (rc) return rc;
        } else
            return INF_BAD;
        if (b.overrun * 8 > b.cnt) return INF_TRUNCATED;
        if (last) break;
    }
    // whole bytes left in the bit buffer belong to whatever follows the stream
    const int back = (b.cnt >> 3) - b.overrun;
    if (back < 0) return INF_TRUNCATED;
    *consumed = (size_t)(b.in - in) - (size_t)back;
    *produced = (size_t)(pos - out);
    return INF_OK;
}

// RFC 1952 member header -> offset of the DEFLATE data, 0 when malformed / truncated
size_t gzip_header(const uint8_t *p, size_t n) {
    if (n < 18 || p[0] != 31 || p[1] != 139 || p[2] != 8 || (p[3] & 0xE0)) return 0;
    const int flg = p[3];
    size_t at = 10;
    if (flg & 4) {
        if (at + 2 > n) return 0;
        at += 2 + (size_t)(p[at] | (p[at + 1] << 8));
    }
    for (int bit : {8, 16})
        if (flg & bit) {
            while (at < n && p[at]) at++;
            at++;
        }
    if (flg & 2) at += 2;
    return at + 8 <= n ? at : 0;
}

}  // namespace

namespace smi {

int host_gunzip(const uint8_t *in, size_t n_in, size_t *in_pos, uint8_t *out, size_t cap, size_t *out_pos) {
    size_t at = *in_pos, total = *out_pos;
    while (at < n_in) {
        const size_t h = gzip_header(in + at, n_in - at);
        if (!h) {
            // Behind a complete member: java.util.zip.GZIPInputStream (which the reference reads *.fastq.gz through) swallows a failed read of
            // the next header and reports end of stream, and gzip / zlib pass over padding the same way -- the stream ends here.
            if (at > 0) break;
            set_error("smi_gz_inflate: not a gzip stream");
            return SMI_ERR_INVALID;
        }
        size_t used = 0, made = 0;
        const int rc = inflate_raw(in + at + h, n_in - at - h, out + total, cap - total, &used, &made);
        if (rc == INF_OUT_FULL) {
            *in_pos = at;
            *out_pos = total;
            return 1;
        }
        if (rc) {
            set_error(rc == INF_TRUNCATED ? "smi_gz_inflate: truncated gzip stream" : "smi_gz_inflate: corrupt gzip data (invalid DEFLATE stream)");
            return SMI_ERR_INVALID;
        }
        const uint8_t *tr = in + at + h + used;
        if ((size_t)(in + n_in - tr) < 8) {
            set_error("smi_gz_inflate: truncated gzip stream");
            return SMI_ERR_INVALID;
        }
        const uint32_t crc = tr[0] | (tr[1] << 8) | (tr[2] << 16) | ((uint32_t)tr[3] << 24);
        const uint32_t isize = tr[4] | (tr[5] << 8) | (tr[6] << 16) | ((uint32_t)tr[7] << 24);
        if (isize != (uint32_t)made || crc != host_crc32(0, out + total, made)) {
            set_error("smi_gz_inflate: corrupt gzip data (CRC-32 or length of a member does not match)");
            return SMI_ERR_INVALID;
        }
        total += made;
        at += h + used + 8;
    }
    *in_pos = at;
    *out_pos = total;
    return SMI_OK;
}

// one raw DEFLATE stream that must fill `out` exactly and end with the input (a BGZF block's payload)
int host_inflate_exact(const uint8_t *in, size_t n_in, uint8_t *out, size_t n_out) {
    size_t used = 0, made = 0;
    const int rc = inflate_raw(in, n_in, out, n_out, &used, &made);
    return rc == INF_OK && made == n_out ? 0 : 1;
}

}  // namespace smi
