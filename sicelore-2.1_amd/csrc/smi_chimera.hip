// smi_chimera.hip -- K-PACKR + K-CHIM: the reference's chimera splitter (pass 2, 3' barcoding) on gfx950.
//
// Reference units (bytecode; citation form in DESIGN.md):
//   ChimeraFindernew.findSplitPositions                FJ!nanoporereadscanner/analyzers/ChimeraFindernew.java:L107-332
//   PolyATadapterInternalSearcherBase.aTscan / searchATend / adapterScan
//                                                      FJ!nanopore/analyzers/PolyATadapterInternalSearcherBase.java:L78-270
//   AdapterTSOanalyzer.scanForAdapterOrTSOseqKMERsForInternal  FJ!nanopore/analyzers/AdapterTSOanalyzer.java:L130-156
//   $AdapterScanRslt.getPosbelowMaxMismatches / getPosForBestScore  (same file, L278-308)
//
// MI355X mapping.  K-PACKR turns every read into four IUPAC bit-planes (A, G, C, T bits of the 4-bit code), one
// wavefront per read.  K-CHIM runs one wavefront per read on those planes:
//   * internal TSO scan (both orientations): the 4-mer gate for 64 positions per lane (bit-parallel), then one
//     27 x 27 Needleman-Wunsch per candidate with lane = candidate (error count only, smi_nw.h); the reference's
//     position-skip rule is a scalar fold over the candidates in order
//   * internal polyA / polyT windows: bit-sliced 14-base counters give the trigger positions for 64 positions per
//     lane; the few triggers are walked in order (the walk is data-dependent), each followed by the 51-base
//     adapter scan (gate in one register, 22 x 22 alignments with lane = position)
//   * the split rules run on the handful of matches in LDS.
// Integer / bitwise work: no MFMA.  Planes are read from global memory (they are touched a few times, L2-resident),
// so the read length is not limited by LDS.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "smi_internal.h"
#include "smi_nw.h"

namespace smi {

// The two pattern lengths are template parameters <TL, AL> of the kernel: <27, 22> for 3' barcoding (complete TSO /
// complete adapter, Jar/config.xml:170,113) and <22, 25> for 5' barcoding (5' adapter / 3' adapter, config.xml:126,141).
constexpr int kMaxPat = 27;
constexpr int kCap = 64;     // accepted TSO positions per orientation / matches per read kept in LDS
constexpr int kPadWords = kReadPadWords; // plane spacing: ceil(len / 32) data words + 4 zero words (gates and windows run past the end)

struct ChimParams {
    uint32_t tso4[2][kMaxPat];  // [0] the pattern searched in both orientations (complete TSO), [1] its reverse complement
    uint32_t ad4[kMaxPat];      // the adapter searched next to internal polyA / polyT
    int tso_max, ad_max;
    int pat_len;   // internalpATlength (15)
    int pat_thr;   // least count c with c / (float)pat_len >= internalFractionATInPolyAT
    int off;       // windowSearchForPolyA + 70
    int bc_umi;    // cell barcode + UMI length
    // exact pre-filter of the alignments (myers_bound below): plane index (A 0, G 1, C 2, T 3) of the pattern bases in the order the
    // bit-vector recurrence consumes them; prefilter = 0 when a pattern holds an IUPAC code other than A, G, C, T
    int prefilter;   // K-CHIM-A runs first and leaves its verdicts in out[r].n_matches
    int myers_ok;    // every pattern base is A / G / C / T: the bit-vector bound applies
    int ablate;    // measurement builds only (SMI_CHIM_ABLATE): 1 no TSO scan, 2 no polyA/T scan, 4 pre-filter without the bit-vector pass, 8 gates only
    int tso_lead_max, ad_lead_max;       // most leading template gaps an accepted alignment can have (1.1 * lead <= max errors)
    uint8_t tso_fwd_idx[kMaxPat];        // orientation 0: TSO[m-1-j]
    uint8_t tso_rev_idx[kMaxPat];        // orientation 1: complement of TSO[j] on the bit-reversed planes
    uint8_t ad_idx[kMaxPat];             // adapter[m-1-j]
};

// planes of one read: word w of plane c at p[c][w]
struct ReadPlanes {
    const uint32_t *p[4];
};


__device__ __forceinline__ uint64_t gget64(const uint32_t *pl, int bitpos) {
    const int w = bitpos >> 5, s = bitpos & 31;
    const uint64_t lo = ((uint64_t)pl[w + 1] << 32) | pl[w];
    uint64_t r = lo >> s;
    if (s) r |= (uint64_t)pl[w + 2] << (64 - s);
    return r;
}
__device__ __forceinline__ uint32_t gget32(const uint32_t *pl, int bitpos) {
    const int w = bitpos >> 5, s = bitpos & 31;
    return (uint32_t)((((uint64_t)pl[w + 1] << 32) | pl[w]) >> s);
}
__device__ __forceinline__ uint64_t gmatch64(const ReadPlanes &rp, uint32_t a4, int bitpos) {
    uint64_t m = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if ((a4 >> c) & 1u) m |= gget64(rp.p[c], bitpos);
    return m;
}
__device__ __forceinline__ uint32_t gmatch32(const ReadPlanes &rp, uint32_t a4, int bitpos) {
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if ((a4 >> c) & 1u) m |= gget32(rp.p[c], bitpos);
    return m;
}
// exact A (code 1) / exact T (code 8): the packer emits A, G, C, T, N (all four bits) and nothing else
__device__ __forceinline__ uint64_t gexact64(const ReadPlanes &rp, int is_t, int bitpos) {
    return is_t ? (gget64(rp.p[3], bitpos) & ~gget64(rp.p[0], bitpos)) : (gget64(rp.p[0], bitpos) & ~gget64(rp.p[1], bitpos));
}
__device__ __forceinline__ uint32_t gexact_bit(const ReadPlanes &rp, int is_t, int idx) {
    const int w = idx >> 5, s = idx & 31;
    const uint32_t x = is_t ? (rp.p[3][w] & ~rp.p[0][w]) : (rp.p[0][w] & ~rp.p[1][w]);
    return (x >> s) & 1u;
}
__device__ __forceinline__ uint32_t comp4(uint32_t b) {
    return ((b & 1u) << 3) | ((b & 8u) >> 3) | ((b & 2u) << 1) | ((b & 4u) >> 1);
}
__device__ __forceinline__ uint64_t keep_low64(uint64_t m, int n_bits) {
    return n_bits <= 0 ? 0ull : (n_bits >= 64 ? m : (m & ((1ull << n_bits) - 1ull)));
}
__device__ __forceinline__ int kth_bit64(uint64_t m, int k) {
    for (; k > 0; k--) m &= m - 1;
    return __builtin_ctzll(m);
}
__device__ __forceinline__ int wave_exscan_i(int v, int lane, int &total) {
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(inc, o);
        if (lane >= o) inc += y;
    }
    total = __shfl(inc, 63);
    return inc - v;
}
__device__ __forceinline__ int jround(float a) { return (int)floorf(__fadd_rn(a, 0.5f)); }  // Math.round(float)

// ---------------------------------------------------------------------------------------------------------------
// K-PACKR: ASCII reads -> bit-planes.  planes: [4][stride] u32; read r starts at word plane_start(offsets[r], r).
// ---------------------------------------------------------------------------------------------------------------
// one wave per read, one lane per 32-base plane word: a wave reads 2 KiB of consecutive ASCII and writes four 256-B rows.  The offsets of
// the read after the next and the text of the next read are requested before the current one is encoded (the loop is otherwise three
// dependent round trips per read: offsets, text, stores).
// kStarts: the bases sit in the FASTQ text (starts[r]), not in a gathered copy (offsets[r]).  A template parameter, not a test of the pointer:
// as `starts ? starts[r] : beg` the source position depended on a value just asked for, and the wait the compiler put in front of that
// select sat in the middle of the loop -- behind the text loads of the next read, which were meant to be in flight while this one is encoded.
template <bool kStarts>
__global__ __launch_bounds__(256) void k_pack_reads(const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets,
                                                    const uint64_t *__restrict__ starts, size_t n, size_t stride,
                                                    uint64_t total_bases, uint32_t *__restrict__ planes) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    struct Meta {
        uint64_t beg, end, src;
    };
    auto load_meta = [&](size_t r) -> Meta {
        Meta m{0, 0, 0};
        if (r < n) {
            m.beg = offsets[r];
            m.end = offsets[r + 1];
            m.src = kStarts ? starts[r] : m.beg;
        }
        return m;
    };
    // the 32 bases of word w: two 16-byte loads (a wave reads 2 KiB of consecutive text); the partial last word byte by byte
    auto load_word = [&](const Meta &m, int64_t w, uint32_t (&v)[8]) {
        const int64_t len = (int64_t)(m.end - m.beg), p0 = 32 * w;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = 0;
        // The last, partial word of a read takes the same two 16-byte loads when the bytes behind the read's end are known to exist:
        // in the FASTQ text the line end, the '+' line and the read's own quality line follow (>= len + 3 bytes), in a gathered array
        // the next reads do (up to total_bases).  encode_store drops the codes behind the end.  (A byte loop here -- counted or unrolled
        // under `if (i < nb)` -- is compiled into dependent round trips and ~300 wave instructions per read on ONE lane, as much as the
        // encoding of the whole read: 587 M instead of ~300 M VALU instructions per 0.9 M reads.)
        const bool wide = p0 + 32 <= len || (p0 < len && (kStarts ? len >= 29 : m.beg + (uint64_t)p0 + 32 <= total_bases));
        if (wide) {
            __builtin_memcpy(v, reads + m.src + p0, 32);
        } else {
            for (int i = 0; i < (int)(len - p0); i++) {  // reads shorter than 29 bases, the last read of a gathered array
                const uint32_t b = (uint32_t)reads[m.src + p0 + i] << (8 * (i & 3));
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] |= (i >> 2) == j ? b : 0u;
            }
        }
    };
    auto encode_store = [&](const Meta &m, size_t r, int64_t w, const uint32_t (&v)[8]) {
        const int64_t len = (int64_t)(m.end - m.beg), p0 = 32 * w;
        uint32_t pl[4] = {0, 0, 0, 0};
        if (p0 < len) {
            const int nb = (int)min((int64_t)32, len - p0);
#pragma unroll
            for (int j = 0; j < 8; j++) {  // four bases per enc4x4
                uint32_t code = enc4x4<false>(v[j]);
                if (4 * j + 4 > nb) code &= 4 * j >= nb ? 0u : (0xFFFFFFFFu >> (8 * (4 * j + 4 - nb)));  // nothing behind the read's end
#pragma unroll
                for (int c = 0; c < 4; c++) pl[c] |= plane_nibble(code, c) << (4 * j);
            }
        }
        const size_t w0 = plane_start(m.beg, r);
#pragma unroll
        for (int c = 0; c < 4; c++) planes[c * stride + w0 + w] = pl[c];
    };
    Meta cur = load_meta(wave), nxt = load_meta(wave + n_waves);
    uint32_t vcur[8];
    load_word(cur, lane, vcur);
    for (size_t r = wave; r < n; r += n_waves) {
        uint32_t vnxt[8];
        load_word(nxt, lane, vnxt);                       // (an empty Meta behind the last read: nothing is loaded)
        const Meta nxt2 = load_meta(r + 2 * n_waves);
        const int64_t n_words = ((int64_t)(cur.end - cur.beg) + 31) / 32 + kPadWords - 1;  // the pad words are written as zeros
        if (lane < n_words) encode_store(cur, r, lane, vcur);
        for (int64_t w = lane + 64; w < n_words; w += 64) {  // reads beyond 2048 bases
            uint32_t v[8];
            load_word(cur, w, v);
            encode_store(cur, r, w, v);
        }
        cur = nxt;
        nxt = nxt2;
#pragma unroll
        for (int j = 0; j < 8; j++) vcur[j] = vnxt[j];
    }
}

// K-PACKR, flat (round 5): a thread per PLANE WORD instead of a wave per read.  The planes tile the word axis (read r owns words plane_start(offsets[r], r) ..
// plane_start(offsets[r + 1], r + 1): its data words, then four or five zero words), so word g belongs to the last read whose start is <= g: a binary search over
// the offsets (once per wave, for its first word; the lanes step on from there), then the word's 32 characters as two 16-byte loads, the encoder, four stores.
// Every lane has a word whatever the read lengths are -- a wave per read kept 42 of 64 lanes busy on 1,300-base reads and paid a round trip per read for its
// offsets.  The words between a read's data and the next read are written as zeros (all of them; the wave kernel left the fifth one alone, nobody reads it).
template <bool kStarts>
__global__ __launch_bounds__(256) void k_pack_reads_flat(const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets,
                                                         const uint64_t *__restrict__ starts, size_t n, size_t stride, uint64_t total_bases,
                                                         uint32_t *__restrict__ planes) {
    const size_t n_words_all = plane_start(total_bases, n);  // behind the last read's words: not written (as before)
    // every wave takes a CONTIGUOUS run of 64-word tiles: one binary search for its first word (uniform: scalar loads), then the lanes step from read to read as
    // their words move on (a tile spans one or two reads of ordinary length, a dozen of the shortest) -- no search, and no far jump, per tile
    const size_t n_tiles = (n_words_all + 63) / 64;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t per = (n_tiles + n_waves - 1) / n_waves;
    const size_t t_begin = wave * per, t_end = min(n_tiles, t_begin + per);
    if (t_begin >= t_end) return;
    size_t r;
    {
        const size_t g0 = t_begin * 64;
        size_t lo = 0, hi = n;
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) >> 1;
            if (plane_start(offsets[mid], mid) <= g0)
                lo = mid;
            else
                hi = mid;
        }
        r = lo;
    }
    for (size_t t = t_begin; t < t_end; t++) {
        const size_t g = t * 64 + (threadIdx.x & 63);
        if (g >= n_words_all) break;
        while (r + 1 < n && plane_start(offsets[r + 1], r + 1) <= g) r++;
        const uint64_t beg = offsets[r], end = offsets[r + 1];
        const int64_t len = (int64_t)(end - beg), w = (int64_t)g - (int64_t)plane_start(beg, r), p0 = 32 * w;
        if (w < 0) continue;  // (offsets[0] > 31: words in front of the first read belong to nobody)
        uint32_t pl[4] = {0, 0, 0, 0};
        if (p0 < len) {
            const uint64_t src = (kStarts ? starts[r] : beg) + (uint64_t)p0;
            uint32_t v[8];
            // the partial last word takes the same two loads when the bytes behind the read's end exist (k_pack_reads has the argument)
            const bool wide = p0 + 32 <= len || (kStarts ? len >= 29 : beg + (uint64_t)p0 + 32 <= total_bases);
            if (wide) {
                __builtin_memcpy(v, reads + src, 32);
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = 0;
                for (int i = 0; i < (int)(len - p0); i++) {
                    const uint32_t b = (uint32_t)reads[src + i] << (8 * (i & 3));
#pragma unroll
                    for (int k = 0; k < 8; k++) v[k] |= (i >> 2) == k ? b : 0u;
                }
            }
            const int nb = (int)min((int64_t)32, len - p0);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t code = enc4x4<false>(v[k]);
                if (4 * k + 4 > nb) code &= 4 * k >= nb ? 0u : (0xFFFFFFFFu >> (8 * (4 * k + 4 - nb)));
#pragma unroll
                for (int c = 0; c < 4; c++) pl[c] |= plane_nibble(code, c) << (4 * k);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) planes[c * stride + g] = pl[c];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K-CHIM
// ---------------------------------------------------------------------------------------------------------------
// 128-bit window of each base plane starting at bit b: everything the gates of 64 scan positions need
struct PlaneWin {
    uint64_t lo[4], hi[4];
};
__device__ __forceinline__ PlaneWin load_window(const ReadPlanes &rp, int b) {
    PlaneWin w;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        w.lo[c] = gget64(rp.p[c], b);
        w.hi[c] = gget64(rp.p[c], b + 64);
    }
    return w;
}
// match bits of IUPAC code a4 at scan positions b+i .. b+i+63 (i < 64, compile-time)
__device__ __forceinline__ uint64_t win_match(const PlaneWin &w, uint32_t a4, int i) {
    uint64_t m = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if ((a4 >> c) & 1u) m |= i == 0 ? w.lo[c] : ((w.lo[c] >> i) | (w.hi[c] << (64 - i)));
    return m;
}
// 4-mer gate (Kmers.nKmersMatching >= 2) for 64 scan positions
template <int N>
__device__ __forceinline__ uint64_t gate_two(const PlaneWin &w, const uint32_t *code) {
    uint64_t any = 0, two = 0;
    uint64_t m0 = win_match(w, code[0], 0), m1 = win_match(w, code[1], 1), m2 = win_match(w, code[2], 2);
#pragma unroll
    for (int i = 0; i + 3 < N; i++) {
        const uint64_t m3 = win_match(w, code[i + 3], i + 3);
        const uint64_t k = m0 & m1 & m2 & m3;
        two |= any & k;
        any |= k;
        m0 = m1;
        m1 = m2;
        m2 = m3;
    }
    return two;
}

// ---------------------------------------------------------------------------------------------------------------
// Exact pre-filter of the Needleman-Wunsch acceptance tests.
//
// Both internal scans accept a position only when nErrors = #x - 0.9 * lead is at most max (AdapterTSOanalyzer.java:L143-144,
// L302), #x = alignment columns that are not a match, lead = leading columns that hold a template gap (Match.java:L31-34).
// Dropping the lead columns from the traceback leaves an alignment of the WHOLE pattern against the read slice without its
// first `lead` bases that has #x - lead <= nErrors edit operations, so
//        min over s <= lead_max of  Levenshtein(pattern, slice[s ..])  <=  nErrors,
// and lead <= nErrors / 1.1 because every leading template gap is paid back by a read gap further on (both sequences are
// m long).  A position whose bound exceeds max can therefore never be accepted -- whatever the position-skip rule lets the
// scan visit -- and a read (or an internal polyA/T site) without any position under the bound has NO accepted match: the
// alignments, whose exact error counts only matter for the skip distances in front of an accepted position, need not run.
//
// The bound is Myers' bit-vector edit distance (Hyyro's global form) with the roles swapped: the bit-vector runs over the
// read slice REVERSED (prefixes of it = suffixes of the slice), the pattern is consumed base by base from its end, and the
// match vector of a pattern base is simply the read's bit-plane of that base (an N in the read sets all four planes and
// matches everything, as (a & b) != 0 does).  The last column's vertical deltas give D[i][m] for every prefix length i.
// Sixteen instructions per pattern base, all of the 2-cycle forms (and / or / xor / add / not on VGPRs; written as one asm
// block so that the compiler does not fuse them into three-operand forms, which issue at half the rate and drag their
// neighbours along: profiles/r02/valu_peak.json).
// ---------------------------------------------------------------------------------------------------------------
// (SMI_MYERS_STEP: smi_nw.h)

// V[b]: plane b of the read slice, bit-reversed into the low M bits (bit k = slice base M-1-k).  idx[j] (wave-uniform): plane of the
// j-th pattern base consumed.  -> min over i in [M - lead_max, M] of D[i][M]
template <int M>
__device__ __forceinline__ int myers_bound(const uint32_t (&V)[4], const uint8_t *idx, int lead_max) {
    uint32_t pv = 0xFFFFFFFFu, mv = 0u, t_xv, t_a, t_b;
#pragma unroll
    for (int j = 0; j < M; j++) {
        switch (idx[j]) {  // uniform: a scalar branch
        case 0: SMI_MYERS_STEP(V[0]); break;
        case 1: SMI_MYERS_STEP(V[1]); break;
        case 2: SMI_MYERS_STEP(V[2]); break;
        default: SMI_MYERS_STEP(V[3]); break;
        }
    }
    const uint32_t mask = (1u << M) - 1u;
    int d = M + __popc(pv & mask) - __popc(mv & mask);  // D[M][M]
    int best = d;
    for (int k = M - 1; k >= M - lead_max && k >= 0; k--) {  // D[k][M] = D[k+1][M] - (Pv - Mv)(row k+1 = bit k)
        d -= (int)((pv >> k) & 1u) - (int)((mv >> k) & 1u);
        best = min(best, d);
    }
    return best;
}

// The shipped patterns as compile-time constants (Jar/config.xml:170 complete TSO for 3' barcoding; :126 the 5' adapter for 5'
// barcoding): with them the gates and the bit-vector bound of K-CHIM-A need no pattern look-ups at all.  PAT 0 = any pattern, read
// from the kernel arguments.
template <int PAT>
struct PatSeq;
template <>
struct PatSeq<1> {
    static constexpr const char *s = "AAGCAGTGGTATCAACGCAGAGTACAT";
};
template <>
struct PatSeq<2> {
    static constexpr const char *s = "CTACACGACGCTCTTCCGATCT";
};
__host__ __device__ constexpr int plane_c(char c) { return c == 'A' ? 0 : c == 'G' ? 1 : c == 'C' ? 2 : 3; }
// plane of base j of the pattern searched in orientation o (1: the reverse complement)
template <int PAT, int M>
__host__ __device__ constexpr int gate_plane(int o, int j) {
    return o == 0 ? plane_c(PatSeq<PAT>::s[j]) : 3 - plane_c(PatSeq<PAT>::s[M - 1 - j]);
}
// plane consumed at step j of myers_bound (see ChimParams::tso_fwd_idx / tso_rev_idx)
template <int PAT, int M>
__host__ __device__ constexpr int myers_plane(int o, int j) {
    return o == 0 ? plane_c(PatSeq<PAT>::s[M - 1 - j]) : 3 - plane_c(PatSeq<PAT>::s[j]);
}

template <int M, int PAT, int ORI>
__device__ __forceinline__ int myers_bound_ct(const uint32_t (&V)[4], int lead_max) {
    uint32_t pv = 0xFFFFFFFFu, mv = 0u, t_xv, t_a, t_b;
#pragma unroll
    for (int j = 0; j < M; j++) {
        constexpr int dummy = 0;
        (void)dummy;
        const uint32_t eqv = V[(myers_plane<PAT, M>(ORI, j))];
        SMI_MYERS_STEP(eqv);
    }
    const uint32_t mask = (1u << M) - 1u;
    int d = M + __popc(pv & mask) - __popc(mv & mask);
    int best = d;
    for (int k = M - 1; k >= M - lead_max && k >= 0; k--) {
        d -= (int)((pv >> k) & 1u) - (int)((mv >> k) & 1u);
        best = min(best, d);
    }
    return best;
}

struct MatchRec {  // ChimeraFindernew$AdapterTSOmatch
    int begin;
    int is_reverse;
    int is_adapter;
};

// per-wave LDS
struct WaveLds {
    unsigned long long cmask[2][64];  // candidate bits of the current segment, per orientation, one word per lane
    int coff[2][64];                  // exclusive prefix of the candidate counts (orientation 1 continues orientation 0)
    int acc_pos[2][kCap];             // accepted TSO positions per orientation ...
    float acc_ne[2][kCap];            // ... and their error counts
    int srt_pos[kCap];
    float srt_ne[kCap];
    int m_begin[kCap];             // matches entering the split rules
    int m_kind[kCap];              // bit 0 is_reverse, bit 1 is_adapter
    int order[kCap];
};

// searchATend (PolyATadapterInternalSearcherBase.java:L233-270); every lane runs the same walk.  The exact-base bits are
// read through a 128-bit register window that is refilled every 48 positions, not bit by bit from memory.
__device__ __forceinline__ int search_at_end(const ReadPlanes &rp, int len, int pos, int is_t, int cur, const ChimParams &P) {
    const int ml = P.pat_len;  // <= 15
    const int lim = len - P.off - ml - 1;
    int pb = pos + 1;
    bool stop = !(pb < lim && cur >= P.pat_thr);
    while (!stop) {
        const uint64_t w0 = gexact64(rp, is_t, pb), w1 = gexact64(rp, is_t, pb + 64);  // bits pb .. pb+127
        for (int j = 0; j < 48; j++, pb++) {
            if (!(pb < lim && cur >= P.pat_thr)) {
                stop = true;
                break;
            }
            auto bit = [&](int k) -> int { return (int)((k < 64 ? (w0 >> k) : (w1 >> (k - 64))) & 1ull); };  // base at pb - j + k
            cur -= bit(j);
            cur += bit(j + ml);
            if (cur >= P.pat_thr) pos = pb;
            if (!bit(j + ml - 1) && !bit(j + ml - 2)) {
                stop = true;
                break;
            }
        }
    }
    int end = pos + ml - 1;
    for (;;) {
        const uint32_t x = (uint32_t)(gexact64(rp, is_t, end - 3) & 15ull);  // bases end-3 .. end
        if (__popc(x) >= 2) break;
        end -= 4;
    }
    while (!gexact_bit(rp, is_t, end)) end--;
    return end;
}

// adapterScan (L159-221): read coordinate of the first accepted adapter match next to an internal polyA/T, 0 = none
template <int kAdLen>
__device__ __forceinline__ int adapter_scan(const ReadPlanes &rp, int at_begin, int at_end, int is_t, int lane,
                                            const ChimParams &P) {
    int start_range, end_range;
    if (is_t) {
        start_range = at_begin - P.bc_umi - 30 - 10;
        end_range = start_range + 30 + 20;
    } else {
        end_range = at_end + P.bc_umi + 30 + 10;
        start_range = end_range - 30 - 20;
    }
    constexpr int NPOS = 51 - kAdLen;  // scan positions 1 .. 29 (bits 0 .. 28)
    // match bits of adapter base i against the sub-sequence (reverse-complemented when the stretch is polyA)
    auto col_bits = [&](int i, int shift) -> uint32_t {
        // bit k = sub[shift + k] matches adapter base i
        if (is_t) return gmatch32(rp, P.ad4[i], start_range - 1 + shift);
        // sub_rc[x] = comp(seq[end_range - 1 - x]): 32 read bases ending at end_range - 1 - shift, bit-reversed
        return __brev(gmatch32(rp, comp4(P.ad4[i]), end_range - 32 - shift));
    };
    uint32_t any = 0, two = 0, three = 0;
    {
        uint32_t m0 = col_bits(0, 0), m1 = col_bits(1, 1), m2 = col_bits(2, 2);
#pragma unroll
        for (int i = 0; i + 3 < kAdLen; i++) {
            const uint32_t m3 = col_bits(i + 3, i + 3);
            const uint32_t k = m0 & m1 & m2 & m3;
            three |= two & k;
            two |= any & k;
            any |= k;
            m0 = m1;
            m1 = m2;
            m2 = m3;
        }
    }
    const uint32_t gate = three & ((1u << NPOS) - 1u);  // minKmersMatching = 3 (L173)
    if (gate == 0) return 0;
    if (P.myers_ok) {
        // no scan position with an alignment that could be accepted (round(nErrors) <= max => Levenshtein bound <= max): the
        // result list is empty whatever the skip rule visits (myers_bound)
        bool hot = false;
        if (lane < NPOS && ((gate >> lane) & 1u)) {
            uint32_t V[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                // plane c of the sub-sequence at scan position lane + 1, bit k = sub[lane + k]
                const uint32_t w = is_t ? gget32(rp.p[c], start_range - 1 + lane) : __brev(gget32(rp.p[3 - c], end_range - 32 - lane));
                V[c] = __brev(w) >> (32 - kAdLen);
            }
            hot = myers_bound<kAdLen>(V, P.ad_idx, P.ad_lead_max) <= P.ad_max;
        }
        if (!__ballot(hot)) return 0;
    }
    // lane l < 29 aligns scan position l + 1
    float ne = 0.0f;
    int nmis = 0;
    if (lane < NPOS && ((gate >> lane) & 1u)) {
        uint32_t col[kAdLen];
#pragma unroll
        for (int c = 0; c < kAdLen; c++) col[c] = col_bits(c, lane) & ((1u << kAdLen) - 1u);
        AlnStats st;
        nw_full<kAdLen, false, false, nw_band<kAdLen, 6>()>(col, 0, st);  // only ne and nmis are read; the gate asked for 3 matching 4-mers = 6 diagonal matches
        ne = st.ne;
        nmis = st.nmis;
    }
    const float maxe = (float)P.ad_max;
    const int ok_l = !((float)jround(ne) > maxe) ? 1 : 0;
    int delta_l = 1;
    if (maxe < ne) {
        delta_l = jround(__fsub_rn(ne, maxe)) - 1;
        if (delta_l < 1) delta_l = 1;
    }
    // scanForAdapterOrTSOseqKMERsForInternal's position skip: scalar fold in scan order
    uint32_t accepted = 0;
    int skip = 0;
    for (uint32_t g = gate; g; g &= g - 1) {
        const int i = __builtin_ctz(g);
        if (i + 1 < skip) continue;
        if (__builtin_amdgcn_readlane(ok_l, i)) accepted |= 1u << i;
        skip = i + 1 + __builtin_amdgcn_readlane(delta_l, i);
    }
    if (!accepted) return 0;
    // getPosForBestScore(MAX_VALUE): positions sharing the least key, neighbours < 2 apart dropped (L180-183)
    const bool mine = lane < NPOS && ((accepted >> lane) & 1u);
    float best = mine ? ne : 3.4028234663852886e+38f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = fminf(best, __shfl_xor(best, o));
    const uint32_t eq = (uint32_t)__ballot(mine && ne == best);
    const uint32_t keep = eq & ~(eq << 1);
    const uint32_t good = keep & (uint32_t)__ballot(nmis <= P.ad_max);  // L203
    if (!good) return 0;
    const int o1 = __builtin_ctz(good) + 1;
    return is_t ? start_range + o1 - 1 : start_range + 51 - o1;  // L207 / L211
}

// ---------------------------------------------------------------------------------------------------------------
// K-CHIM-A: the pre-filter pass of the internal TSO scan (myers_bound).  One wave per read, one lane per 32 scan positions:
// the 4-mer gates of both orientations in 32-bit words (every lane busy for reads of >= 2 kb, 41 of 64 for 1.3 kb), the
// gate-passing positions pooled per orientation through LDS and bounded 64 at a time.  Writes hot[r] = 1 when some position
// could be accepted.  No Needleman-Wunsch here: ~1,300 instructions per read instead of ~17,000 per candidate batch.
// ---------------------------------------------------------------------------------------------------------------
struct FilterParams {  // what K-CHIM-A needs of ChimParams, bytes instead of words (kernel arguments live in SGPRs)
    uint8_t gate_idx[2][kMaxPat];  // plane (A 0, G 1, C 2, T 3) of pattern base j, per orientation
    uint8_t fwd_idx[kMaxPat], rev_idx[kMaxPat];
    uint8_t tso4[2][kMaxPat];      // 4-bit codes of the pattern, per orientation (gates)
    int tso_max, lead_max;
    int myers_ok;                  // 0: a pattern base is not A / G / C / T -> every read counts as possible
    int force_all;                 // measurement / cross-check: every read gets both verdicts (the kernels behind run on all reads)
    int pat_len, pat_thr, off;     // internal polyA / polyT windows (aTscan)
};
#define SMI_CHIMA_TSO 1  /* K-CHIM-A verdicts left in out[r].n_matches: some position may hold an accepted internal TSO alignment */
#define SMI_CHIMA_PAT 2  /* some position satisfies the local start condition of an internal polyA / polyT stretch */

struct FilterLds {
    uint32_t cmask[2][64];
    int coff[2][64];
};

__device__ __forceinline__ int kth_bit32(uint32_t m, int k) {
    for (; k > 0; k--) m &= m - 1;
    return __builtin_ctz(m);
}

template <int kTsoLen, int PAT>
__global__ __launch_bounds__(256) void k_chim_tso_filter(const uint32_t *__restrict__ planes, size_t stride, const uint32_t *__restrict__ pstart,
                                                         const uint64_t *__restrict__ offsets, size_t n, FilterParams P,
                                                         smi_chimera_result *__restrict__ out, uint32_t *__restrict__ list,
                                                         uint32_t *__restrict__ list_count) {
    __shared__ FilterLds lds_all[4];
    FilterLds &L = lds_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < n; r += n_waves) {
        const uint64_t beg = offsets[r];
        const int len = (int)(offsets[r + 1] - beg);
        bool any_hot = false, any_pat = false;
        if (len >= 2 * 70 + 100) {
            ReadPlanes rp;
            const size_t w0 = pstart ? (size_t)pstart[r] : plane_start(beg, r);  // pstart: explicit word start per read (segmented host packer)
#pragma unroll
            for (int c = 0; c < 4; c++) rp.p[c] = planes + c * stride + w0;
            const int last = len - 70;
            const int pa_first = P.off - 1, pa_stop = len - P.off;  // aTscan positions (0-based) [pa_first, pa_stop)
            int extra[2] = {0, 0};
            if (pa_first < pa_stop && P.pat_thr <= 15) {
                extra[0] = (int)gexact_bit(rp, 0, P.off + P.pat_len - 2);
                extra[1] = (int)gexact_bit(rp, 1, P.off + P.pat_len - 2);
            } else
                any_pat = pa_first < pa_stop;  // thresholds the bit-sliced counter cannot hold: leave it to the big kernel
            if (!P.myers_ok) any_hot = true;
            if (P.force_all) any_hot = any_pat = true;
            for (int p0 = 70; p0 <= last && !(any_hot && any_pat); p0 += 2048) {
                const int pl = p0 + 32 * lane;  // first scan position of this lane (1-based); its bit is pl - 1
                uint32_t cm[2] = {0, 0};
                uint64_t Wk[4] = {0, 0, 0, 0};
                if (pl <= last) {
#pragma unroll
                    for (int c = 0; c < 4; c++) Wk[c] = gget64(rp.p[c], pl - 1);
                    const uint64_t(&W)[4] = Wk;
#pragma unroll
                    for (int o = 0; o < 2 && P.myers_ok; o++) {
                        // m_j = match bits of pattern base j at scan positions pl .. pl+31; a 4-mer at i matches where m_i & m_i+1 & m_i+2 & m_i+3
                        auto mj = [&](int j) -> uint32_t {
                            if constexpr (PAT != 0) {
                                return (uint32_t)(W[gate_plane<PAT, kTsoLen>(o, j)] >> j);
                            } else {
                                const uint32_t a4 = P.tso4[o][j];
                                uint64_t x = 0;
#pragma unroll
                                for (int c = 0; c < 4; c++)
                                    if ((a4 >> c) & 1u) x |= W[c];
                                return (uint32_t)(x >> j);
                            }
                        };
                        uint32_t any = 0, two = 0;
                        const uint32_t m0 = mj(0), m1 = mj(1);
                        uint32_t m_prev = mj(2);
                        uint32_t p01 = m0 & m1, p12 = m1 & m_prev;
#pragma unroll
                        for (int i = 0; i + 3 < kTsoLen; i++) {
                            const uint32_t m3 = mj(i + 3);
                            const uint32_t p23 = m_prev & m3;
                            const uint32_t k = p01 & p23;
                            two |= any & k;
                            any |= k;
                            p01 = p12;
                            p12 = p23;
                            m_prev = m3;
                        }
                        const int nb = last - pl + 1;
                        cm[o] = nb >= 32 ? two : (two & ((1u << nb) - 1u));
                    }
                }
                // internal polyA / polyT: is the local start condition of aTscan (window count >= threshold, bases pos and pos + 1 exact)
                // met anywhere?  (The big kernel walks the stretches in order; a read without any such position has none.)
                if (!any_pat) {
                    uint32_t trig = 0;
                    if (pl <= last) {
#pragma unroll
                        for (int t = 0; t < 2; t++) {
                            const uint64_t e = t ? (Wk[3] & ~Wk[0]) : (Wk[0] & ~Wk[1]);  // exact T / exact A from bit pl - 1
                            uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
                            for (int k = 1; k < 15; k++) {
                                if (k >= P.pat_len) break;
                                const uint32_t x = (uint32_t)(e >> k);
                                const uint32_t t0 = c0 & x;
                                c0 ^= x;
                                const uint32_t t1 = c1 & t0;
                                c1 ^= t0;
                                const uint32_t t2 = c2 & t1;
                                c2 ^= t1;
                                c3 ^= t2;
                            }
                            if (extra[t]) {
                                const uint32_t t0 = c0;
                                c0 = ~c0;
                                const uint32_t t1 = c1 & t0;
                                c1 ^= t0;
                                const uint32_t t2 = c2 & t1;
                                c2 ^= t1;
                                c3 ^= t2;
                            }
                            const uint32_t cb[4] = {c0, c1, c2, c3};
                            uint32_t gt = 0, eq = ~0u;
#pragma unroll
                            for (int b = 3; b >= 0; b--) {
                                if ((P.pat_thr >> b) & 1)
                                    eq &= cb[b];
                                else
                                    gt |= eq & cb[b];
                            }
                            // positions q = pl - 1 + k (0-based) with first <= q < stop
                            uint32_t m = (gt | eq) & (uint32_t)e & (uint32_t)(e >> 1);
                            const int q0 = pl - 1;
                            if (q0 < pa_first) m = pa_first - q0 >= 32 ? 0u : (m & (~0u << (pa_first - q0)));
                            if (q0 + 32 > pa_stop) m = pa_stop - q0 <= 0 ? 0u : (m & ((1u << (pa_stop - q0)) - 1u));
                            trig |= m;
                        }
                    }
                    if (__ballot(trig != 0)) any_pat = true;
                }
#pragma unroll 1
                for (int o = 0; o < 2 && !any_hot; o++) {
                    int total;
                    const int off = wave_exscan_i(__popc(cm[o]), lane, total);
                    if (total == 0) continue;
                    L.cmask[o][lane] = cm[o];
                    L.coff[o][lane] = off;
                    wave_sync();
                    for (int base = 0; base < total && !any_hot; base += 64) {
                        const int en = base + lane;
                        const bool live = en < total;
                        uint32_t V[4] = {0, 0, 0, 0};
                        if (live) {
                            int lo = 0, hi = 64;
                            while (hi - lo > 1) {
                                const int mid = (lo + hi) >> 1;
                                if (L.coff[o][mid] <= en)
                                    lo = mid;
                                else
                                    hi = mid;
                            }
                            const int pos = p0 + 32 * lo + kth_bit32(L.cmask[o][lo], en - L.coff[o][lo]);
#pragma unroll
                            for (int c = 0; c < 4; c++) V[c] = __brev(gget32(rp.p[c], pos - 1)) >> (32 - kTsoLen);
                        }
                        int b;
                        if constexpr (PAT != 0)
                            b = o ? myers_bound_ct<kTsoLen, PAT, 1>(V, P.lead_max) : myers_bound_ct<kTsoLen, PAT, 0>(V, P.lead_max);
                        else
                            b = myers_bound<kTsoLen>(V, o ? P.rev_idx : P.fwd_idx, P.lead_max);
                        if (__ballot(live && b <= P.tso_max)) any_hot = true;
                    }
                    wave_sync();
                }
            }
        }
        // a read without any verdict is finished here (no internal match, no split); the others queue up for K-CHIM-B / -C
        if (lane == 0) {
            smi_chimera_result res;
            res.n_split = 0;
            res.pos[0] = res.pos[1] = 0;
            res.reason[0] = res.reason[1] = 0;
            res.flags = 0;
            res.n_matches = (any_hot ? SMI_CHIMA_TSO : 0) | (any_pat ? SMI_CHIMA_PAT : 0);
            out[r] = res;
            if (res.n_matches) list[atomicAdd(list_count, 1u)] = (uint32_t)r;
        }
    }
}

// Matches of the exact internal-TSO scan of one queued read (K-CHIM-B -> K-CHIM-C), in the order the split rules take them
struct TsoSlot {
    int n;  // number of matches, > kCap: overflow
    int begin[kCap];
    int kind[kCap];
};

// K-CHIM-B (PART 1): the exact internal TSO scan of the queued reads with the TSO verdict -> TsoSlot.
// K-CHIM-C (PART 2): internal polyA / polyT + adapter for the reads with that verdict, then the split rules on all matches of the
// queued read -> out[r].  Two kernels instead of one so that each is register-allocated for its own part.
#ifndef SMI_CHIM_B_WAVES
#define SMI_CHIM_B_WAVES 4  // waves per SIMD K-CHIM-B is held to
#endif
#ifndef SMI_CHIM_C_WAVES
#define SMI_CHIM_C_WAVES 4  // waves per SIMD K-CHIM-C is held to
#endif
// The first-generation kernel (today: the second-chance path and the cross-check reference) is held to two waves per SIMD, its TSO part to one: at 128 VGPRs it spilled
// a hundred values, and a VGPR spilled inside a divergent region is saved for the lanes active there only (NOTES R4.5) -- room to keep everything
// in registers is worth more here than occupancy.
template <int kTsoLen, int kAdLen, int PART>
__global__ __launch_bounds__(256, PART == 1 ? 1 : 2) void k_chimera(const uint32_t *__restrict__ planes, size_t stride, const uint32_t *__restrict__ pstart,
                                                                   const uint64_t *__restrict__ offsets,
                                                                   const uint32_t *__restrict__ list,
                                                                   const uint32_t *__restrict__ list_count, ChimParams P,
                                                                   TsoSlot *__restrict__ slots, smi_chimera_result *__restrict__ out) {
    __shared__ WaveLds lds_all[4];
    WaveLds &L = lds_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_list = *list_count;
    for (size_t li = wave; li < n_list; li += n_waves) {
        const size_t r = list[li];
        const int verdict = out[r].n_matches;  // left by K-CHIM-A
        if (PART == 1 && !(verdict & SMI_CHIMA_TSO)) continue;
        const uint64_t beg = offsets[r];
        const int len = (int)(offsets[r + 1] - beg);
        smi_chimera_result res;
        res.n_split = 0;
        res.pos[0] = res.pos[1] = 0;
        res.reason[0] = res.reason[1] = 0;
        res.flags = 0;
        res.n_matches = 0;
        ReadPlanes rp;  // (reads shorter than 2 * 70 + 100 never reach the queue, L169)
        const size_t w0 = pstart ? (size_t)pstart[r] : plane_start(beg, r);  // pstart: explicit word start per read (segmented host packer)
#pragma unroll
        for (int c = 0; c < 4; c++) rp.p[c] = planes + c * stride + w0;
        int n_m = 0;        // matches in L.m_*
        bool overflow = false;

        // ---- internal TSO, both orientations (lambda$1 L107-125, lambda$5 L156-166) --------------------------
        // One candidate pool per segment: the forward-TSO candidates, then the reverse-complement ones, aligned 64 at
        // a time whatever their orientation.  The column masks of a 27 x 27 alignment are just the four base planes of
        // the read window (col[c] = plane of template base c), so a candidate costs 4 window fetches, not 27.
        const int last = len - 70;  // min(len - 27, len - 70)
        int n_acc2[2] = {0, 0};
        if constexpr (PART == 1) {
            int skip[2] = {0, 0};  // next position the reference's scan looks at, per orientation
            for (int p0 = 70; p0 <= last; p0 += 4096) {
                const int pl = p0 + 64 * lane;
                unsigned long long cm[2] = {0, 0};
                if (pl <= last) {
                    const PlaneWin pw = load_window(rp, pl - 1);
#pragma unroll
                    for (int o = 0; o < 2; o++) cm[o] = keep_low64(gate_two<kTsoLen>(pw, P.tso4[o]), last - pl + 1);
                }
                int tot0, tot1;
                const int off0 = wave_exscan_i(__popcll(cm[0]), lane, tot0);
                const int off1 = wave_exscan_i(__popcll(cm[1]), lane, tot1) + tot0;
                L.cmask[0][lane] = cm[0];
                L.cmask[1][lane] = cm[1];
                L.coff[0][lane] = off0;
                L.coff[1][lane] = off1;
                wave_sync();
                const int total = tot0 + tot1;
                for (int base = 0; base < total; base += 64) {
                    const int en = base + lane;
                    int pos = 0x7FFFFFFF;
                    float ne = 0.0f;
                    if (en < total) {
                        const int o = en >= tot0 ? 1 : 0;
                        int lo = 0, hi = 64;
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (L.coff[o][mid] <= en)
                                lo = mid;
                            else
                                hi = mid;
                        }
                        pos = p0 + 64 * lo + kth_bit64(L.cmask[o][lo], en - L.coff[o][lo]);
                        uint32_t W[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) W[c] = gget32(rp.p[c], pos - 1) & ((1u << kTsoLen) - 1u);
                        uint32_t col[kTsoLen];
#pragma unroll
                        for (int c = 0; c < kTsoLen; c++) {
                            uint32_t f = 0, r = 0;
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                if ((P.tso4[0][c] >> k) & 1u) f |= W[k];
                                if ((P.tso4[1][c] >> k) & 1u) r |= W[k];
                            }
                            col[c] = o ? r : f;
                        }
                        ne = nw_errors<kTsoLen, nw_band<kTsoLen, 5>()>(col);  // gate_two: >= 5 diagonal matches (smi_nw.h "Band")
                    }
                    const float maxe = (float)P.tso_max;
                    int delta_l = 1;
                    if (maxe < ne) {  // L146-150
                        delta_l = jround(__fsub_rn(ne, maxe)) - 1;
                        if (delta_l < 1) delta_l = 1;
                    }
                    const int ok_l = !(ne > maxe) ? 1 : 0;  // getPosbelowMaxMismatches: key <= max (L302)
                    const int cnt = min(64, total - base);
                    unsigned long long taken[2] = {0, 0};
                    for (int i = 0; i < cnt; i++) {  // scalar fold, scan order inside each orientation
                        const int o = base + i >= tot0 ? 1 : 0;
                        const int p = __builtin_amdgcn_readlane(pos, i);
                        if (p < skip[o]) continue;
                        if (__builtin_amdgcn_readlane(ok_l, i)) taken[o] |= 1ull << i;
                        skip[o] = p + __builtin_amdgcn_readlane(delta_l, i);
                    }
#pragma unroll
                    for (int o = 0; o < 2; o++) {
                        if ((taken[o] >> lane) & 1ull) {
                            const int slot = n_acc2[o] + __popcll(taken[o] & ((1ull << lane) - 1ull));
                            if (slot < kCap) {
                                L.acc_pos[o][slot] = pos;
                                L.acc_ne[o][slot] = ne;
                            }
                        }
                        n_acc2[o] += __popcll(taken[o]);
                    }
                }
                wave_sync();
            }
#pragma unroll 1
        for (int o = 0; o < 2; o++) {
            int n_acc = n_acc2[o];
            if (n_acc > kCap) {
                overflow = true;
                n_acc = kCap;
            }
            wave_sync();
            // sort by (score, position): rank by counting (n_acc <= 64)
            if (lane < n_acc) {
                const float me = L.acc_ne[o][lane];
                const int mp = L.acc_pos[o][lane];
                int rank = 0;
                for (int j = 0; j < n_acc; j++) {
                    const float e = L.acc_ne[o][j];
                    const int p = L.acc_pos[o][j];
                    rank += (e < me || (e == me && p < mp)) ? 1 : 0;
                }
                L.srt_pos[rank] = mp;
                L.srt_ne[rank] = me;
            }
            wave_sync();
            // L115-123: entry i dropped when < 3 away from entry i-1 of the sorted list
            int mypos = 0;
            bool keep = false;
            if (lane < n_acc) {
                mypos = L.srt_pos[lane];
                keep = lane == 0 || abs(mypos - L.srt_pos[lane - 1]) >= 3;
            }
            const unsigned long long kb = __ballot(keep);
            const int idx = __popcll(kb & ((1ull << lane) - 1ull));
            const int n_keep = __popcll(kb);
            wave_sync();
            const int begin = o ? mypos + kTsoLen - 1 : mypos;  // L161
            if (keep) L.acc_pos[o][idx] = begin;
            wave_sync();
            // L163: begin > prev.getAndSet(begin) + 120, prev = previous list element
            bool pass = false;
            if (lane < n_keep) pass = lane == 0 || L.acc_pos[o][lane] > L.acc_pos[o][lane - 1] + 120;
            const unsigned long long pb = __ballot(pass);
            if (pass) {
                const int slot = n_m + __popcll(pb & ((1ull << lane) - 1ull));
                if (slot < kCap) {
                    L.m_begin[slot] = L.acc_pos[o][lane];
                    L.m_kind[slot] = o;
                }
            }
            n_m += __popcll(pb);
            wave_sync();
        }
            // hand the matches to K-CHIM-C
            if (lane < min(n_m, kCap)) {
                slots[li].begin[lane] = L.m_begin[lane];
                slots[li].kind[lane] = L.m_kind[lane];
            }
            if (lane == 0) slots[li].n = overflow ? kCap + 1 : n_m;
            wave_sync();
            continue;
        } else {
            if (verdict & SMI_CHIMA_TSO) {
                n_m = slots[li].n;
                if (n_m > kCap) {
                    overflow = true;
                    n_m = kCap;
                }
                if (lane < n_m) {
                    L.m_begin[lane] = slots[li].begin[lane];
                    L.m_kind[lane] = slots[li].kind[lane];
                }
                wave_sync();
            }
        }

        // ---- internal polyA / polyT + adapter (aTscan L92-136, adapterScan) --------------------------------------
        // window count at pos = exact bases in [pos+1, pos+14] + the base at index off+13, which the reference's
        // window update counts twice (L99-121)
        const int first = P.off - 1, stop = len - P.off;  // pos in [first, stop)
        if (PART == 2 && first < stop && !(P.ablate & 2) && (verdict & SMI_CHIMA_PAT)) {
            int end_cur[2] = {0, 0};                               // [0] A, [1] T
            int fired[2] = {-1, -1};                               // last position that produced an ATposition
            long long prev_start[2] = {-2147483648LL, -2147483648LL};  // prevA_Position / prevT_Position
            const int extra[2] = {(int)gexact_bit(rp, 0, P.off + P.pat_len - 2), (int)gexact_bit(rp, 1, P.off + P.pat_len - 2)};
            for (int p0 = first; p0 < stop; p0 += 4096) {
                const int pl = p0 + 64 * lane;
                unsigned long long trig[2] = {0, 0};
                if (pl < stop && P.pat_thr <= 15) {
                    // exact-A / exact-T bits of positions pl .. pl+127 from one 128-bit window per plane (the packer emits
                    // A, G, C, T, N and nothing else: exact A = A & ~G, exact T = T & ~A)
                    const PlaneWin pw = load_window(rp, pl);
                    const uint64_t elo[2] = {pw.lo[0] & ~pw.lo[1], pw.lo[3] & ~pw.lo[0]};
                    const uint64_t ehi[2] = {pw.hi[0] & ~pw.hi[1], pw.hi[3] & ~pw.hi[0]};
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
                        for (int k = 1; k < 15; k++) {  // bit-sliced sum of the shifted planes (pat_len - 1 of them)
                            if (k >= P.pat_len) break;
                            const unsigned long long x = (elo[t] >> k) | (ehi[t] << (64 - k));
                            const unsigned long long t0 = c0 & x;
                            c0 ^= x;
                            const unsigned long long t1 = c1 & t0;
                            c1 ^= t0;
                            const unsigned long long t2 = c2 & t1;
                            c2 ^= t1;
                            c3 ^= t2;
                        }
                        if (extra[t]) {
                            const unsigned long long t0 = c0;
                            c0 = ~c0;
                            const unsigned long long t1 = c1 & t0;
                            c1 ^= t0;
                            const unsigned long long t2 = c2 & t1;
                            c2 ^= t1;
                            c3 ^= t2;
                        }
                        const unsigned long long cb[4] = {c0, c1, c2, c3};
                        unsigned long long gt = 0, eq = ~0ull;
#pragma unroll
                        for (int b = 3; b >= 0; b--) {  // count >= pat_thr
                            if ((P.pat_thr >> b) & 1)
                                eq &= cb[b];
                            else
                                gt |= eq & cb[b];
                        }
                        const unsigned long long e1 = (elo[t] >> 1) | (ehi[t] << 63);
                        trig[t] = keep_low64((gt | eq) & elo[t] & e1, stop - pl);
                    }
                }
                // walk the triggers in position order (T before A never collide: a base is one or the other)
                for (; !(P.ablate & 64);) {
                    int nxt[2];
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        // first trigger position > end_cur[t] in this segment
                        unsigned long long m = trig[t];
                        const int rel = max(end_cur[t], fired[t]) + 1 - pl;  // bits below rel are not eligible
                        if (rel >= 64)
                            m = 0;
                        else if (rel > 0)
                            m &= ~0ull << rel;
                        const unsigned long long lanes = __ballot(m != 0);
                        if (lanes) {
                            const int l0 = __builtin_ctzll(lanes);
                            const unsigned long long mm = ((unsigned long long)__builtin_amdgcn_readlane((int)(m >> 32), l0) << 32) |
                                                          (uint32_t)__builtin_amdgcn_readlane((int)m, l0);
                            nxt[t] = p0 + 64 * l0 + __builtin_ctzll(mm);
                        } else
                            nxt[t] = 0x7FFFFFFF;
                    }
                    if (nxt[0] == 0x7FFFFFFF && nxt[1] == 0x7FFFFFFF) break;
                    const int t = nxt[1] < nxt[0] ? 1 : 0;
                    const int pos = nxt[t];
                    // currentInWindow at pos
                    int cur = extra[t];
                    {
                        const unsigned long long x = gexact64(rp, t, pos + 1);
                        cur += __popcll(x & ((1ull << (P.pat_len - 1)) - 1ull));
                    }
                    const int at_begin = pos + 1;
                    const int at_end = search_at_end(rp, len, pos, t, cur, P) + 1;
                    end_cur[t] = at_end;
                    fired[t] = pos;
                    const int start = (P.ablate & 32) ? 0 : adapter_scan<kAdLen>(rp, at_begin, at_end, t, lane, P);
                    if (start != 0) {  // lambda$7 L204-207
                        const long long lim = prev_start[t] + 120;
                        prev_start[t] = start;
                        if ((long long)start > lim) {
                            if (n_m < kCap && lane == 0) {
                                L.m_begin[n_m] = start;
                                L.m_kind[n_m] = 2 | (t == 0 ? 1 : 0);  // polyA stretch = reverse adapter
                            }
                            n_m++;
                        }
                    }
                }
            }
        }
        if (n_m > kCap) {
            overflow = true;
            n_m = kCap;
        }
        wave_sync();

        // ---- split rules (L229-286) ----------------------------------------------------------------------------
        // stable sort by begin: rank by counting
        if (lane < n_m) {
            const int mb = L.m_begin[lane];
            int rank = 0;
            for (int j = 0; j < n_m; j++) {
                const int b = L.m_begin[j];
                rank += (b < mb || (b == mb && j < lane)) ? 1 : 0;
            }
            L.order[rank] = lane;
        }
        wave_sync();
        // The < 100 filter (L273-281) compares NEIGHBOURS of the unfiltered list, so it is applied while the list
        // is produced: element i is dropped iff pos[i] - pos[i-1] < 100.
        int n_kept = 0, kept_pos[3], kept_reason[3], prev_sp = 0;
        bool have_prev_sp = false;
        auto emit = [&](int reason, int pos) {
            const bool drop = have_prev_sp && (pos - prev_sp < 100);
            prev_sp = pos;
            have_prev_sp = true;
            if (!drop) {
                if (n_kept < 3) {
                    kept_pos[n_kept] = pos;
                    kept_reason[n_kept] = reason;
                }
                n_kept++;
            }
        };
        auto isolated = [&](int m) {
            const int k = L.m_kind[m], b = L.m_begin[m];
            emit((k & 1) ? SMI_SPLIT_REV_ADAPTER : SMI_SPLIT_FWD_ADAPTER, (k & 1) ? b + 25 : b - 25);  // lambda$10
        };
        if (n_m == 1) {
            if (L.m_kind[0] & 2) isolated(0);
        } else if (n_m > 1) {
            int it = 0;
            int prev = L.order[it++];
            while (it < n_m && prev >= 0) {
                const int cur = L.order[it++];
                const int pk = L.m_kind[prev], ck = L.m_kind[cur];
                const int pbeg = L.m_begin[prev], cbeg = L.m_begin[cur];
                if (cbeg - pbeg > 160) {
                    if (pk & 2) isolated(prev);
                    prev = cur;
                } else if ((pk & 1) && !(ck & 1)) {
                    const int reason = (pk & 2) ? ((ck & 2) ? SMI_SPLIT_RA_FA : SMI_SPLIT_RA_FT)
                                                : ((ck & 2) ? SMI_SPLIT_RT_FA : SMI_SPLIT_RT_FT);  // lambda$11
                    emit(reason, pbeg + (cbeg - pbeg) / 2);
                    prev = it < n_m ? L.order[it++] : -1;
                } else
                    prev = cur;
                if (it >= n_m && prev >= 0 && (L.m_kind[prev] & 2)) isolated(prev);  // L263-264
            }
        }
        res.n_matches = n_m;
        if (n_kept > 2) {
            res.flags |= SMI_CHIM_MULTI;  // MULTI_CHIMERIC_READS_DISCARDED | FAILED, read kept whole (L284-286)
        } else {
            res.n_split = n_kept;
            int lastp = 0;
            for (int i = 0; i < n_kept; i++) {
                res.pos[i] = kept_pos[i];
                res.reason[i] = (uint8_t)kept_reason[i];
                if (kept_pos[i] < lastp || kept_pos[i] > len) res.flags |= SMI_CHIM_RANGE;  // substring would throw
                lastp = kept_pos[i];
            }
        }
        if (overflow) res.flags |= SMI_CHIM_OVERFLOW;
        if (lane == 0) out[r] = res;
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K-CHIM-S: the reads K-CHIM-B / -C flagged SMI_CHIM_OVERFLOW (more than kCap accepted positions or matches: TSO
// concatemers, homopolymers, all-N reads) once more, with every list in global scratch sized by the read's length -- the
// reference has no such cap (ChimeraFindernew.java:L107-332).  One wave per read, UNIFORM control flow: every lane runs the same
// statements on the same values (the alignments of a batch excepted), lane 0 stores.  Rare reads; speed is not a concern here.
// Scratch of read k of the queue: 7 * cap(k) words at scr[scr_off[k]], cap = length + 8.
// ---------------------------------------------------------------------------------------------------------------
template <int kTsoLen, int kAdLen>
__global__ __launch_bounds__(64) void k_chimera_serial(const uint32_t *__restrict__ planes, size_t stride, const uint32_t *__restrict__ pstart,
                                                       const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ list,
                                                       uint32_t n_list, const uint64_t *__restrict__ scr_off, int32_t *__restrict__ scr,
                                                       ChimParams P, smi_chimera_result *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        const size_t r = list[li];
        const uint64_t beg = offsets[r];
        const int len = (int)(offsets[r + 1] - beg);
        const int cap = len + 8;
        int32_t *ps_pos = scr + scr_off[li];
        float *ps_ne = reinterpret_cast<float *>(ps_pos + cap);
        int32_t *m_begin = ps_pos + 2 * cap, *m_kind = ps_pos + 3 * cap, *order = ps_pos + 4 * cap;  // cap entries each (matches are > 120 apart)
        ReadPlanes rp;
        const size_t w0 = pstart ? (size_t)pstart[r] : plane_start(beg, r);  // pstart: explicit word start per read (segmented host packer)
#pragma unroll
        for (int c = 0; c < 4; c++) rp.p[c] = planes + c * stride + w0;
        smi_chimera_result res;
        res.n_split = 0;
        res.pos[0] = res.pos[1] = 0;
        res.reason[0] = res.reason[1] = 0;
        res.flags = 0;
        res.n_matches = 0;
        int n_m = 0;
        const int last = len - 70;
        // ---- internal TSO, orientation by orientation (scanForAdapterOrTSOseqKMERsForInternal L130-156 as the scan runs it) ----
        for (int o = 0; o < 2; o++) {
            int n_ps = 0, delta = 1;
            for (int pos = 70; pos <= last; pos += delta) {
                delta = 1;
                const PlaneWin pw = load_window(rp, pos - 1);
                if (!(gate_two<kTsoLen>(pw, P.tso4[o]) & 1ull)) continue;  // bit 0 = this position (Kmers.nKmersMatching >= 2)
                uint32_t W[4], col[kTsoLen];
#pragma unroll
                for (int c = 0; c < 4; c++) W[c] = gget32(rp.p[c], pos - 1) & ((1u << kTsoLen) - 1u);
#pragma unroll
                for (int c = 0; c < kTsoLen; c++) {
                    uint32_t m = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if ((P.tso4[o][c] >> k) & 1u) m |= W[k];
                    col[c] = m;
                }
                const float ne = nw_errors<kTsoLen>(col);
                const float maxe = (float)P.tso_max;
                if (!((float)jround(ne) > maxe) && !(ne > maxe)) {  // L143-144, then getPosbelowMaxMismatches (key <= max, L302)
                    if (lane == 0) {
                        ps_pos[n_ps] = pos;
                        ps_ne[n_ps] = ne;
                    }
                    n_ps++;
                }
                if (maxe < ne) {  // L146-150
                    delta = jround(__fsub_rn(ne, maxe)) - 1;
                    if (delta < 1) delta = 1;
                }
            }
            __threadfence_block();
            wave_sync();
            // stable sort by score (positions ascending inside one score): selection into `order`
            if (lane == 0) {
                for (int i = 0; i < n_ps; i++) {
                    int rank = 0;
                    for (int j = 0; j < n_ps; j++)
                        rank += (ps_ne[j] < ps_ne[i] || (ps_ne[j] == ps_ne[i] && ps_pos[j] < ps_pos[i])) ? 1 : 0;
                    order[rank] = ps_pos[i];
                }
            }
            __threadfence_block();
            wave_sync();
            // L115-123: entry i dropped when < 3 away from entry i-1 of the sorted list; then begin > prev + 120 (L161-163)
            long long prev = -2147483648LL;
            int prev_sorted = 0;
            for (int i = 0; i < n_ps; i++) {
                const int pp = order[i];
                const bool keep = i == 0 || abs(pp - prev_sorted) >= 3;
                prev_sorted = pp;
                if (!keep) continue;
                const int begin = o ? pp + kTsoLen - 1 : pp;
                const long long lim = prev + 120;
                prev = begin;
                if ((long long)begin > lim) {
                    if (lane == 0) {
                        m_begin[n_m] = begin;
                        m_kind[n_m] = o;
                    }
                    n_m++;
                }
            }
            wave_sync();
        }
        // ---- internal polyA / polyT + adapter (aTscan L92-136): one position at a time --------------------------------
        const int first = P.off - 1, stop = len - P.off;
        if (first < stop) {
            int cur[2] = {0, 0};  // [0] A, [1] T: exact bases in the window
            for (int i = first; i < first + P.pat_len && i < len; i++) {
                cur[0] += (int)gexact_bit(rp, 0, i);
                cur[1] += (int)gexact_bit(rp, 1, i);
            }
            int end_cur[2] = {0, 0};
            long long prev_start[2] = {-2147483648LL, -2147483648LL};
            for (int pos = first; pos < stop; pos++) {
                for (int t = 0; t < 2; t++) {
                    cur[t] -= (int)gexact_bit(rp, t, pos);
                    cur[t] += (int)gexact_bit(rp, t, pos + P.pat_len - 1);
                }
                for (int w = 0; w < 2; w++) {  // T first (L123), then A (L129)
                    const int t = w == 0 ? 1 : 0;
                    if (cur[t] < P.pat_thr || pos <= end_cur[t] || !gexact_bit(rp, t, pos) || !gexact_bit(rp, t, pos + 1)) continue;
                    const int at_begin = pos + 1;
                    const int at_end = search_at_end(rp, len, pos, t, cur[t], P) + 1;
                    end_cur[t] = at_end;
                    const int start = adapter_scan<kAdLen>(rp, at_begin, at_end, t, lane, P);
                    if (start == 0) continue;
                    const long long lim = prev_start[t] + 120;
                    prev_start[t] = start;
                    if ((long long)start > lim) {
                        if (lane == 0) {
                            m_begin[n_m] = start;
                            m_kind[n_m] = 2 | (t == 0 ? 1 : 0);
                        }
                        n_m++;
                    }
                }
            }
        }
        __threadfence_block();
        wave_sync();
        // ---- split rules (L229-286) on the whole list -------------------------------------------------------------------
        if (lane == 0) {
            for (int i = 0; i < n_m; i++) {
                int rank = 0;
                for (int j = 0; j < n_m; j++) rank += (m_begin[j] < m_begin[i] || (m_begin[j] == m_begin[i] && j < i)) ? 1 : 0;
                order[rank] = i;
            }
        }
        __threadfence_block();
        wave_sync();
        int n_kept = 0, kept_pos[3], kept_reason[3], prev_sp = 0;
        bool have_prev_sp = false;
        auto emit = [&](int reason, int pos) {
            const bool drop = have_prev_sp && (pos - prev_sp < 100);
            prev_sp = pos;
            have_prev_sp = true;
            if (!drop) {
                if (n_kept < 3) {
                    kept_pos[n_kept] = pos;
                    kept_reason[n_kept] = reason;
                }
                n_kept++;
            }
        };
        auto isolated = [&](int m) {
            const int k = m_kind[m], b = m_begin[m];
            emit((k & 1) ? SMI_SPLIT_REV_ADAPTER : SMI_SPLIT_FWD_ADAPTER, (k & 1) ? b + 25 : b - 25);
        };
        if (n_m == 1) {
            if (m_kind[0] & 2) isolated(0);
        } else if (n_m > 1) {
            int it = 0;
            int prev = order[it++];
            while (it < n_m && prev >= 0) {
                const int cur = order[it++];
                const int pk = m_kind[prev], ck = m_kind[cur];
                const int pbeg = m_begin[prev], cbeg = m_begin[cur];
                if (cbeg - pbeg > 160) {
                    if (pk & 2) isolated(prev);
                    prev = cur;
                } else if ((pk & 1) && !(ck & 1)) {
                    const int reason = (pk & 2) ? ((ck & 2) ? SMI_SPLIT_RA_FA : SMI_SPLIT_RA_FT) : ((ck & 2) ? SMI_SPLIT_RT_FA : SMI_SPLIT_RT_FT);
                    emit(reason, pbeg + (cbeg - pbeg) / 2);
                    prev = it < n_m ? order[it++] : -1;
                } else
                    prev = cur;
                if (it >= n_m && prev >= 0 && (m_kind[prev] & 2)) isolated(prev);
            }
        }
        res.n_matches = n_m;
        if (n_kept > 2) {
            res.flags |= SMI_CHIM_MULTI;
        } else {
            res.n_split = n_kept;
            int lastp = 0;
            for (int i = 0; i < n_kept; i++) {
                res.pos[i] = kept_pos[i];
                res.reason[i] = (uint8_t)kept_reason[i];
                if (kept_pos[i] < lastp || kept_pos[i] > len) res.flags |= SMI_CHIM_RANGE;
                lastp = kept_pos[i];
            }
        }
        if (lane == 0) out[r] = res;
        wave_sync();
    }
}

// =================================================================================================================
// K-CHIM-B, second generation (round 4): SELECT -> ALIGN -> FOLD.
//
// The exact internal TSO scan of one read is a chain: the scan visits the gated positions in order, aligns each one it
// visits, and a rejected alignment with nErrors > max makes it skip delta = round(nErrors - max) - 1 positions
// (AdapterTSOanalyzer.java:L146-150).  The first generation (k_chimera<.., 1>) therefore aligned EVERY gated position
// of a queued read (~70 per read, one wave per read, 2 batches at half-filled lanes).  Almost all of that work cannot
// matter:
//   * only a position whose Levenshtein bound (myers_bound) is <= max can be ACCEPTED ("hot");
//   * a gated position c only ever influences whether positions in (c, c + delta(c)) are visited, and
//     delta(c) < kDmax for every gated position (below);
// so the accepted set is decided by the hot positions and, transitively, by the gated positions less than kDmax in
// front of a position that matters ("needed": the backward closure).  A gated position with nothing needed inside
// (c, c + kDmax) is dropped without changing what the scan accepts: the first needed position behind such a gap is
// visited whatever happened before it, because a skip from a position p lands at most at p + kDmax - 1.
//
// delta < kDmax.  A gated position has >= 5 pattern bases matching on the diagonal (two 4-mers), so the optimal
// alignment scores at least the diagonal's 5*5 - 5*(N-5).  With m matches, x mismatches and g gaps per sequence
// (m + x + g = N) a path scores at most 5m - 5x - 9g (a leading template gap costs 4, every other gap 5), hence
// 10m - 4g >= 50, and #x = x + 2g = N - m + g <= N - 5 + 0.6g with g <= (N - 5) / 1.4: #x <= 31 for N = 27, 24 for
// N = 22, and delta = round(#x - 0.9 lead - max) - 1 <= 30.  kDmax = 32; K-CHIM-B-FOLD still checks every delta it
// computes and hands the read to the serial kernel if one ever reached kDmax.
//
//   k_chimb_select  one wave per queued read: 4-mer gates (lane = 32 positions), the bound for every gated position,
//                   the closure from the right, the needed positions appended to a global queue
//   k_chimb_align   one LANE per queued position, all lanes busy: 27 x 27 banded alignment, error count only
//   k_chimb_fold    one wave per queued read: the skip rule over its needed positions, the accepted ones sorted and
//                   filtered as before -> TsoSlot
// =================================================================================================================
constexpr int kDmax = 32;
constexpr int kSubQ = 64;   // regions of the position queue (k_chimb_select2)
constexpr int kSelCap = 512;  // gated positions per orientation of one read held in LDS; more: the serial kernel takes the read

struct BHead {  // per queue slot: where the read's needed positions sit in the global queue (orientation 0 first)
    uint32_t first, n0, n1, flags;  // flags 1: over capacity -> serial kernel
};

struct SelLds {
    uint32_t cmask[2][64];
    int coff[2][64];
    int cpos[2][kSelCap];
    uint8_t cflag[2][kSelCap];  // bit 0 hot, bit 1 needed
};

// 4-mer gate (>= 2 matching 4-mers on the diagonal) of pattern orientation O for the 32 scan positions whose window is W
template <int kTsoLen, int PAT, int O>
__device__ __forceinline__ uint32_t tso_gate32(const uint64_t (&W)[4], const FilterParams &P) {
    auto mj = [&](int j) -> uint32_t {
        if constexpr (PAT != 0) {
            return (uint32_t)(W[gate_plane<PAT, kTsoLen>(O, j)] >> j);
        } else {
            const uint32_t a4 = P.tso4[O][j];
            uint64_t x = 0;
#pragma unroll
            for (int c = 0; c < 4; c++)
                if ((a4 >> c) & 1u) x |= W[c];
            return (uint32_t)(x >> j);
        }
    };
    uint32_t any = 0, two = 0;
    const uint32_t m0 = mj(0), m1 = mj(1);
    uint32_t m_prev = mj(2);
    uint32_t p01 = m0 & m1, p12 = m1 & m_prev;
#pragma unroll
    for (int i = 0; i + 3 < kTsoLen; i++) {
        const uint32_t m3 = mj(i + 3);
        const uint32_t p23 = m_prev & m3;
        const uint32_t k = p01 & p23;
        two |= any & k;
        any |= k;
        p01 = p12;
        p12 = p23;
        m_prev = m3;
    }
    return two;
}

// the gated positions of one orientation in this 2048-position segment: bound each, append (position, hot) to the LDS list
template <int kTsoLen, int PAT, int O>
__device__ __forceinline__ void select_orient(SelLds &L, const ReadPlanes &rp, const FilterParams &P, int p0, int lane, uint32_t cm, int &n_c) {
    int total;
    const int off = wave_exscan_i(__popc(cm), lane, total);
    if (total == 0) return;
    L.cmask[O][lane] = cm;
    L.coff[O][lane] = off;
    wave_sync();
    for (int base = 0; base < total; base += 64) {
        const int en = base + lane;
        const bool live = en < total;
        uint32_t V[4] = {0, 0, 0, 0};
        int pos = 0;
        if (live) {
            int lo = 0, hi = 64;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (L.coff[O][mid] <= en)
                    lo = mid;
                else
                    hi = mid;
            }
            pos = p0 + 32 * lo + kth_bit32(L.cmask[O][lo], en - L.coff[O][lo]);
#pragma unroll
            for (int c = 0; c < 4; c++) V[c] = __brev(gget32(rp.p[c], pos - 1)) >> (32 - kTsoLen);
        }
        bool hot = live;
        if (P.myers_ok) {  // (a pattern with IUPAC codes has no bound: every gated position counts as hot, i.e. everything is aligned)
            int b;
            if constexpr (PAT != 0)
                b = myers_bound_ct<kTsoLen, PAT, O>(V, P.lead_max);
            else
                b = myers_bound<kTsoLen>(V, O ? P.rev_idx : P.fwd_idx, P.lead_max);
            hot = live && b <= P.tso_max;
        }
        const int slot = n_c + en;
        if (live && slot < kSelCap) {
            L.cpos[O][slot] = pos;
            L.cflag[O][slot] = hot ? 1 : 0;
        }
    }
    n_c += total;
    wave_sync();
}

template <int kTsoLen, int PAT>
__global__ __launch_bounds__(256) void k_chimb_select(const uint32_t *__restrict__ planes, size_t stride, const uint32_t *__restrict__ pstart,
                                                      const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ list,
                                                      const uint32_t *__restrict__ list_count, FilterParams P,
                                                      const smi_chimera_result *__restrict__ out, BHead *__restrict__ heads,
                                                      uint64_t *__restrict__ cand_abs, uint32_t *__restrict__ gcount, uint32_t cap, uint32_t *__restrict__ dbg) {
    __shared__ SelLds lds_all[4];
    SelLds &L = lds_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_list = *list_count;
    for (size_t li = wave; li < n_list; li += n_waves) {
        const size_t r = list[li];
        if (!(out[r].n_matches & SMI_CHIMA_TSO)) continue;  // K-CHIM-A's verdict
        const uint64_t beg = offsets[r];
        const int len = (int)(offsets[r + 1] - beg);
        ReadPlanes rp;
        const size_t w0 = pstart ? (size_t)pstart[r] : plane_start(beg, r);
#pragma unroll
        for (int c = 0; c < 4; c++) rp.p[c] = planes + c * stride + w0;
        const int last = len - 70;
        int n_c[2] = {0, 0};
        for (int p0 = 70; p0 <= last; p0 += 2048) {
            const int pl = p0 + 32 * lane;
            uint32_t cm0 = 0, cm1 = 0;
            if (pl <= last) {
                uint64_t W[4];
#pragma unroll
                for (int c = 0; c < 4; c++) W[c] = gget64(rp.p[c], pl - 1);
                const int nb = last - pl + 1;
                const uint32_t keep = nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u);
                cm0 = tso_gate32<kTsoLen, PAT, 0>(W, P) & keep;
                cm1 = tso_gate32<kTsoLen, PAT, 1>(W, P) & keep;
            }
            select_orient<kTsoLen, PAT, 0>(L, rp, P, p0, lane, cm0, n_c[0]);
            select_orient<kTsoLen, PAT, 1>(L, rp, P, p0, lane, cm1, n_c[1]);
        }
        BHead h;
        h.first = h.n0 = h.n1 = h.flags = 0;
        if (n_c[0] > kSelCap || n_c[1] > kSelCap) {
            h.flags = 1;
            if (lane == 0) {
                heads[li] = h;
                atomicAdd(dbg + 5, 1u);
            }
            continue;
        }
        // closure from the right, 64 gated positions at a time: needed = hot, or something needed lies less than kDmax behind
        uint32_t n_need[2] = {0, 0};
#pragma unroll 1
        for (int o = 0; o < 2; o++) {
            long long next_needed = 1LL << 40;
            for (int hi = n_c[o]; hi > 0;) {
                const int lo = hi > 64 ? hi - 64 : 0;
                const int i = lo + lane;
                const bool live = i < hi;
                const int pos_i = live ? L.cpos[o][i] : 0;
                const bool hot_i = live && (L.cflag[o][i] & 1);
                unsigned long long N = __ballot(hot_i);
                bool need_i = hot_i;
                for (;;) {
                    const unsigned long long above = lane == 63 ? 0ull : ((N >> (lane + 1)) << (lane + 1));
                    const int j = above ? __builtin_ctzll(above) : -1;
                    const int pj_in = __shfl(pos_i, j < 0 ? 0 : j);
                    const long long pj = j >= 0 ? (long long)pj_in : next_needed;
                    need_i = live && (hot_i || pj - (long long)pos_i < kDmax);
                    const unsigned long long N2 = __ballot(need_i);
                    if (N2 == N) break;
                    N = N2;
                }
                if (live) L.cflag[o][i] = (uint8_t)((hot_i ? 1 : 0) | (need_i ? 2 : 0));
                if (N) next_needed = __shfl(pos_i, __builtin_ctzll(N));
                n_need[o] += (uint32_t)__popcll(N);
                hi = lo;
            }
        }
        wave_sync();
        const uint32_t total = n_need[0] + n_need[1];
        uint32_t first = 0;
        if (total) {
            if (lane == 0) first = atomicAdd(gcount, total);
            first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        }
        if ((unsigned long long)first + total > cap) {
            h.flags = 1;
            for (unsigned long long e = (unsigned long long)first + lane; e < cap; e += 64) cand_abs[e] = ~0ull;  // nothing of this read is queued: mark the room it took
            if (lane == 0) {
                heads[li] = h;
                atomicAdd(dbg + 6, 1u);
            }
            continue;
        }
        const uint64_t bit_base = (uint64_t)w0 * 32u;
#pragma unroll 1
        for (int o = 0; o < 2; o++) {
            uint32_t run = first + (o ? n_need[0] : 0u);
            for (int lo = 0; lo < n_c[o]; lo += 64) {
                const int i = lo + lane;
                const bool need = i < n_c[o] && (L.cflag[o][i] & 2);
                const unsigned long long nb = __ballot(need);
                if (need) cand_abs[run + __popcll(nb & ((1ull << lane) - 1ull))] = (bit_base + (uint64_t)(L.cpos[o][i] - 1)) | ((uint64_t)o << 63);
                run += (uint32_t)__popcll(nb);
            }
        }
        h.first = first;
        h.n0 = n_need[0];
        h.n1 = n_need[1];
        if (lane == 0) heads[li] = h;
        wave_sync();
    }
}

// one lane per queued position
template <int kTsoLen>
__global__ __launch_bounds__(256, SMI_CHIM_B_WAVES) void k_chimb_align(const uint32_t *__restrict__ planes, size_t stride, const uint64_t *__restrict__ cand_abs,
                                                                         const uint32_t *__restrict__ gcount, uint32_t cap, ChimParams P,
                                                                         float *__restrict__ cand_ne) {
    // gridDim.y regions of cap / gridDim.y entries, a counter each (1: one queue)
    const uint32_t sub_cap = cap / gridDim.y, sub_base = blockIdx.y * sub_cap;
    const uint32_t total = min(gcount[blockIdx.y], sub_cap);
    for (uint32_t e0 = blockIdx.x * blockDim.x + threadIdx.x; e0 < total; e0 += gridDim.x * blockDim.x) {
        const uint32_t e = sub_base + e0;
        const uint64_t a = cand_abs[e];
        if (a == ~0ull) continue;  // room that was reserved by reads which then did not fit (k_chimb_select*)
        const int o = (int)(a >> 63);
        const uint64_t bit = a & 0x7FFFFFFFFFFFFFFFull;
        const size_t w = (size_t)(bit >> 5);
        const int sh = (int)(bit & 31u);
        uint32_t W[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t *pl = planes + c * stride + w;
            W[c] = (uint32_t)((((uint64_t)pl[1] << 32) | pl[0]) >> sh) & ((1u << kTsoLen) - 1u);
        }
        uint32_t col[kTsoLen];
#pragma unroll
        for (int c = 0; c < kTsoLen; c++) {
            uint32_t f = 0, r = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((P.tso4[0][c] >> k) & 1u) f |= W[k];
                if ((P.tso4[1][c] >> k) & 1u) r |= W[k];
            }
            col[c] = o ? r : f;
        }
        cand_ne[e] = nw_errors<kTsoLen, nw_band<kTsoLen, 5>()>(col);
    }
}

struct FoldLds {
    int acc_pos[2][kCap];
    float acc_ne[2][kCap];
    int srt_pos[kCap];
    float srt_ne[kCap];
    int m_begin[kCap];
    int m_kind[kCap];
};

template <int kTsoLen>
__global__ __launch_bounds__(256) void k_chimb_fold(const uint32_t *__restrict__ pstart, const uint64_t *__restrict__ offsets,
                                                    const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_count, int tso_max,
                                                    const smi_chimera_result *__restrict__ out, const BHead *__restrict__ heads,
                                                    const uint64_t *__restrict__ cand_abs, const float *__restrict__ cand_ne,
                                                    TsoSlot *__restrict__ slots, uint32_t *__restrict__ dbg) {
    __shared__ FoldLds lds_all[4];
    FoldLds &L = lds_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_list = *list_count;
    for (size_t li = wave; li < n_list; li += n_waves) {
        const size_t r = list[li];
        if (!(out[r].n_matches & SMI_CHIMA_TSO)) continue;
        const BHead h = heads[li];
        if (h.flags & 1u) {
            if (lane == 0) slots[li].n = kCap + 1;
            continue;
        }
        const size_t w0 = pstart ? (size_t)pstart[r] : plane_start(offsets[r], r);
        const uint64_t bit_base = (uint64_t)w0 * 32u;
        int n_m = 0;
        bool overflow = false;
        int n_acc2[2] = {0, 0};
        const float maxe = (float)tso_max;
#pragma unroll 1
        for (int o = 0; o < 2; o++) {
            const int n = (int)(o ? h.n1 : h.n0);
            const uint32_t e0 = h.first + (o ? h.n0 : 0u);
            int skip = 0;  // next position the reference's scan looks at
            for (int k = 0; k < n; k += 64) {
                const int e = k + lane;
                int pos = 0x7FFFFFFF;
                float ne = 0.0f;
                if (e < n) {
                    pos = (int)((cand_abs[e0 + e] & 0x7FFFFFFFFFFFFFFFull) - bit_base) + 1;
                    ne = cand_ne[e0 + e];
                }
                int delta_l = 1;
                if (maxe < ne) {  // L146-150
                    delta_l = jround(__fsub_rn(ne, maxe)) - 1;
                    if (delta_l < 1) delta_l = 1;
                }
                if (__ballot(e < n && delta_l >= kDmax)) overflow = true;  // (never seen; the closure of k_chimb_select rests on it)
                const int ok_l = !(ne > maxe) ? 1 : 0;  // getPosbelowMaxMismatches: key <= max (L302)
                const int cnt = min(64, n - k);
                unsigned long long taken = 0;
                for (int i = 0; i < cnt; i++) {  // scalar fold in scan order
                    const int p = __builtin_amdgcn_readlane(pos, i);
                    if (p < skip) continue;
                    if (__builtin_amdgcn_readlane(ok_l, i)) taken |= 1ull << i;
                    skip = p + __builtin_amdgcn_readlane(delta_l, i);
                }
                if ((taken >> lane) & 1ull) {
                    const int slot = n_acc2[o] + __popcll(taken & ((1ull << lane) - 1ull));
                    if (slot < kCap) {
                        L.acc_pos[o][slot] = pos;
                        L.acc_ne[o][slot] = ne;
                    }
                }
                n_acc2[o] += __popcll(taken);
            }
        }
#pragma unroll 1
        for (int o = 0; o < 2; o++) {
            int n_acc = n_acc2[o];
            if (n_acc > kCap) {
                overflow = true;
                n_acc = kCap;
            }
            wave_sync();
            // sort by (score, position): rank by counting (n_acc <= 64)
            if (lane < n_acc) {
                const float me = L.acc_ne[o][lane];
                const int mp = L.acc_pos[o][lane];
                int rank = 0;
                for (int j = 0; j < n_acc; j++) {
                    const float e = L.acc_ne[o][j];
                    const int p = L.acc_pos[o][j];
                    rank += (e < me || (e == me && p < mp)) ? 1 : 0;
                }
                L.srt_pos[rank] = mp;
                L.srt_ne[rank] = me;
            }
            wave_sync();
            // L115-123: entry i dropped when < 3 away from entry i-1 of the sorted list
            int mypos = 0;
            bool keep = false;
            if (lane < n_acc) {
                mypos = L.srt_pos[lane];
                keep = lane == 0 || abs(mypos - L.srt_pos[lane - 1]) >= 3;
            }
            const unsigned long long kb = __ballot(keep);
            const int idx = __popcll(kb & ((1ull << lane) - 1ull));
            const int n_keep = __popcll(kb);
            wave_sync();
            const int begin = o ? mypos + kTsoLen - 1 : mypos;  // L161
            if (keep) L.acc_pos[o][idx] = begin;
            wave_sync();
            // L163: begin > prev.getAndSet(begin) + 120, prev = previous list element
            bool pass = false;
            if (lane < n_keep) pass = lane == 0 || L.acc_pos[o][lane] > L.acc_pos[o][lane - 1] + 120;
            const unsigned long long pb = __ballot(pass);
            if (pass) {
                const int slot = n_m + __popcll(pb & ((1ull << lane) - 1ull));
                if (slot < kCap) {
                    L.m_begin[slot] = L.acc_pos[o][lane];
                    L.m_kind[slot] = o;
                }
            }
            n_m += __popcll(pb);
            wave_sync();
        }
        if (lane < min(n_m, kCap)) {
            slots[li].begin[lane] = L.m_begin[lane];
            slots[li].kind[lane] = L.m_kind[lane];
        }
        if (lane == 0) {
            slots[li].n = (overflow || n_m > kCap) ? kCap + 1 : n_m;
            if (overflow || n_m > kCap) atomicAdd(dbg + 7, 1u);
        }
        wave_sync();
    }
}

// =================================================================================================================
// K-CHIM-C, second generation (round 4).  The first generation walked the internal polyA / polyT stretches of ONE read
// per wave (a serial, data-dependent walk: 17 of 64 lanes active, 38 % of the wave cycles waiting).  The walk is serial
// per read but the reads are independent, so the unit of the walk is now a LANE:
//   k_chimc_trig   wave per queued read, lane per 32 positions: the start condition of aTscan as bit masks (bit-sliced
//                  window counters) -> two words per plane word
//   k_chimc_walk   LANE per queued read: aTscan's walk over its trigger bits, searchATend -> stretch records
//   k_chimc_gate   lane per stretch: the 51-base window's 3 x 4-mer gate, the Levenshtein bound of the gated positions;
//                  the positions of a stretch with a possible match are queued
//   k_chimc_align  lane per queued position: the kAdLen x kAdLen alignment (error count + mismatches)
//   k_chimc_rules  lane per queued read: position-skip fold and best-score choice per stretch, the > 120 filter, the
//                  matches of the TSO scan (TsoSlot), the split rules -> smi_chimera_result
// Per-lane lists are capped (kStPerRead stretches, kMatchCap matches); a read over a cap gets SMI_CHIM_OVERFLOW and goes
// through the serial kernel, which has no caps.
// =================================================================================================================
constexpr int kStPerRead = 16;
constexpr int kMatchCap = 64;  // = kCap of the first generation

struct Stretch {
    uint32_t li;        // queue slot of the read
    int t;              // 0 polyA, 1 polyT
    int at_begin, at_end;
};
struct StretchRes {     // k_chimc_gate -> k_chimc_rules
    uint32_t gate;      // gated scan positions (bit i = position i + 1), 0: no match possible
    uint32_t first;     // first entry of the stretch in the alignment queue
};
struct AdEntry {
    uint32_t s;         // stretch
    uint32_t i;         // scan position - 1
};

// start condition of aTscan for the 32 positions q = 32 w .. 32 w + 31 of a read (0-based; L92-136): window count >= threshold,
// bases q and q + 1 exact.  E: exact-base bits from bit 32 w on (64 of them).
__device__ __forceinline__ uint32_t at_triggers32(uint64_t e, int extra, int pat_len, int pat_thr, int q0, int first, int stop) {
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
    for (int k = 1; k < 15; k++) {  // bit-sliced sum of the shifted planes (pat_len - 1 of them)
        if (k >= pat_len) break;
        const uint32_t x = (uint32_t)(e >> k);
        const uint32_t t0 = c0 & x;
        c0 ^= x;
        const uint32_t t1 = c1 & t0;
        c1 ^= t0;
        const uint32_t t2 = c2 & t1;
        c2 ^= t1;
        c3 ^= t2;
    }
    if (extra) {  // the base at index off + 13, which the reference's window update counts twice (L99-121)
        const uint32_t t0 = c0;
        c0 = ~c0;
        const uint32_t t1 = c1 & t0;
        c1 ^= t0;
        const uint32_t t2 = c2 & t1;
        c2 ^= t1;
        c3 ^= t2;
    }
    const uint32_t cb[4] = {c0, c1, c2, c3};
    uint32_t gt = 0, eq = ~0u;
#pragma unroll
    for (int b = 3; b >= 0; b--) {
        if ((pat_thr >> b) & 1)
            eq &= cb[b];
        else
            gt |= eq & cb[b];
    }
    uint32_t m = (gt | eq) & (uint32_t)e & (uint32_t)(e >> 1);
    if (q0 < first) m = first - q0 >= 32 ? 0u : (m & (~0u << (first - q0)));
    if (q0 + 32 > stop) m = stop - q0 <= 0 ? 0u : (m & ((1u << (stop - q0)) - 1u));
    return m;
}

__global__ __launch_bounds__(256) void k_chimc_trig(const uint32_t *__restrict__ planes, size_t stride, const uint32_t *__restrict__ pstart,
                                                    const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ list,
                                                    const uint32_t *__restrict__ list_count, ChimParams P,
                                                    const smi_chimera_result *__restrict__ out, uint32_t *__restrict__ trig) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_list = *list_count;
    for (size_t li = wave; li < n_list; li += n_waves) {
        const size_t r = list[li];
        if (!(out[r].n_matches & SMI_CHIMA_PAT)) continue;
        const uint64_t beg = offsets[r];
        const int len = (int)(offsets[r + 1] - beg);
        const size_t w0 = pstart ? (size_t)pstart[r] : plane_start(beg, r);
        const uint32_t *pa = planes + w0, *pg = planes + stride + w0, *pt = planes + 3 * stride + w0;
        const int first = P.off - 1, stop = len - P.off;
        const int xb = P.off + P.pat_len - 2;
        const uint32_t xa = pa[xb >> 5], xg = pg[xb >> 5], xt = pt[xb >> 5];
        const int extra_a = (int)(((xa & ~xg) >> (xb & 31)) & 1u), extra_t = (int)(((xt & ~xa) >> (xb & 31)) & 1u);
        const int n_words = (len + 31) >> 5;
        for (int w = lane; w < n_words; w += 64) {
            uint32_t ma = 0, mt = 0;
            if (P.pat_thr <= 15 && 32 * w < stop && 32 * w + 32 > first) {
                const uint64_t A = ((uint64_t)pa[w + 1] << 32) | pa[w], G = ((uint64_t)pg[w + 1] << 32) | pg[w], T = ((uint64_t)pt[w + 1] << 32) | pt[w];
                ma = at_triggers32(A & ~G, extra_a, P.pat_len, P.pat_thr, 32 * w, first, stop);
                mt = at_triggers32(T & ~A, extra_t, P.pat_len, P.pat_thr, 32 * w, first, stop);
            }
            trig[w0 + w] = ma;
            trig[stride + w0 + w] = mt;
        }
    }
}

__global__ __launch_bounds__(64) void k_chimc_walk(const uint32_t *__restrict__ planes, size_t stride, const uint32_t *__restrict__ pstart,
                                                   const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ list,
                                                   const uint32_t *__restrict__ list_count, ChimParams P,
                                                   const smi_chimera_result *__restrict__ out, const uint32_t *__restrict__ trig,
                                                   Stretch *__restrict__ st, uint32_t *__restrict__ st_count, uint32_t st_cap,
                                                   uint32_t *__restrict__ read_st, uint8_t *__restrict__ read_nst, uint32_t *__restrict__ dbg) {
    const size_t n_list = *list_count;
    const size_t li = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (li >= n_list) return;
    const size_t r = list[li];
    if (!(out[r].n_matches & SMI_CHIMA_PAT)) {
        read_nst[li] = 0;
        return;
    }
    const uint64_t beg = offsets[r];
    const int len = (int)(offsets[r + 1] - beg);
    ReadPlanes rp;
    const size_t w0 = pstart ? (size_t)pstart[r] : plane_start(beg, r);
#pragma unroll
    for (int c = 0; c < 4; c++) rp.p[c] = planes + c * stride + w0;
    const int first = P.off - 1, stop = len - P.off;
    int n_st = 0;
    bool over = false;
    if (first < stop) {
        const int extra[2] = {(int)gexact_bit(rp, 0, P.off + P.pat_len - 2), (int)gexact_bit(rp, 1, P.off + P.pat_len - 2)};
        int end_cur[2] = {0, 0}, fired[2] = {-1, -1};
        auto next_trig = [&](int t, int from) -> int {  // first trigger position >= from
            if (from < first) from = first;
            const uint32_t *tw = trig + (size_t)t * stride + w0;
            // eight words per round trip (the words behind the read's end are zero pads / other reads' words inside the buffer: the
            // position test below drops what lies beyond `stop`)
            for (int w = from >> 5; 32 * w < stop; w += 8) {
                uint32_t m[8];
#pragma unroll
                for (int k = 0; k < 8; k++) m[k] = tw[w + k];
                if (w == (from >> 5)) m[0] &= ~0u << (from & 31);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (m[k]) {
                        const int q = 32 * (w + k) + __builtin_ctz(m[k]);
                        return q < stop ? q : 0x7FFFFFFF;
                    }
                }
            }
            return 0x7FFFFFFF;
        };
        for (;;) {
            const int n0 = next_trig(0, max(end_cur[0], fired[0]) + 1), n1 = next_trig(1, max(end_cur[1], fired[1]) + 1);
            if (n0 == 0x7FFFFFFF && n1 == 0x7FFFFFFF) break;
            const int t = n1 < n0 ? 1 : 0;
            const int pos = t ? n1 : n0;
            int cur = extra[t];
            cur += __popcll(gexact64(rp, t, pos + 1) & ((1ull << (P.pat_len - 1)) - 1ull));
            const int at_end = search_at_end(rp, len, pos, t, cur, P) + 1;
            end_cur[t] = at_end;
            fired[t] = pos;
            if (n_st < kStPerRead) {
                const uint32_t idx = atomicAdd(st_count, 1u);
                if (idx < st_cap) {
                    Stretch x;
                    x.li = (uint32_t)li;
                    x.t = t;
                    x.at_begin = pos + 1;
                    x.at_end = at_end;
                    st[idx] = x;
                    read_st[li * kStPerRead + n_st] = idx;
                    n_st++;
                } else {
                    if (!over) atomicAdd(dbg + 1, 1u);
                    over = true;
                }
            } else {
                if (!over) atomicAdd(dbg + 0, 1u);
                over = true;
            }
        }
    }
    read_nst[li] = (uint8_t)(over ? 255 : n_st);
}

template <int kAdLen>
struct AdWindow {  // the 51-base window next to a stretch (adapterScan L159-221), per lane
    ReadPlanes rp;
    int start_range, end_range, is_t;
    __device__ __forceinline__ void set(const uint32_t *planes, size_t stride, size_t w0, const Stretch &x, const ChimParams &P) {
#pragma unroll
        for (int c = 0; c < 4; c++) rp.p[c] = planes + c * stride + w0;
        is_t = x.t;
        if (is_t) {
            start_range = x.at_begin - P.bc_umi - 30 - 10;
            end_range = start_range + 30 + 20;
        } else {
            end_range = x.at_end + P.bc_umi + 30 + 10;
            start_range = end_range - 30 - 20;
        }
    }
    // bit k = sub[shift + k] matches adapter base i (the sub-sequence is reverse-complemented when the stretch is polyA)
    __device__ __forceinline__ uint32_t col_bits(const ChimParams &P, int i, int shift) const {
        if (is_t) return gmatch32(rp, P.ad4[i], start_range - 1 + shift);
        return __brev(gmatch32(rp, comp4(P.ad4[i]), end_range - 32 - shift));
    }
};

template <int kAdLen>
__global__ __launch_bounds__(256) void k_chimc_gate(const uint32_t *__restrict__ planes, size_t stride, const uint32_t *__restrict__ pstart,
                                                    const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ list, ChimParams P,
                                                    const Stretch *__restrict__ st, const uint32_t *__restrict__ st_count, uint32_t st_cap,
                                                    StretchRes *__restrict__ sres, AdEntry *__restrict__ entries, uint32_t *__restrict__ n_entries,
                                                    uint32_t entry_cap, uint32_t *__restrict__ dbg) {
    constexpr int NPOS = 51 - kAdLen;
    const uint32_t n_st = min(*st_count, st_cap);
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_st; s += gridDim.x * blockDim.x) {
        const Stretch x = st[s];
        const size_t r = list[x.li];
        const size_t w0 = pstart ? (size_t)pstart[r] : plane_start(offsets[r], r);
        AdWindow<kAdLen> W;
        W.set(planes, stride, w0, x, P);
        uint32_t any = 0, two = 0, three = 0;
        {
            uint32_t m0 = W.col_bits(P, 0, 0), m1 = W.col_bits(P, 1, 1), m2 = W.col_bits(P, 2, 2);
#pragma unroll
            for (int i = 0; i + 3 < kAdLen; i++) {
                const uint32_t m3 = W.col_bits(P, i + 3, i + 3);
                const uint32_t k = m0 & m1 & m2 & m3;
                three |= two & k;
                two |= any & k;
                any |= k;
                m0 = m1;
                m1 = m2;
                m2 = m3;
            }
        }
        uint32_t gate = three & ((1u << NPOS) - 1u);  // minKmersMatching = 3 (L173)
        if (gate && P.myers_ok) {
            // no scan position with an alignment that could be accepted: the result list is empty whatever the skip rule visits
            bool hot = false;
            for (uint32_t g = gate; g && !hot; g &= g - 1) {
                const int i = __builtin_ctz(g);
                uint32_t V[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t w = W.is_t ? gget32(W.rp.p[c], W.start_range - 1 + i) : __brev(gget32(W.rp.p[3 - c], W.end_range - 32 - i));
                    V[c] = __brev(w) >> (32 - kAdLen);
                }
                hot = myers_bound<kAdLen>(V, P.ad_idx, P.ad_lead_max) <= P.ad_max;
            }
            if (!hot) gate = 0;
        }
        StretchRes res;
        res.gate = gate;
        res.first = 0;
        if (gate) {
            const uint32_t n = (uint32_t)__popc(gate);
            const uint32_t first = atomicAdd(n_entries, n);
            if ((unsigned long long)first + n > entry_cap) {
                res.gate = 0xFFFFFFFFu;  // out of queue space: the first-generation kernels take the read
                atomicAdd(dbg + 2, 1u);
                for (unsigned long long e = first; e < entry_cap; e++) {  // mark the room it took
                    AdEntry x;
                    x.s = 0xFFFFFFFFu;
                    x.i = 0;
                    entries[e] = x;
                }
            } else {
                res.first = first;
                uint32_t k = 0;
                for (uint32_t g = gate; g; g &= g - 1) {
                    AdEntry e;
                    e.s = s;
                    e.i = (uint32_t)__builtin_ctz(g);
                    entries[first + k++] = e;
                }
            }
        }
        sres[s] = res;
    }
}

template <int kAdLen>
__global__ __launch_bounds__(256, SMI_CHIM_C_WAVES) void k_chimc_align(const uint32_t *__restrict__ planes, size_t stride, const uint32_t *__restrict__ pstart,
                                                                         const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ list, ChimParams P,
                                                                         const Stretch *__restrict__ st, const AdEntry *__restrict__ entries,
                                                                         const uint32_t *__restrict__ n_entries, uint32_t entry_cap,
                                                                         float *__restrict__ e_ne, int *__restrict__ e_nmis) {
    const uint32_t n = min(*n_entries, entry_cap);
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const AdEntry a = entries[e];
        if (a.s == 0xFFFFFFFFu) continue;  // room reserved by a stretch that did not fit
        const Stretch x = st[a.s];
        const size_t r = list[x.li];
        const size_t w0 = pstart ? (size_t)pstart[r] : plane_start(offsets[r], r);
        AdWindow<kAdLen> W;
        W.set(planes, stride, w0, x, P);
        uint32_t col[kAdLen];
#pragma unroll
        for (int c = 0; c < kAdLen; c++) col[c] = W.col_bits(P, c, (int)a.i) & ((1u << kAdLen) - 1u);
        AlnStats sa;
        nw_full<kAdLen, false, false, nw_band<kAdLen, 6>()>(col, 0, sa);  // the gate asked for 3 matching 4-mers = 6 diagonal matches
        e_ne[e] = sa.ne;
        e_nmis[e] = sa.nmis;
    }
}

struct RulesLds {  // per lane: [slot][lane]; a match is begin << 2 | kind (bit 0 is_reverse, bit 1 is_adapter)
    int m[kMatchCap][64];
};

__global__ __launch_bounds__(64) void k_chimc_rules(const uint32_t *__restrict__ pstart, const uint64_t *__restrict__ offsets,
                                                    const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_count, ChimParams P,
                                                    const TsoSlot *__restrict__ slots, const Stretch *__restrict__ st,
                                                    const StretchRes *__restrict__ sres, const uint32_t *__restrict__ read_st,
                                                    const uint8_t *__restrict__ read_nst, const float *__restrict__ e_ne,
                                                    const int *__restrict__ e_nmis, smi_chimera_result *__restrict__ out, uint32_t *__restrict__ dbg) {
    __shared__ RulesLds L;
    const int lane = threadIdx.x;
    const size_t n_list = *list_count;
    const size_t li = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (li >= n_list) return;
    const size_t r = list[li];
    const int verdict = out[r].n_matches;  // left by K-CHIM-A
    const int len = (int)(offsets[r + 1] - offsets[r]);
    int n_m = 0;
    bool overflow = false;
    auto push = [&](int begin, int kind) {
        if (n_m < kMatchCap)
            L.m[n_m][lane] = (begin << 2) | kind;
        else
            overflow = true;
        n_m++;
    };
    if (verdict & SMI_CHIMA_TSO) {
        const int n = slots[li].n;
        if (n > kCap) {
            overflow = true;
            atomicAdd(dbg + 4, 1u);
        } else
            for (int k = 0; k < n; k++) push(slots[li].begin[k], slots[li].kind[k]);
    }
    if (verdict & SMI_CHIMA_PAT) {
        const int nst = read_nst[li];
        if (nst == 255) overflow = true;
        long long prev_start[2] = {-2147483648LL, -2147483648LL};  // prevA_Position / prevT_Position
        const float maxe = (float)P.ad_max;
        for (int k = 0; k < nst && nst != 255; k++) {
            const uint32_t s = read_st[li * kStPerRead + k];
            const StretchRes sr = sres[s];
            if (sr.gate == 0) continue;
            if (sr.gate == 0xFFFFFFFFu) {
                overflow = true;
                continue;
            }
            const Stretch x = st[s];
            // scanForAdapterOrTSOseqKMERsForInternal's position skip, in scan order
            uint32_t accepted = 0;
            int skip = 0, q = 0;
            float best = 3.4028234663852886e+38f;
            for (uint32_t g = sr.gate; g; g &= g - 1, q++) {
                const int i = __builtin_ctz(g);
                if (i + 1 < skip) continue;
                const float ne = e_ne[sr.first + q];
                if (!((float)jround(ne) > maxe)) {
                    accepted |= 1u << i;
                    best = fminf(best, ne);
                }
                int delta = 1;
                if (maxe < ne) {
                    delta = jround(__fsub_rn(ne, maxe)) - 1;
                    if (delta < 1) delta = 1;
                }
                skip = i + 1 + delta;
            }
            if (!accepted) continue;
            // getPosForBestScore(MAX_VALUE): positions sharing the least key, neighbours < 2 apart dropped (L180-183); then nmis <= max (L203)
            uint32_t eq = 0, okm = 0;
            q = 0;
            for (uint32_t g = sr.gate; g; g &= g - 1, q++) {
                const int i = __builtin_ctz(g);
                if (!((accepted >> i) & 1u)) continue;
                if (e_ne[sr.first + q] == best) eq |= 1u << i;
                if (e_nmis[sr.first + q] <= P.ad_max) okm |= 1u << i;
            }
            const uint32_t keep = eq & ~(eq << 1);
            const uint32_t good = keep & okm;
            if (!good) continue;
            const int o1 = __builtin_ctz(good) + 1;
            int start_range;
            if (x.t)
                start_range = x.at_begin - P.bc_umi - 30 - 10;
            else
                start_range = x.at_end + P.bc_umi + 30 + 10 - 30 - 20;
            const int start = x.t ? start_range + o1 - 1 : start_range + 51 - o1;  // L207 / L211
            if (start != 0) {  // lambda$7 L204-207
                const long long lim = prev_start[x.t] + 120;
                prev_start[x.t] = start;
                if ((long long)start > lim) push(start, 2 | (x.t == 0 ? 1 : 0));  // polyA stretch = reverse adapter
            }
        }
    }
    if (n_m > kMatchCap) {  // (overflow is set: the serial kernel redoes the read)
        n_m = kMatchCap;
        atomicAdd(dbg + 3, 1u);
    }
    // ---- split rules (L229-286) ----------------------------------------------------------------------------
    for (int i = 1; i < n_m; i++) {  // stable sort by begin (insertion: the lists are a handful of entries)
        const int v = L.m[i][lane];
        int j = i - 1;
        while (j >= 0 && (L.m[j][lane] >> 2) > (v >> 2)) {
            L.m[j + 1][lane] = L.m[j][lane];
            j--;
        }
        L.m[j + 1][lane] = v;
    }
    int n_kept = 0, kept_pos[3] = {0, 0, 0}, kept_reason[3] = {0, 0, 0}, prev_sp = 0;
    bool have_prev_sp = false;
    auto emit = [&](int reason, int pos) {  // the < 100 filter (L273-281) compares neighbours of the unfiltered list
        const bool drop = have_prev_sp && (pos - prev_sp < 100);
        prev_sp = pos;
        have_prev_sp = true;
        if (!drop) {
            if (n_kept == 0) { kept_pos[0] = pos; kept_reason[0] = reason; }
            else if (n_kept == 1) { kept_pos[1] = pos; kept_reason[1] = reason; }
            else if (n_kept == 2) { kept_pos[2] = pos; kept_reason[2] = reason; }
            n_kept++;
        }
    };
    auto isolated = [&](int m) {
        const int k = L.m[m][lane] & 3, b = L.m[m][lane] >> 2;
        emit((k & 1) ? SMI_SPLIT_REV_ADAPTER : SMI_SPLIT_FWD_ADAPTER, (k & 1) ? b + 25 : b - 25);  // lambda$10
    };
    if (n_m == 1) {
        if (L.m[0][lane] & 2) isolated(0);
    } else if (n_m > 1) {
        int it = 0;
        int prev = it++;
        while (it < n_m && prev >= 0) {
            const int cur = it++;
            const int pk = L.m[prev][lane] & 3, ck = L.m[cur][lane] & 3;
            const int pbeg = L.m[prev][lane] >> 2, cbeg = L.m[cur][lane] >> 2;
            if (cbeg - pbeg > 160) {
                if (pk & 2) isolated(prev);
                prev = cur;
            } else if ((pk & 1) && !(ck & 1)) {
                const int reason = (pk & 2) ? ((ck & 2) ? SMI_SPLIT_RA_FA : SMI_SPLIT_RA_FT)
                                            : ((ck & 2) ? SMI_SPLIT_RT_FA : SMI_SPLIT_RT_FT);  // lambda$11
                emit(reason, pbeg + (cbeg - pbeg) / 2);
                prev = it < n_m ? it++ : -1;
            } else
                prev = cur;
            if (it >= n_m && prev >= 0 && (L.m[prev][lane] & 2)) isolated(prev);  // L263-264
        }
    }
    smi_chimera_result res;
    res.n_split = 0;
    res.pos[0] = res.pos[1] = 0;
    res.reason[0] = res.reason[1] = 0;
    res.flags = 0;
    res.n_matches = n_m;
    if (n_kept > 2) {
        res.flags |= SMI_CHIM_MULTI;  // MULTI_CHIMERIC_READS_DISCARDED | FAILED, read kept whole (L284-286)
    } else {
        res.n_split = n_kept;
        int lastp = 0;
        for (int i = 0; i < n_kept; i++) {
            const int kp = i == 0 ? kept_pos[0] : kept_pos[1];
            res.pos[i] = kp;
            res.reason[i] = (uint8_t)(i == 0 ? kept_reason[0] : kept_reason[1]);
            if (kp < lastp || kp > len) res.flags |= SMI_CHIM_RANGE;  // substring would throw
            lastp = kp;
        }
    }
    if (overflow) {  // over one of this pipeline's caps: the first-generation kernels redo the read from K-CHIM-A's verdict
        res.flags = SMI_CHIM_OVERFLOW;
        res.n_split = 0;
        res.n_matches = verdict;
    }
    out[r] = res;
}

// =================================================================================================================
// K-CHIM-A, second generation (round 4): the read-level filter over the FLAT plane stream.
//
// The first generation gave every read a wave (lane = 32 positions: 41 of 64 lanes busy for a 1.3 kb read) and bounded the
// gated positions of one read and one orientation at a time (two half-empty batches per read).  Here a workgroup takes a
// TILE of 256 consecutive plane words whatever reads they belong to (`own`: read of every plane word): every lane computes
// the gates and the polyA / polyT start condition of its 32 positions, the gated positions of the whole tile are pooled per
// orientation and bounded 256 at a time with all lanes busy.  Besides the verdict per read it leaves what the later stages
// would otherwise compute again: the gate words (`cand`), the positions under the bound (`hotw`), the trigger words (`trig`).
// =================================================================================================================
constexpr int kFlatList = 2048;  // gated positions of one orientation in a tile that are bounded one by one; a tile with more (poly-N, homopolymers)
                                 // counts all of them as possible -- that only adds alignments later

struct FlatLds {
    uint32_t pw[4][260];          // the tile's plane words (+ the first of the next tile)
    uint32_t cmask[2][256];
    uint32_t hot[2][256];
    uint16_t clist[2][kFlatList]; // thread << 5 | bit
    int wsum[2][4];
};

__global__ __launch_bounds__(256) void k_chim_owner(const uint32_t *__restrict__ pstart, const uint64_t *__restrict__ offsets, size_t n, size_t stride,
                                                    uint32_t *__restrict__ own) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < n; r += n_waves) {
        const uint64_t beg = offsets[r], end = offsets[r + 1];
        const int len = (int)(end - beg);
        const size_t w0 = pstart ? (size_t)pstart[r] : plane_start(beg, r);
        const int n_words = (len + 31) >> 5;
        // Without explicit word starts the reads tile the planes (read r + 1 starts where the pad words of read r end): the wave also marks its
        // pad words -- and the last read the rest of the plane -- as nobody's, so the array needs no fill in front of this kernel (a 75 MB
        // memset per chunk of 0.45 M reads).  With explicit starts the caller has filled it.
        const size_t w1 = pstart ? w0 + (size_t)n_words : (r + 1 < n ? plane_start(end, r + 1) : stride);
        for (size_t w = lane; w0 + w < w1; w += 64) own[w0 + w] = w < (size_t)n_words ? (uint32_t)r : 0xFFFFFFFFu;
    }
}

template <int kTsoLen, int PAT>
__global__ __launch_bounds__(256) void k_chima_flat(const uint32_t *__restrict__ planes, size_t stride, size_t n_tiles, const uint32_t *__restrict__ own,
                                                    const uint32_t *__restrict__ pstart, const uint64_t *__restrict__ offsets, FilterParams P,
                                                    uint32_t *__restrict__ cand, uint32_t *__restrict__ hotw, uint32_t *__restrict__ trig,
                                                    uint32_t *__restrict__ verd) {
    __shared__ FlatLds L;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t w = tile * 256 + tid;
        const uint32_t r = w < stride ? own[w] : 0xFFFFFFFFu;
        uint32_t cm[2] = {0, 0}, ma = 0, mt = 0;
        uint32_t lo4[4] = {0, 0, 0, 0}, hi4[4] = {0, 0, 0, 0};
        if (w < stride) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                lo4[c] = planes[c * stride + w];
                hi4[c] = w + 1 < stride ? planes[c * stride + w + 1] : 0u;  // (the last plane ends with the buffer)
            }
        }
        if (r != 0xFFFFFFFFu) {
            const uint64_t beg = offsets[r];
            const int len = (int)(offsets[r + 1] - beg);
            const size_t w0 = pstart ? (size_t)pstart[r] : plane_start(beg, r);
            const int q0 = 32 * (int)(w - w0);
            if (len >= 2 * 70 + 100) {
                uint64_t W[4];
#pragma unroll
                for (int c = 0; c < 4; c++) W[c] = ((uint64_t)hi4[c] << 32) | lo4[c];
                // scan positions q0 + 1 + k, 70 <= position <= len - 70
                const int klo = max(0, 69 - q0), khi = min(31, len - 70 - 1 - q0);
                if (klo <= khi) {
                    const uint32_t keep = (0xFFFFFFFFu << klo) & (0xFFFFFFFFu >> (31 - khi));
                    cm[0] = tso_gate32<kTsoLen, PAT, 0>(W, P) & keep;
                    cm[1] = tso_gate32<kTsoLen, PAT, 1>(W, P) & keep;
                }
                const int first = P.off - 1, stop = len - P.off;
                if (first < stop) {
                    if (P.pat_thr <= 15) {
                        if (q0 < stop && q0 + 32 > first) {
                            const int xb = P.off + P.pat_len - 2;
                            const size_t xw = w0 + (size_t)(xb >> 5);
                            const uint32_t xa = planes[xw], xg = planes[stride + xw], xt = planes[3 * stride + xw];
                            const int extra_a = (int)(((xa & ~xg) >> (xb & 31)) & 1u), extra_t = (int)(((xt & ~xa) >> (xb & 31)) & 1u);
                            ma = at_triggers32(W[0] & ~W[1], extra_a, P.pat_len, P.pat_thr, q0, first, stop);
                            mt = at_triggers32(W[3] & ~W[0], extra_t, P.pat_len, P.pat_thr, q0, first, stop);
                            if (ma | mt) atomicOr(verd + r, (uint32_t)SMI_CHIMA_PAT);
                        }
                    } else if (q0 == 0)
                        atomicOr(verd + r, (uint32_t)SMI_CHIMA_PAT);  // thresholds the bit-sliced counter cannot hold: the walk finds no trigger
                }
                if (q0 == 0 && P.force_all) atomicOr(verd + r, (uint32_t)(SMI_CHIMA_TSO | SMI_CHIMA_PAT));
            }
        }
        if (w < stride) {
            cand[w] = cm[0];
            cand[stride + w] = cm[1];
            trig[w] = ma;
            trig[stride + w] = mt;
        }
        // ---- pool the gated positions of the tile per orientation
#pragma unroll
        for (int c = 0; c < 4; c++) L.pw[c][tid] = lo4[c];
        if (tid == 255) {
#pragma unroll
            for (int c = 0; c < 4; c++) L.pw[c][256] = hi4[c];
        }
        int off[2], tot_w[2];
#pragma unroll
        for (int o = 0; o < 2; o++) {
            off[o] = wave_exscan_i(__popc(cm[o]), lane, tot_w[o]);
            if (lane == 0) L.wsum[o][wv] = tot_w[o];
            L.cmask[o][tid] = cm[o];
            L.hot[o][tid] = P.myers_ok ? 0u : cm[o];  // no bound for this pattern: every gated position counts as possible
        }
        __syncthreads();
        int total[2];
#pragma unroll
        for (int o = 0; o < 2; o++) {
            int base = 0, t = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (k < wv) base += L.wsum[o][k];
                t += L.wsum[o][k];
            }
            total[o] = t;
            off[o] += base;
            if (t > kFlatList)
                L.hot[o][tid] = cm[o];
            else {
                int i = off[o];
                for (uint32_t g = cm[o]; g; g &= g - 1) L.clist[o][i++] = (uint16_t)((tid << 5) | __builtin_ctz(g));
            }
        }
        __syncthreads();
        if (P.myers_ok) {
#pragma unroll
            for (int o = 0; o < 2; o++) {
                if (total[o] > kFlatList) continue;
                for (int base = 0; base < total[o]; base += 256) {
                    if (base + (wv << 6) >= total[o]) break;  // nothing for this wave
                    const int e = base + tid;
                    const bool live = e < total[o];
                    uint32_t V[4] = {0, 0, 0, 0};
                    int t = 0, k = 0;
                    if (live) {
                        const int x = L.clist[o][e];
                        t = x >> 5;
                        k = x & 31;
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const uint64_t x64 = ((uint64_t)L.pw[c][t + 1] << 32) | L.pw[c][t];
                            V[c] = __brev((uint32_t)(x64 >> k)) >> (32 - kTsoLen);
                        }
                    }
                    int b;
                    if constexpr (PAT != 0)
                        b = o ? myers_bound_ct<kTsoLen, PAT, 1>(V, P.lead_max) : myers_bound_ct<kTsoLen, PAT, 0>(V, P.lead_max);
                    else
                        b = myers_bound<kTsoLen>(V, o ? P.rev_idx : P.fwd_idx, P.lead_max);
                    if (live && b <= P.tso_max) atomicOr(&L.hot[o][t], 1u << k);
                }
            }
        }
        __syncthreads();
        if (w < stride) {
            const uint32_t h0 = L.hot[0][tid], h1 = L.hot[1][tid];
            hotw[w] = h0;
            hotw[stride + w] = h1;
            if ((h0 | h1) && r != 0xFFFFFFFFu) atomicOr(verd + r, (uint32_t)SMI_CHIMA_TSO);
        }
        __syncthreads();
    }
}

// verdicts -> results of the cleared reads + the queue of the others (one atomic per 1024 reads: the queue counter is a single address)
__global__ __launch_bounds__(1024) void k_chima_finish(const uint32_t *__restrict__ verd, size_t n, smi_chimera_result *__restrict__ out,
                                                       uint32_t *__restrict__ list, uint32_t *__restrict__ list_count) {
    __shared__ uint32_t wcnt[16], wbase[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t r0 = blockIdx.x * (size_t)1024; r0 < n; r0 += (size_t)gridDim.x * 1024) {
        const size_t r = r0 + threadIdx.x;
        const uint32_t v = r < n ? verd[r] : 0u;
        if (r < n) {
            smi_chimera_result res;
            res.n_split = 0;
            res.pos[0] = res.pos[1] = 0;
            res.reason[0] = res.reason[1] = 0;
            res.flags = 0;
            res.n_matches = (int)v;
            out[r] = res;
        }
        const unsigned long long q = __ballot(v != 0);
        if (lane == 0) wcnt[wv] = (uint32_t)__popcll(q);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int k = 0; k < 16; k++) {
                wbase[k] = t;
                t += wcnt[k];
            }
            const uint32_t b = t ? atomicAdd(list_count, t) : 0u;
            for (int k = 0; k < 16; k++) wbase[k] += b;
        }
        __syncthreads();
        if (v) list[wbase[wv] + __popcll(q & ((1ull << lane) - 1ull))] = (uint32_t)r;
        __syncthreads();
    }
}

// K-CHIM-B SELECT on the words K-CHIM-A (second generation) left: a wave per queued read (sixteen reads per workgroup), a lane per
// plane word.  The closure: a gated position is needed if it is under the bound or something needed lies less than kDmax = 32
// positions behind it, i.e. in the same or the next word -- every lane smears its own needed bits and those of the lane above 31
// positions down and takes the gated ones; repeated until no lane changes (the chains are a few positions long).  The positions
// are appended to the global queue in scan order; the sixteen reads of a workgroup reserve their room with ONE atomic on one of
// kSubQ counters (a counter per read would be 10^5 atomics on one address, which serialise).
__device__ __forceinline__ uint32_t closure_step(uint32_t c, uint32_t hh, uint32_t carry_top, int lane) {
    uint32_t x = hh;
    for (;;) {
        const uint32_t up = __shfl_down(x, 1);
        uint64_t y = ((uint64_t)(lane == 63 ? carry_top : up) << 32) | x;
        y |= y >> 1;
        y |= y >> 2;
        y |= y >> 4;
        y |= y >> 8;
        y |= y >> 16;
        const uint32_t x2 = hh | (c & (uint32_t)y);
        const bool changed = x2 != x;
        x = x2;
        if (!__ballot(changed)) break;
    }
    return x;
}

__global__ __launch_bounds__(1024) void k_chimb_select2(size_t stride, const uint32_t *__restrict__ pstart, const uint64_t *__restrict__ offsets,
                                                        const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_count,
                                                        const smi_chimera_result *__restrict__ out, const uint32_t *__restrict__ cand,
                                                        const uint32_t *__restrict__ hotw, BHead *__restrict__ heads, uint64_t *__restrict__ cand_abs,
                                                        uint32_t *__restrict__ gcounts, uint32_t cap, uint32_t *__restrict__ dbg) {
    static_assert(kDmax == 32, "the smear below covers 31 positions");
    __shared__ uint32_t wtot[16], wbase[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t n_list = *list_count;
    const size_t li = blockIdx.x * (size_t)16 + wv;
    const uint32_t sq = blockIdx.x & (kSubQ - 1);
    const uint32_t sub_cap = cap / kSubQ, sub_base = sq * sub_cap;
    bool active = li < n_list;
    size_t r = 0, w0 = 0;
    int nw = 0;
    if (active) {
        r = list[li];
        active = (out[r].n_matches & SMI_CHIMA_TSO) != 0;
    }
    if (active) {
        const uint64_t beg = offsets[r];
        nw = ((int)(offsets[r + 1] - beg) + 31) >> 5;
        w0 = pstart ? (size_t)pstart[r] : plane_start(beg, r);
    }
    const uint64_t bit_base = (uint64_t)w0 * 32u;
    // ---- count
    uint32_t nd[2] = {0, 0}, n_need[2] = {0, 0};
    int offl[2] = {0, 0};
    if (active && nw <= 64) {  // reads up to 2048 bases: the needed words stay in registers
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const uint32_t c = lane < nw ? cand[(size_t)o * stride + w0 + lane] : 0u;
            const uint32_t hh = lane < nw ? hotw[(size_t)o * stride + w0 + lane] : 0u;
            nd[o] = closure_step(c, hh, 0u, lane);
            int tot;
            offl[o] = wave_exscan_i(__popc(nd[o]), lane, tot);
            n_need[o] = (uint32_t)tot;
        }
    } else if (active) {  // longer reads: 64 words at a time from the right, the lowest needed word of the chunk above carried along
        for (int o = 0; o < 2; o++) {
            uint32_t carry = 0;
            for (int ch = ((nw + 63) >> 6) - 1; ch >= 0; ch--) {
                const int w = 64 * ch + lane;
                const uint32_t c = w < nw ? cand[(size_t)o * stride + w0 + w] : 0u;
                const uint32_t hh = w < nw ? hotw[(size_t)o * stride + w0 + w] : 0u;
                const uint32_t x = closure_step(c, hh, carry, lane);
                carry = __shfl(x, 0);
                int tot;
                (void)wave_exscan_i(__popc(x), lane, tot);
                n_need[o] += (uint32_t)tot;
            }
        }
    }
    // ---- one reservation per workgroup
    const uint32_t total = n_need[0] + n_need[1];
    if (lane == 0) wtot[wv] = total;
    __syncthreads();
    __shared__ uint32_t hole_lo, hole_hi;
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int k = 0; k < 16; k++) {
            wbase[k] = t;
            t += wtot[k];
        }
        const uint32_t b = t ? atomicAdd(gcounts + sq, t) : 0u;
        const bool fits = (unsigned long long)b + t <= sub_cap;
        for (int k = 0; k < 16; k++) wbase[k] = fits ? sub_base + b + wbase[k] : 0xFFFFFFFFu;
        // the reservation of a workgroup that does not fit is given up as a whole; what it took of the region is marked so that the
        // alignment kernel passes over it
        hole_lo = fits ? 0u : min(b, sub_cap);
        hole_hi = fits ? 0u : sub_cap;
    }
    __syncthreads();
    for (uint32_t e = hole_lo + threadIdx.x; e < hole_hi; e += 1024) cand_abs[sub_base + e] = ~0ull;
    if (!active) return;
    BHead h;
    h.first = h.n0 = h.n1 = h.flags = 0;
    const uint32_t first = wbase[wv];
    if (first == 0xFFFFFFFFu) {  // out of queue room: the first-generation kernels take the read
        h.flags = 1;
        if (lane == 0) {
            heads[li] = h;
            atomicAdd(dbg + 6, 1u);
        }
        return;
    }
    // ---- write, in scan order
    if (nw <= 64) {
#pragma unroll
        for (int o = 0; o < 2; o++) {
            uint32_t idx = first + (o ? n_need[0] : 0u) + (uint32_t)offl[o];
            for (uint32_t g = nd[o]; g; g &= g - 1) cand_abs[idx++] = (bit_base + (uint64_t)(32 * lane + __builtin_ctz(g))) | ((uint64_t)o << 63);
        }
    } else {
        for (int o = 0; o < 2; o++) {
            uint32_t carry = 0, below = n_need[o];  // needed positions in the chunks below the current one
            const uint32_t run = first + (o ? n_need[0] : 0u);
            for (int ch = ((nw + 63) >> 6) - 1; ch >= 0; ch--) {
                const int w = 64 * ch + lane;
                const uint32_t c = w < nw ? cand[(size_t)o * stride + w0 + w] : 0u;
                const uint32_t hh = w < nw ? hotw[(size_t)o * stride + w0 + w] : 0u;
                const uint32_t x = closure_step(c, hh, carry, lane);
                carry = __shfl(x, 0);
                int tot;
                const int ol = wave_exscan_i(__popc(x), lane, tot);
                below -= (uint32_t)tot;
                uint32_t idx = run + below + (uint32_t)ol;
                for (uint32_t g = x; g; g &= g - 1) cand_abs[idx++] = (bit_base + (uint64_t)(32 * w + __builtin_ctz(g))) | ((uint64_t)o << 63);
            }
        }
    }
    h.first = first;
    h.n0 = n_need[0];
    h.n1 = n_need[1];
    if (lane == 0) heads[li] = h;
}

// queue of the reads whose result carries SMI_CHIM_OVERFLOW
__global__ void k_collect_overflow(const smi_chimera_result *__restrict__ out, const uint32_t *__restrict__ list,
                                   const uint32_t *__restrict__ list_count, uint32_t *__restrict__ over, uint32_t *__restrict__ over_count) {
    const uint32_t n = *list_count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (out[list[i]].flags & SMI_CHIM_OVERFLOW) over[atomicAdd(over_count, 1u)] = list[i];
}

// ---- fragment offsets: read i with k split positions becomes k + 1 consecutive records of the same byte buffer ----
__global__ void k_frag_counts(const smi_chimera_result *__restrict__ chim, size_t n, uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t sh[256];
    const size_t i = blockIdx.x * (size_t)1024 + threadIdx.x * 4;
    uint32_t s = 0;
    for (int k = 0; k < 4; k++)
        if (i + k < n) s += 1u + (uint32_t)chim[i + k].n_split;
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0];
}

__global__ void k_scan_block_sums(uint32_t *__restrict__ block_sums, size_t n_blocks, uint64_t *__restrict__ total) {
    // one block: exclusive scan in place (n_blocks is ~ n / 1024)
    __shared__ uint64_t carry;
    __shared__ uint32_t sh[1024];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (size_t base = 0; base < n_blocks; base += 1024) {
        const size_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sums[i] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const uint32_t y = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += y;
            __syncthreads();
        }
        if (i < n_blocks) block_sums[i] = (uint32_t)(carry + sh[threadIdx.x] - v);
        __syncthreads();
        if (threadIdx.x == 0) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void k_frag_offsets(const smi_chimera_result *__restrict__ chim, const uint64_t *__restrict__ offsets, size_t n,
                               const uint32_t *__restrict__ block_sums, uint64_t *__restrict__ frag_offsets,
                               uint32_t *__restrict__ frag_src) {
    __shared__ uint32_t sh[256];
    const size_t i = blockIdx.x * (size_t)1024 + threadIdx.x * 4;
    uint32_t cnt[4], s = 0;
    for (int k = 0; k < 4; k++) {
        cnt[k] = i + k < n ? 1u + (uint32_t)chim[i + k].n_split : 0u;
        s += cnt[k];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t y = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
        __syncthreads();
        sh[threadIdx.x] += y;
        __syncthreads();
    }
    uint64_t f = (uint64_t)block_sums[blockIdx.x] + sh[threadIdx.x] - s;
    for (int k = 0; k < 4; k++) {
        if (i + k >= n) break;
        const smi_chimera_result c = chim[i + k];
        const uint64_t beg = offsets[i + k];
        for (uint32_t j = 0; j < cnt[k]; j++) {
            frag_offsets[f] = j == 0 ? beg : beg + (uint64_t)c.pos[j - 1];
            if (frag_src) frag_src[f] = (uint32_t)(((i + k) << 2) | j);
            f++;
        }
        if (i + k == n - 1) frag_offsets[f] = offsets[n];
    }
}

static int thr_count(int len, float frac) {
    for (int k = 0; k <= len + 1; k++)
        if (!((float)k / (float)len < frac)) return k;
    return len + 2;
}

static uint32_t code_of(char c) {
    switch (c) {
    case 'A': case 'a': return 1;
    case 'G': case 'g': return 2;
    case 'C': case 'c': return 4;
    case 'T': case 't': return 8;
    default: return 15;
    }
}

size_t read_planes_stride(uint64_t total_bases, size_t n) { return (size_t)(total_bases >> 5) + kPadWords * (n + 1) + 8; }

int launch_pack_reads(smi_ctx *, const uint8_t *d_reads, const uint64_t *d_offsets, const uint64_t *d_starts, size_t n,
                      uint64_t total_bases, uint32_t *d_planes, hipStream_t s) {
    if (!n) return SMI_OK;
    const bool by_wave = getenv("SMI_PACKR_WAVE") != nullptr;  // the wave-per-read kernel of rounds 2 - 4 (cross-checks)
    if (by_wave) {
        const unsigned grid = (unsigned)std::min<size_t>((n + 3) / 4, 256 * 32);
        hipLaunchKernelGGL(d_starts ? k_pack_reads<true> : k_pack_reads<false>, dim3(grid), dim3(256), 0, s, d_reads, d_offsets, d_starts, n, read_planes_stride(total_bases, n),
                           total_bases, d_planes);
    } else {
        const size_t words = plane_start(total_bases, n);
        // 4,096 workgroups: measured (0.9 M reads: 512 / 1,024 / 2,048 / 4,096 / 16,384 / 65,536 workgroups -> 1.02 / 0.77 / 0.65 / 0.62 / 0.65 / 0.86 ms; a wave's one
        // binary search wants a long run of tiles behind it, the memory system wants enough waves)
        const unsigned grid = (unsigned)std::min<size_t>((words + 255) / 256, 4096);
        hipLaunchKernelGGL(d_starts ? k_pack_reads_flat<true> : k_pack_reads_flat<false>, dim3(grid), dim3(256), 0, s, d_reads, d_offsets, d_starts, n,
                           read_planes_stride(total_bases, n), total_bases, d_planes);
    }
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

// SMI_CHIM_SYNC=1 (debugging): wait for every kernel of the splitter and name the one that failed
static bool chim_sync_on() {
    static const bool on = getenv("SMI_CHIM_SYNC") != nullptr;
    return on;
}
#define SMI_CHIM_CHECK(NAME)                                                                                   \
    do {                                                                                                       \
        if (chim_sync_on()) {                                                                                    \
            const hipError_t e_ = hipStreamSynchronize(s);                                                     \
            if (e_ != hipSuccess) return hip_fail(e_, "K-CHIM kernel " NAME);                                  \
        }                                                                                                      \
    } while (0)

int launch_chimera(smi_ctx *ctx, const uint32_t *d_planes, const uint64_t *d_offsets, size_t n, uint64_t total_bases,
                   const smi_chimera_config *cfg, smi_chimera_result *d_out, hipStream_t s, const uint32_t *d_pstart, size_t stride_override) {
    if (!n) return SMI_OK;
    const int tl = (int)std::strlen(cfg->tso_complete), al = (int)std::strlen(cfg->adapter_complete);
    if (!((tl == 27 && al == 22) || (tl == 22 && al == 25))) {
        set_error("smi_chimera_device: this build handles pattern lengths 27 + 22 (3' barcoding) and 22 + 25 (5' barcoding), "
                  "the shipped config.xml values");
        return SMI_ERR_INVALID;
    }
    ChimParams P;
    for (int i = 0; i < tl; i++) {
        P.tso4[0][i] = code_of(cfg->tso_complete[i]);
        const uint32_t b = code_of(cfg->tso_complete[tl - 1 - i]);
        P.tso4[1][i] = ((b & 1u) << 3) | ((b & 8u) >> 3) | ((b & 2u) << 1) | ((b & 4u) >> 1);
    }
    for (int i = 0; i < al; i++) P.ad4[i] = code_of(cfg->adapter_complete[i]);
    P.tso_max = cfg->tso_max_errors;
    P.ad_max = cfg->adapter_max_errors;
    P.pat_len = cfg->internal_pat_len;
    P.pat_thr = thr_count(cfg->internal_pat_len, cfg->internal_pat_frac);
    P.off = cfg->window_polya + 70;
    P.bc_umi = cfg->bc_umi_len;
    auto plane_of = [](uint32_t code) { return code == 1 ? 0 : code == 2 ? 1 : code == 4 ? 2 : code == 8 ? 3 : -1; };
    P.prefilter = 1;
    P.myers_ok = 1;
    for (int j = 0; j < tl; j++) {
        const int f = plane_of(P.tso4[0][tl - 1 - j]), r = plane_of(P.tso4[0][j]);
        if (f < 0 || r < 0) P.myers_ok = 0;
        P.tso_fwd_idx[j] = (uint8_t)(f < 0 ? 0 : f);
        P.tso_rev_idx[j] = (uint8_t)(r < 0 ? 0 : 3 - r);  // complement plane: A <-> T, G <-> C
    }
    for (int j = 0; j < al; j++) {
        const int a = plane_of(P.ad4[al - 1 - j]);
        if (a < 0) P.myers_ok = 0;
        P.ad_idx[j] = (uint8_t)(a < 0 ? 0 : a);
    }
    // SMI_CHIM_ABLATE (timing of the kernel's parts, wrong results by construction): measurement builds only (make MEASURE=1)
#ifdef SMI_MEASURE
    P.ablate = getenv("SMI_CHIM_ABLATE") ? atoi(getenv("SMI_CHIM_ABLATE")) : 0;
#else
    P.ablate = 0;
    if (getenv("SMI_CHIM_ABLATE") && atoi(getenv("SMI_CHIM_ABLATE")) != 0) {
        set_error("smi_chimera_device: SMI_CHIM_ABLATE is set, which yields wrong results by construction; it is honoured by measurement builds only (make MEASURE=1)");
        return SMI_ERR_INVALID;
    }
#endif
    const bool no_filter = getenv("SMI_CHIM_NO_PREFILTER") != nullptr;  // measurement / cross-check switch: every read takes the exact path
    if (no_filter) P.myers_ok = 0;
    // lead <= nErrors / 1.1 (see myers_bound); one more for safety -- a larger range only weakens the filter
    P.tso_lead_max = std::min(tl, (int)(((float)P.tso_max + 0.5f) / 1.1f) + 1);
    P.ad_lead_max = std::min(al, (int)(((float)P.ad_max + 0.5f) / 1.1f) + 1);
    // queue of the reads K-CHIM-A could not clear, and the hand-over slots of K-CHIM-B: device scratch of the context, grow-only
    const size_t list_bytes = (2 * n + 64) * sizeof(uint32_t);
    if (ctx->chim_list_bytes < list_bytes) {
        if (ctx->chim_list) (void)hipFree(ctx->chim_list);
        ctx->chim_list = nullptr;
        ctx->chim_list_bytes = 0;
        SMI_HIP(hipMalloc(&ctx->chim_list, list_bytes));
        ctx->chim_list_bytes = list_bytes;
    }
    uint32_t *d_count = ctx->chim_list, *d_list = ctx->chim_list + 16;
    SMI_HIP(hipMemsetAsync(d_count, 0, 64, s));
    if (int rc = time_begin(ctx, SMI_K_CHIMERA, s)) return rc;
    FilterParams F;
    std::memset(&F, 0, sizeof F);
    for (int j = 0; j < tl; j++) {
        F.gate_idx[0][j] = (uint8_t)std::max(0, plane_of(P.tso4[0][j]));
        F.gate_idx[1][j] = (uint8_t)std::max(0, plane_of(P.tso4[1][j]));
        F.fwd_idx[j] = P.tso_fwd_idx[j];
        F.rev_idx[j] = P.tso_rev_idx[j];
        F.tso4[0][j] = (uint8_t)P.tso4[0][j];
        F.tso4[1][j] = (uint8_t)P.tso4[1][j];
    }
    F.tso_max = P.tso_max;
    F.lead_max = P.tso_lead_max;
    F.myers_ok = P.myers_ok;
    F.force_all = no_filter ? 1 : 0;
    F.pat_len = P.pat_len;
    F.pat_thr = P.pat_thr;
    F.off = P.off;
    const size_t st = stride_override ? stride_override : read_planes_stride(total_bases, n);
    const bool v1 = getenv("SMI_CHIM_V1") != nullptr;        // cross-check switch: the first-generation kernels throughout (one wave per read)
    const bool a1 = v1 || getenv("SMI_CHIM_A1") != nullptr;  // cross-check switch: first-generation filter + the wave-per-read select / trigger kernels
    const bool generic = getenv("SMI_CHIM_GENERIC") != nullptr || !P.myers_ok;
    const bool shipped3 = tl == 27 && !std::strcmp(cfg->tso_complete, PatSeq<1>::s) && !generic;
    const bool shipped5 = tl == 22 && !std::strcmp(cfg->tso_complete, PatSeq<2>::s) && !generic;
    uint32_t *d_cand_w = nullptr, *d_hot_w = nullptr, *d_trig_w = nullptr;
    if (!a1) {
        // flat scratch: own[st] | cand[2 st] | hot[2 st] | trig[2 st] | verd[n]
        const size_t flat_bytes = (7 * st + n + 64) * sizeof(uint32_t);
        if (ctx->chim_flat_bytes < flat_bytes) {
            if (ctx->chim_flat) (void)hipFree(ctx->chim_flat);
            ctx->chim_flat = nullptr;
            ctx->chim_flat_bytes = 0;
            const size_t want = flat_bytes + flat_bytes / 8;
            SMI_HIP(hipMalloc(&ctx->chim_flat, want));
            ctx->chim_flat_bytes = want;
        }
        uint32_t *d_own = static_cast<uint32_t *>(ctx->chim_flat);
        d_cand_w = d_own + st;
        d_hot_w = d_cand_w + 2 * st;
        d_trig_w = d_hot_w + 2 * st;
        uint32_t *d_verd = d_trig_w + 2 * st;
        if (d_pstart || n == 0) SMI_HIP(hipMemsetAsync(d_own, 0xFF, st * sizeof(uint32_t), s));
        SMI_HIP(hipMemsetAsync(d_verd, 0, n * sizeof(uint32_t), s));
        { hipLaunchKernelGGL(k_chim_owner, dim3((unsigned)std::min<size_t>((n + 3) / 4, 256 * 32)), dim3(256), 0, s, d_pstart, d_offsets, n, st, d_own); SMI_CHIM_CHECK("k_chim_owner"); }
        const size_t n_tiles = (st + 255) / 256;
        const unsigned gridF = (unsigned)std::min<size_t>(n_tiles, 256 * 8);
        if (shipped3)
            { hipLaunchKernelGGL((k_chima_flat<27, 1>), dim3(gridF), dim3(256), 0, s, d_planes, st, n_tiles, d_own, d_pstart, d_offsets, F, d_cand_w, d_hot_w, d_trig_w, d_verd); SMI_CHIM_CHECK("k_chima_flat"); }
        else if (shipped5)
            { hipLaunchKernelGGL((k_chima_flat<22, 2>), dim3(gridF), dim3(256), 0, s, d_planes, st, n_tiles, d_own, d_pstart, d_offsets, F, d_cand_w, d_hot_w, d_trig_w, d_verd); SMI_CHIM_CHECK("k_chima_flat"); }
        else if (tl == 27)
            { hipLaunchKernelGGL((k_chima_flat<27, 0>), dim3(gridF), dim3(256), 0, s, d_planes, st, n_tiles, d_own, d_pstart, d_offsets, F, d_cand_w, d_hot_w, d_trig_w, d_verd); SMI_CHIM_CHECK("k_chima_flat"); }
        else
            { hipLaunchKernelGGL((k_chima_flat<22, 0>), dim3(gridF), dim3(256), 0, s, d_planes, st, n_tiles, d_own, d_pstart, d_offsets, F, d_cand_w, d_hot_w, d_trig_w, d_verd); SMI_CHIM_CHECK("k_chima_flat"); }
        { hipLaunchKernelGGL(k_chima_finish, dim3((unsigned)std::min<size_t>((n + 1023) / 1024, 256 * 8)), dim3(1024), 0, s, d_verd, n, d_out, d_list, d_count); SMI_CHIM_CHECK("k_chima_finish"); }
        SMI_HIP(hipGetLastError());
    } else {
        const unsigned gridA = (unsigned)std::min<size_t>((n + 3) / 4, 256 * 32);
        if (shipped3)
            { hipLaunchKernelGGL((k_chim_tso_filter<27, 1>), dim3(gridA), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, n, F, d_out, d_list, d_count); SMI_CHIM_CHECK("k_chim_tso_filter"); }
        else if (shipped5)
            { hipLaunchKernelGGL((k_chim_tso_filter<22, 2>), dim3(gridA), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, n, F, d_out, d_list, d_count); SMI_CHIM_CHECK("k_chim_tso_filter"); }
        else if (tl == 27)
            { hipLaunchKernelGGL((k_chim_tso_filter<27, 0>), dim3(gridA), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, n, F, d_out, d_list, d_count); SMI_CHIM_CHECK("k_chim_tso_filter"); }
        else
            { hipLaunchKernelGGL((k_chim_tso_filter<22, 0>), dim3(gridA), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, n, F, d_out, d_list, d_count); SMI_CHIM_CHECK("k_chim_tso_filter"); }
        SMI_HIP(hipGetLastError());
    }
    // the queue length decides the size of the hand-over slots (one small copy + sync per launch)
    uint32_t n_list = 0;
    {
        uint32_t *pw = static_cast<uint32_t *>(pin_words(ctx));
        SMI_HIP(hipMemcpyAsync(pw ? pw : &n_list, d_count, 4, hipMemcpyDeviceToHost, s));
        SMI_HIP(hipStreamSynchronize(s));
        if (pw) n_list = *pw;
    }
#ifdef SMI_MEASURE
    if (getenv("SMI_CHIM_STATS")) {  // how many reads K-CHIM-A queues, and for what
        std::vector<smi_chimera_result> h(n);
        SMI_HIP(hipMemcpy(h.data(), d_out, n * sizeof(smi_chimera_result), hipMemcpyDeviceToHost));
        size_t n_tso = 0, n_pat = 0, n_both = 0;
        for (size_t i = 0; i < n; i++) {
            const int v = h[i].n_matches;
            n_tso += (v & SMI_CHIMA_TSO) != 0;
            n_pat += (v & SMI_CHIMA_PAT) != 0;
            n_both += v == (SMI_CHIMA_TSO | SMI_CHIMA_PAT);
        }
        fprintf(stderr, "[chim stats] reads %zu queued %u tso %zu pat %zu both %zu\n", n, n_list, n_tso, n_pat, n_both);
    }
#endif
    if (n_list) {
        const size_t slot_bytes = (size_t)n_list * sizeof(TsoSlot);
        if (ctx->chim_slot_bytes < slot_bytes) {
            if (ctx->chim_slots) (void)hipFree(ctx->chim_slots);
            ctx->chim_slots = nullptr;
            ctx->chim_slot_bytes = 0;
            const size_t want = slot_bytes + slot_bytes / 4;
            SMI_HIP(hipMalloc(&ctx->chim_slots, want));
            ctx->chim_slot_bytes = want;
        }
        TsoSlot *d_slots = static_cast<TsoSlot *>(ctx->chim_slots);
        const unsigned grid = (unsigned)std::min<size_t>(((size_t)n_list + 3) / 4, 256 * 16);
        if (!(P.ablate & 16)) {
            if (!v1) {
                // scratch of the second-generation pipelines, one grow-only buffer:
                //   B: heads | queue of positions (u64) | error counts (f32)
                //   C: trigger words (2 x stride) | stretches | their results | per-read stretch lists | alignment queue | its results
                auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
                const size_t cap = (std::max<size_t>((size_t)n_list * 48, (size_t)1 << 16) + kSubQ - 1) / kSubQ * kSubQ;
                const size_t st_cap = std::max<size_t>((size_t)n_list * 4, (size_t)1 << 12);
                const size_t e_cap = std::max<size_t>((size_t)n_list * 8, (size_t)1 << 12);
                size_t off = 0;
                auto take = [&](size_t bytes) { const size_t o = off; off += up(bytes); return o; };
                const size_t o_bcnt = take(kSubQ * 4);
                const size_t o_heads = take((size_t)n_list * sizeof(BHead)), o_cand = take(cap * 8), o_ne = take(cap * 4);
                const size_t o_trig = take(a1 ? 2 * st * 4 : 0), o_st = take(st_cap * sizeof(Stretch)), o_sres = take(st_cap * sizeof(StretchRes));
                const size_t o_rst = take((size_t)n_list * kStPerRead * 4), o_rn = take(n_list), o_ent = take(e_cap * sizeof(AdEntry));
                const size_t o_ene = take(e_cap * 4), o_enm = take(e_cap * 4);
                if (ctx->chim_work_bytes < off) {
                    if (ctx->chim_work) (void)hipFree(ctx->chim_work);
                    ctx->chim_work = nullptr;
                    ctx->chim_work_bytes = 0;
                    const size_t want = off + off / 4;
                    SMI_HIP(hipMalloc(&ctx->chim_work, want));
                    ctx->chim_work_bytes = want;
                }
                char *wk = static_cast<char *>(ctx->chim_work);
                BHead *d_heads = reinterpret_cast<BHead *>(wk + o_heads);
                uint64_t *d_cand = reinterpret_cast<uint64_t *>(wk + o_cand);
                float *d_ne = reinterpret_cast<float *>(wk + o_ne);
                uint32_t *d_trig = a1 ? reinterpret_cast<uint32_t *>(wk + o_trig) : d_trig_w;
                Stretch *d_st = reinterpret_cast<Stretch *>(wk + o_st);
                StretchRes *d_sres = reinterpret_cast<StretchRes *>(wk + o_sres);
                uint32_t *d_rst = reinterpret_cast<uint32_t *>(wk + o_rst);
                uint8_t *d_rn = reinterpret_cast<uint8_t *>(wk + o_rn);
                AdEntry *d_ent = reinterpret_cast<AdEntry *>(wk + o_ent);
                float *d_ene = reinterpret_cast<float *>(wk + o_ene);
                int *d_enm = reinterpret_cast<int *>(wk + o_enm);
                uint32_t *d_gcount = a1 ? d_count + 2 : reinterpret_cast<uint32_t *>(wk + o_bcnt);  // (second generation: kSubQ counters)
                if (!a1) SMI_HIP(hipMemsetAsync(d_gcount, 0, kSubQ * 4, s));
                uint32_t *d_stcount = d_count + 3, *d_ecount = d_count + 4;  // zeroed with the queue counters above
                uint32_t *d_dbg = d_count + 8;  // why reads went to the serial kernel: 0 stretches per read, 1 stretch queue, 2 alignment queue, 3 matches per read, 4 TSO slot, 5 gated positions per read, 6 position queue, 7 accepted positions
                const unsigned grid_flat = 256 * 8;
                const unsigned grid_lane = (unsigned)(((size_t)n_list + 63) / 64);
                // B (select -> align -> fold) and C (walk -> gate -> align) share nothing but the filter's words and the queue: C runs on the
                // context's side stream beside B -- both are chains of short kernels that leave most of the chip idle (C's walk is a lane per
                // queued read: under one wave per SIMD) -- and the rules kernel waits for both.  SMI_CHIM_SYNC keeps everything on one stream.
                const bool use_side = !chim_sync_on();
                if (use_side && !ctx->side_stream) {
                    SMI_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
                    SMI_HIP(hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
                    SMI_HIP(hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming));
                }
                hipStream_t sc = use_side ? ctx->side_stream : s;
                if (use_side) {
                    SMI_HIP(hipEventRecord(ctx->side_fork, s));
                    SMI_HIP(hipStreamWaitEvent(sc, ctx->side_fork, 0));
                }
                // (advisor, round 5) an error return between here and the join must not leave chain C running on arena buffers the caller reuses or
                // frees: every such exit drains the side stream first
                SideStreamGuard side_guard{sc, use_side};
                // ---- B: select -> align -> fold
                if (!a1)
                    { hipLaunchKernelGGL(k_chimb_select2, dim3((unsigned)(((size_t)n_list + 15) / 16)), dim3(1024), 0, s, st, d_pstart, d_offsets, d_list, d_count, d_out, d_cand_w, d_hot_w, d_heads, d_cand, d_gcount,
                                       (uint32_t)cap, d_dbg); SMI_CHIM_CHECK("k_chimb_select2"); }
                else if (shipped3)
                    { hipLaunchKernelGGL((k_chimb_select<27, 1>), dim3(grid), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, d_list, d_count, F, d_out, d_heads, d_cand, d_gcount, (uint32_t)cap, d_dbg); SMI_CHIM_CHECK("k_chimb_select"); }
                else if (shipped5)
                    { hipLaunchKernelGGL((k_chimb_select<22, 2>), dim3(grid), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, d_list, d_count, F, d_out, d_heads, d_cand, d_gcount, (uint32_t)cap, d_dbg); SMI_CHIM_CHECK("k_chimb_select"); }
                else if (tl == 27)
                    { hipLaunchKernelGGL((k_chimb_select<27, 0>), dim3(grid), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, d_list, d_count, F, d_out, d_heads, d_cand, d_gcount, (uint32_t)cap, d_dbg); SMI_CHIM_CHECK("k_chimb_select"); }
                else
                    { hipLaunchKernelGGL((k_chimb_select<22, 0>), dim3(grid), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, d_list, d_count, F, d_out, d_heads, d_cand, d_gcount, (uint32_t)cap, d_dbg); SMI_CHIM_CHECK("k_chimb_select"); }
                // ---- C: triggers -> walk
                if (a1) { hipLaunchKernelGGL(k_chimc_trig, dim3(grid), dim3(256), 0, sc, d_planes, st, d_pstart, d_offsets, d_list, d_count, P, d_out, d_trig); SMI_CHIM_CHECK("k_chimc_trig"); }
                { hipLaunchKernelGGL(k_chimc_walk, dim3(grid_lane), dim3(64), 0, sc, d_planes, st, d_pstart, d_offsets, d_list, d_count, P, d_out, d_trig, d_st, d_stcount,
                                   (uint32_t)st_cap, d_rst, d_rn, d_dbg); SMI_CHIM_CHECK("k_chimc_walk"); }
                if (tl == 27) {
                    { hipLaunchKernelGGL((k_chimb_align<27>), a1 ? dim3(grid_flat) : dim3(32, kSubQ), dim3(256), 0, s, d_planes, st, d_cand, d_gcount, (uint32_t)cap, P, d_ne); SMI_CHIM_CHECK("k_chimb_align"); }
                    { hipLaunchKernelGGL((k_chimb_fold<27>), dim3(grid), dim3(256), 0, s, d_pstart, d_offsets, d_list, d_count, P.tso_max, d_out, d_heads, d_cand, d_ne, d_slots, d_dbg); SMI_CHIM_CHECK("k_chimb_fold"); }
                    { hipLaunchKernelGGL((k_chimc_gate<22>), dim3(grid_flat), dim3(256), 0, sc, d_planes, st, d_pstart, d_offsets, d_list, P, d_st, d_stcount, (uint32_t)st_cap, d_sres,
                                       d_ent, d_ecount, (uint32_t)e_cap, d_dbg); SMI_CHIM_CHECK("k_chimc_gate"); }
                    { hipLaunchKernelGGL((k_chimc_align<22>), dim3(grid_flat), dim3(256), 0, sc, d_planes, st, d_pstart, d_offsets, d_list, P, d_st, d_ent, d_ecount, (uint32_t)e_cap, d_ene, d_enm); SMI_CHIM_CHECK("k_chimc_align"); }
                } else {
                    { hipLaunchKernelGGL((k_chimb_align<22>), a1 ? dim3(grid_flat) : dim3(32, kSubQ), dim3(256), 0, s, d_planes, st, d_cand, d_gcount, (uint32_t)cap, P, d_ne); SMI_CHIM_CHECK("k_chimb_align"); }
                    { hipLaunchKernelGGL((k_chimb_fold<22>), dim3(grid), dim3(256), 0, s, d_pstart, d_offsets, d_list, d_count, P.tso_max, d_out, d_heads, d_cand, d_ne, d_slots, d_dbg); SMI_CHIM_CHECK("k_chimb_fold"); }
                    { hipLaunchKernelGGL((k_chimc_gate<25>), dim3(grid_flat), dim3(256), 0, sc, d_planes, st, d_pstart, d_offsets, d_list, P, d_st, d_stcount, (uint32_t)st_cap, d_sres,
                                       d_ent, d_ecount, (uint32_t)e_cap, d_dbg); SMI_CHIM_CHECK("k_chimc_gate"); }
                    { hipLaunchKernelGGL((k_chimc_align<25>), dim3(grid_flat), dim3(256), 0, sc, d_planes, st, d_pstart, d_offsets, d_list, P, d_st, d_ent, d_ecount, (uint32_t)e_cap, d_ene, d_enm); SMI_CHIM_CHECK("k_chimc_align"); }
                }
                if (use_side) {
                    SMI_HIP(hipGetLastError());
                    SMI_HIP(hipEventRecord(ctx->side_join, sc));
                    SMI_HIP(hipStreamWaitEvent(s, ctx->side_join, 0));
                    side_guard.armed = false;  // s is ordered behind the side stream from here on
                }
                { hipLaunchKernelGGL(k_chimc_rules, dim3(grid_lane), dim3(64), 0, s, d_pstart, d_offsets, d_list, d_count, P, d_slots, d_st, d_sres, d_rst, d_rn, d_ene, d_enm, d_out, d_dbg); SMI_CHIM_CHECK("k_chimc_rules"); }
                // second chance for the reads over a cap of the lane-per-read kernels (one in 10^5): the first-generation kernels, one wave per read,
                // with their caps of 64; what is over those too goes to the serial kernel below.  The usual chunk has none: the two kernels then
                // find an empty list (they take its length from the device, 64 workgroups each: ~ 10 us together) -- reading the length back
                // first, to skip them, cost a round trip to the host of ~ 45 us per chunk
                uint32_t *d_over2_count = d_count + 5, *d_over2 = d_list + n;
                { hipLaunchKernelGGL(k_collect_overflow, dim3(64), dim3(256), 0, s, d_out, d_list, d_count, d_over2, d_over2_count); SMI_CHIM_CHECK("k_collect_overflow"); }
#ifdef SMI_MEASURE
                if (getenv("SMI_CHIM_STATS")) {
                    uint32_t h[16];
                    SMI_HIP(hipMemcpy(h, d_count, 64, hipMemcpyDeviceToHost));
                    fprintf(stderr, "[chim stats] queue %u second chance %u stretches %u ad-entries %u | caps hit: st/read %u st-queue %u ad-queue %u matches %u slot %u gated %u pos-queue %u accepted %u\n",
                            h[0], h[5], h[3], h[4], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
                }
#endif
                if (tl == 27) {
                    { hipLaunchKernelGGL((k_chimera<27, 22, 1>), dim3(64), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, d_over2, d_over2_count, P, d_slots, d_out); SMI_CHIM_CHECK("k_chimera"); }
                    { hipLaunchKernelGGL((k_chimera<27, 22, 2>), dim3(64), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, d_over2, d_over2_count, P, d_slots, d_out); SMI_CHIM_CHECK("k_chimera"); }
                } else {
                    { hipLaunchKernelGGL((k_chimera<22, 25, 1>), dim3(64), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, d_over2, d_over2_count, P, d_slots, d_out); SMI_CHIM_CHECK("k_chimera"); }
                    { hipLaunchKernelGGL((k_chimera<22, 25, 2>), dim3(64), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, d_over2, d_over2_count, P, d_slots, d_out); SMI_CHIM_CHECK("k_chimera"); }
                }
            } else if (tl == 27) {
                { hipLaunchKernelGGL((k_chimera<27, 22, 1>), dim3(grid), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, d_list, d_count, P, d_slots, d_out); SMI_CHIM_CHECK("k_chimera"); }
                { hipLaunchKernelGGL((k_chimera<27, 22, 2>), dim3(grid), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, d_list, d_count, P, d_slots, d_out); SMI_CHIM_CHECK("k_chimera"); }
            } else {
                { hipLaunchKernelGGL((k_chimera<22, 25, 1>), dim3(grid), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, d_list, d_count, P, d_slots, d_out); SMI_CHIM_CHECK("k_chimera"); }
                { hipLaunchKernelGGL((k_chimera<22, 25, 2>), dim3(grid), dim3(256), 0, s, d_planes, st, d_pstart, d_offsets, d_list, d_count, P, d_slots, d_out); SMI_CHIM_CHECK("k_chimera"); }
            }
            SMI_HIP(hipGetLastError());
            // reads with more than kCap accepted positions / matches: once more without the cap (K-CHIM-S).  The overflow queue sits
            // behind the main queue in the same buffer (the main queue never uses more than n entries, the buffer holds 2 n + 64)
            uint32_t *d_over_count = d_count + 1, *d_over = d_list + n;
            { hipLaunchKernelGGL(k_collect_overflow, dim3(64), dim3(256), 0, s, d_out, d_list, d_count, d_over, d_over_count); SMI_CHIM_CHECK("k_collect_overflow"); }
            uint32_t n_over = 0;
            {
                uint32_t *pw = static_cast<uint32_t *>(pin_words(ctx));
                SMI_HIP(hipMemcpyAsync(pw ? pw : &n_over, d_over_count, 4, hipMemcpyDeviceToHost, s));
                SMI_HIP(hipStreamSynchronize(s));
                if (pw) n_over = *pw;
            }
#ifdef SMI_MEASURE
            if (getenv("SMI_CHIM_STATS")) {
                uint32_t h[16];
                SMI_HIP(hipMemcpy(h, d_count, 64, hipMemcpyDeviceToHost));
                fprintf(stderr, "[chim stats] queue %u over %u positions %u stretches %u ad-entries %u | caps hit: st/read %u st-queue %u ad-queue %u matches %u slot %u gated %u pos-queue %u accepted %u\n",
                        h[0], h[1], h[2], h[3], h[4], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
            }
#endif
            if (n_over) {
                std::vector<uint32_t> over(n_over);
                std::vector<uint64_t> offs(n + 1);
                SMI_HIP(hipMemcpyAsync(over.data(), d_over, (size_t)n_over * 4, hipMemcpyDeviceToHost, s));
                SMI_HIP(hipMemcpyAsync(offs.data(), d_offsets, (n + 1) * 8, hipMemcpyDeviceToHost, s));
                SMI_HIP(hipStreamSynchronize(s));
                std::sort(over.begin(), over.end());  // the collection order is not deterministic; the results do not depend on it
                std::vector<uint64_t> scr_off(n_over + 1, 0);
                for (uint32_t i = 0; i < n_over; i++) scr_off[i + 1] = scr_off[i] + 7 * (offs[over[i] + 1] - offs[over[i]] + 8);
                uint64_t *d_scr_off = nullptr;
                int32_t *d_scr = nullptr;
                uint32_t *d_over_sorted = nullptr;
                SMI_HIP(hipMalloc(&d_scr_off, (n_over + 1) * 8));
                hipError_t e = hipMalloc(&d_scr, scr_off[n_over] * 4);
                if (e == hipSuccess) e = hipMalloc(&d_over_sorted, (size_t)n_over * 4);
                if (e != hipSuccess) {
                    (void)hipFree(d_scr_off);
                    (void)hipFree(d_scr);
                    return hip_fail(e, "hipMalloc (K-CHIM-S scratch)");
                }
                SMI_HIP(hipMemcpyAsync(d_scr_off, scr_off.data(), (n_over + 1) * 8, hipMemcpyHostToDevice, s));
                SMI_HIP(hipMemcpyAsync(d_over_sorted, over.data(), (size_t)n_over * 4, hipMemcpyHostToDevice, s));
                const unsigned gs = (unsigned)std::min<uint32_t>(n_over, 4096);
                if (tl == 27)
                    { hipLaunchKernelGGL((k_chimera_serial<27, 22>), dim3(gs), dim3(64), 0, s, d_planes, st, d_pstart, d_offsets, d_over_sorted, n_over, d_scr_off, d_scr, P, d_out); SMI_CHIM_CHECK("k_chimera_serial"); }
                else
                    { hipLaunchKernelGGL((k_chimera_serial<22, 25>), dim3(gs), dim3(64), 0, s, d_planes, st, d_pstart, d_offsets, d_over_sorted, n_over, d_scr_off, d_scr, P, d_out); SMI_CHIM_CHECK("k_chimera_serial"); }
                hipError_t e2 = hipStreamSynchronize(s);
                (void)hipFree(d_scr_off);
                (void)hipFree(d_scr);
                (void)hipFree(d_over_sorted);
                if (e2 != hipSuccess) return hip_fail(e2, "k_chimera_serial");
            }
        }
    }
    if (int rc = time_end(ctx, SMI_K_CHIMERA, s)) return rc;
    return SMI_OK;
}

// text positions of the bases and qualities of output record f (a fragment of input record frag_src[f] >> 2, or the record itself)
__global__ __launch_bounds__(256) void k_frag_text_starts(const uint64_t *__restrict__ seq_start, const uint64_t *__restrict__ qual_start,
                                                          const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ frag_offsets,
                                                          const uint32_t *__restrict__ frag_src, size_t m, uint64_t *__restrict__ bstart,
                                                          uint64_t *__restrict__ qstart) {
    const size_t f = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (f >= m) return;
    const size_t r = frag_src ? (size_t)(frag_src[f] >> 2) : f;
    const uint64_t in_read = frag_src ? frag_offsets[f] - offsets[r] : 0;
    bstart[f] = seq_start[r] + in_read;
    if (qstart) qstart[f] = qual_start[r] + in_read;
}

int launch_frag_text_starts(smi_ctx *, const uint64_t *d_seq_start, const uint64_t *d_qual_start, const uint64_t *d_offsets,
                            const uint64_t *d_frag_offsets, const uint32_t *d_frag_src, size_t m, uint64_t *d_bstart, uint64_t *d_qstart,
                            hipStream_t s) {
    if (!m) return SMI_OK;
    { hipLaunchKernelGGL(k_frag_text_starts, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d_seq_start, d_qual_start, d_offsets,
                       d_frag_offsets, d_frag_src, m, d_bstart, d_qstart); SMI_CHIM_CHECK("k_frag_text_starts"); }
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

int launch_split_offsets(smi_ctx *, const smi_chimera_result *d_chim, const uint64_t *d_offsets, size_t n,
                         uint32_t *d_scratch, uint64_t *d_total, uint64_t *d_frag_offsets, uint32_t *d_frag_src,
                         hipStream_t s) {
    if (!n) return SMI_OK;
    const size_t n_blocks = (n + 1023) / 1024;
    { hipLaunchKernelGGL(k_frag_counts, dim3((unsigned)n_blocks), dim3(256), 0, s, d_chim, n, d_scratch); SMI_CHIM_CHECK("k_frag_counts"); }
    { hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, d_scratch, n_blocks, d_total); SMI_CHIM_CHECK("k_scan_block_sums"); }
    { hipLaunchKernelGGL(k_frag_offsets, dim3((unsigned)n_blocks), dim3(256), 0, s, d_chim, d_offsets, n, d_scratch,
                       d_frag_offsets, d_frag_src); SMI_CHIM_CHECK("k_frag_offsets"); }
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

}  // namespace smi
