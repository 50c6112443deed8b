// smi_scan.hip -- K-SCAN: the reference's 3' read scan on gfx950 (hand-written HIP).
//
// Reference units (bytecode, see DESIGN.md for the citation form):
//   PolyATSearcher.findpolyAT                    FJ!nanopore/analyzers/PolyATSearcher.java:L56-252
//   AdapterTSOanalyzer.scanForAdapterOrTSOseq    FJ!nanopore/analyzers/AdapterTSOanalyzer.java:L84-110
//   $Kmers.nKmersMatching_4mer                   TB!nuc/encoding/onebyte/NucleicAcidInmutableOneBytePerBase.java:L533-543
//   NeedlemanWunsch / SequenceAlignment          TB!nuc/alignment/needleman/*.java
//   Match.countErrorsInNeedleman, NeedlemanMatch FJ!nanopore/analyzers/{Match,NeedlemanMatch}.java
//   PolyATadapterAnalyzerBase.analyze etc.       FJ!nanopore/analyzers/PolyATadapterAnalyzerBase.java:L145-319
//   pass-1 quality filter                        FJ!nanoporereadscanner/analyzers/UsedCellBCListGenerator.java:L198-202
//
// MI355X mapping.  Input = both 208-base ends of every read in scan orientation, stored as four IUPAC bit-planes
// (A,G,C,T bits of the reference's 4-bit code) in a [plane-word][end] layout, so a wave's loads are 256-B
// coalesced rows.  One lane owns one read END (lanes 2i / 2i+1 = head / reverse-complemented tail of read i);
// the two lanes meet through one `__shfl_xor` for the strand decision.  Everything positional is bit-parallel on
// the planes staged in LDS ([word][lane] layout, conflict-free):
//   * polyT: popcount of 15-bit windows of the exact-T plane
//   * 4-mer gate: 64 scan positions per step (AND of four shifted match planes, two-counter), candidates = set bits
//   * Needleman-Wunsch only on candidates: the scan-phase DP carries the statistics the reference reads off the
//     traceback (#x, leading template gaps) forward in a packed int, so no traceback is needed there; the
//     accepted alignment is re-run once with 2-bit directions in LDS and walked back from the end, which is the
//     order all of NeedlemanMatch's statistics are defined in.
// Java float semantics are kept with explicit non-fused fp32 ops.  Integer/bitwise work: no MFMA.
#include "smi_internal.h"
#include "smi_nw.h"

namespace smi {

constexpr int kEndBases = SMI_END_BASES;     // 208
constexpr int kPlaneWords = SMI_PLANE_WORDS;  // 7
constexpr int kLdsWords = 7;                  // per plane in LDS; fetches past the plane read as 0
constexpr int kBlock = 256;
#ifndef SMI_SCAN_WAVES
#define SMI_SCAN_WAVES 4  // waves per SIMD the register allocation of K-SCAN is held to (LDS: 40 KiB per block)
#endif
// ... for the kernels of the shipped 10-mer (pass 2: the 3' one keeps one loop-invariant value in scratch, stored in the prologue).  The others get
// the registers they need to hold everything: at 128 VGPRs the 22-mer kernels spill 13 - 47 values, and a value spilled INSIDE a divergent region is
// stored for the lanes active there only -- a lane that sat the region out reads back whatever its scratch slot held.  That is how two never-taken
// branches made the 22-mer kernel drop the records of reads with no side chosen (the even lane's `side` and `read` came back from scratch: NOTES R4.5).
// The generic kernels (any adapter / TSO / polyA window: what a config.xml with other sequences runs) fill the same exact band as the specialised ones --
// the band follows from the 4-mer gate, not from the sequence (smi_nw.h "Band").  -DSMI_SCAN_GENERIC_FULL=1 (a variant build) gives them the whole matrix
// again, as an independent check of the band.
#ifndef SMI_SCAN_GENERIC_FULL
#define SMI_SCAN_GENERIC_FULL 0
#endif
#ifndef SMI_SCAN_GENERIC_WAVES
#define SMI_SCAN_GENERIC_WAVES 3  // (four: 128 registers with 17 spilled values -- not in this kernel, see above; 4.78 instead of 5.47 ms per 10 M reads)
#endif
template <int AD, bool SHIP>
constexpr int scan_waves() {
    return SHIP ? (AD == 10 ? SMI_SCAN_WAVES : 3) : (AD == 10 ? SMI_SCAN_GENERIC_WAVES : 2);
}

// SMI_SCAN_ABLATE's switches are compiled into measurement builds only.  (They used to be run-time tests of a field that is 0 in the shipped
// library; two more of them -- uniform, never taken -- made the 22-mer kernel lose the records of reads with no side chosen: a kernel at 128
// VGPRs with spilled SGPRs is not the place for code that does nothing.)
#ifdef SMI_MEASURE
#define SMI_ABLATED(bit) ((P.ablate & (bit)) != 0)
#else
#define SMI_ABLATED(bit) false
#endif

struct ScanParams {
    int min_read_length;
    int polya_len;       // 15
    int window;          // 150
    int thr_first;       // min T count with (float)k/len >= frac           (12 for 15 / 0.75)
    int thr_adv;         // min T count with (double)((float)k/len) >= (double)frac - 0.1   (10)
    int max_mm;          // maxNeedlemanMismatches (3)
    int min_3p;          // minAdapter3pMatches (8)
    int min_bc_qv;       // 8
    int min_read_qv;     // 8
    uint32_t adapter_nib[3];  // 4-bit codes of the adapter, eight per word, padded with 0 (three SGPRs instead of 22)
    __host__ __device__ __forceinline__ uint32_t a4(int i) const { return (adapter_nib[i >> 3] >> ((i & 7) * 4)) & 15u; }
    int dont_polya;      // --noPolyARequired (dontSearchPolyAFor5pBarcoding)
    int window5;         // AdapterSearchWindow (110)
    uint32_t tso_nib[2];  // 4-bit codes of the read scan's TSO (generic kernels; the shipped kernels carry tso4() as constants)
    int tso_window, tso_max_mm, tso_min_consec, tso_min_two;  // 90, 5, 8, 12
    __host__ __device__ __forceinline__ uint32_t t4(int i) const { return (tso_nib[i >> 3] >> ((i & 7) * 4)) & 15u; }
    int finder_bits;     // host side only: the bit-parallel polyT finder applies (polya_len 15, thresholds 12 / 10, window <= 160) -- otherwise the generic kernels run
    int ablate;          // measurement only (SMI_SCAN_ABLATE): 1 no TSO alignments, 2 no adapter alignments, 4 no polyT finder, 8 no TSO gates, 16 no TSO pre-filter (results unchanged), 32 finder: first loop only, 64 no adapter gates, 128 no folds
};

// ---- LDS plane access -----------------------------------------------------------------------------------------
// planes: [4][kLdsWords][kBlock] u32 (A, G, C, T bits of the 4-bit code); the lane's column is `tid`
__device__ __forceinline__ uint64_t get64(const uint32_t *lds_plane, int tid, int bitpos) {
    // 64 bits starting at bitpos (0 <= bitpos < 224); words past the plane read as 0
    const int w = bitpos >> 5, s = bitpos & 31;
    const uint32_t w0 = lds_plane[w * kBlock + tid];
    const uint32_t w1 = w + 1 < kLdsWords ? lds_plane[(w + 1) * kBlock + tid] : 0u;
    const uint32_t w2 = w + 2 < kLdsWords ? lds_plane[(w + 2) * kBlock + tid] : 0u;
    const uint64_t lo = ((uint64_t)w1 << 32) | w0;
    uint64_t r = lo >> s;
    if (s) r |= (uint64_t)w2 << (64 - s);
    return r;
}
__device__ __forceinline__ uint32_t get32(const uint32_t *lds_plane, int tid, int bitpos) {
    const int w = bitpos >> 5, s = bitpos & 31;
    const uint32_t w0 = lds_plane[w * kBlock + tid];
    const uint32_t w1 = w + 1 < kLdsWords ? lds_plane[(w + 1) * kBlock + tid] : 0u;
    return (uint32_t)((((uint64_t)w1 << 32) | w0) >> s);
}

// 64 match bits (read base at bitpos+i matches IUPAC code a4) = OR of the planes selected by a4
__device__ __forceinline__ uint64_t match64(const uint32_t *planes, int tid, uint32_t a4, int bitpos) {
    uint64_t m = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if ((a4 >> c) & 1u) m |= get64(planes + c * kLdsWords * kBlock, tid, bitpos);
    return m;
}
__device__ __forceinline__ uint32_t match32(const uint32_t *planes, int tid, uint32_t a4, int bitpos) {
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if ((a4 >> c) & 1u) m |= get32(planes + c * kLdsWords * kBlock, tid, bitpos);
    return m;
}

// exact T (code 8, not N = 15): T plane without the A plane (the packer only emits A, G, C, T, N and '-')
__device__ __forceinline__ uint32_t get32_t(const uint32_t *planes, int tid, int bitpos) {
    return get32(planes + 3 * kLdsWords * kBlock, tid, bitpos) & ~get32(planes, tid, bitpos);
}

// ---- polyT finder (PolyATSearcher.java:L56-252) ---------------------------------------------------------------
// Entry `pos` of the reference's score list is the T fraction of bases [pos+1, pos+15].
__device__ __forceinline__ bool find_polyt(const uint32_t *planes, int tid, const ScanParams &P, int &begin1, int &end1) {
    const int ML = P.polya_len;
    const uint32_t wmask = (1u << ML) - 1u;
    const int n = P.window + ML + 10;  // sub-sequence length (175)
    int first = -1;
    for (int pos = 0; pos < P.window; pos++) {
        const uint32_t x = get32_t(planes, tid, pos);
        // every window of the entries pos .. pos+16 lies inside bits 1..31 of x: too few T's there -> none can pass
        if (ML <= 15 && __popc(x >> 1) < P.thr_first) {
            pos += 16;
            continue;
        }
        const int cnt = __popc((x >> 1) & wmask);                       // L199-200
        if (cnt >= P.thr_first && (x & 1u) && __popc(x & 31u) > 2) {    // L217-218, lambda$2 L98-101
            first = pos;
            break;
        }
        // the entry of a position can only pass when its own base is a T: go straight to the next T
        const uint32_t rest = x >> 1;
        pos += rest ? __builtin_ctz(rest) : 30;
    }
    if (first < 0) return false;
#ifdef SMI_MEASURE
    if (SMI_ABLATED(32)) {  // timing only: the first loop alone
        begin1 = first + 1;
        end1 = first + ML;
        return true;
    }
#endif
    // From here on the walk stays around the run, so the exact-T bits are read through a 64-bit register window (two
    // 64-bit plane fetches per refill) instead of two plane fetches per look.
    int wbase = 0;
    uint64_t wbits = 0;
    auto refill = [&](int p) {
        wbase = p;
        wbits = get64(planes + 3 * kLdsWords * kBlock, tid, p) & ~get64(planes, tid, p);
    };
    auto look = [&](int p, bool backwards) -> uint32_t {  // 32 exact-T bits from position p on
        if (p < wbase || p + 32 > wbase + 64) refill(backwards ? max(p - 31, 0) : p);
        return (uint32_t)(wbits >> (p - wbase));
    };
    refill(first);
    int start = first;
    const int INC[8] = {20, 15, 10, 5, 4, 3, 2, 1};  // L223-230
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int inc = INC[k];
        while (start + inc < P.window && __popc((look(start + inc, false) >> 1) & wmask) >= P.thr_adv) start += inc;
    }
    int endpos = start + ML - 1;  // L231
    // lambda$findpolyAT$3 L122-142: walk back until T at endpos, >=2 T in the last 2, >=3 in 4, >=4 in 5
    while (endpos > 4) {
        const uint32_t x = look(endpos - 4, true);  // bit i = base endpos-4+i
        const bool ok = ((x >> 4) & 1u) && __popc((x >> 3) & 3u) >= 2 && __popc((x >> 1) & 15u) >= 3 && __popc(x & 31u) >= 4;
        if (ok) break;
        endpos--;
    }
    while (n > endpos + 6 && __popc(look(endpos + 1, false) & 31u) > 3) endpos += 5;  // L145-147
    while (n > endpos + 4 && __popc(look(endpos + 1, false) & 7u) > 1) endpos += 3;   // L156-158
    while (endpos < n - 1 && (look(endpos + 1, false) & 1u)) endpos++;                // L171-172
    begin1 = first + 1;
    end1 = endpos + 1;
    return true;
}

// ---- the same finder, bit-parallel over positions (round 4) ----------------------------------------------------------------------------------
// The loop above costs a wave 1,620 VALU instructions of K-SCAN's 6,480 (profiles/r04/k_scan_budget.json): ~ 16 turns of a loop whose body
// fetches two planes at a run-time offset, for the 32 ends of a wave that have no polyT at all, then the increments of the extension one
// probe at a time.  Everything it asks is a window count over the exact-T bits, and those can be had for ALL positions at once: the lane
// keeps its 224 exact-T bits in seven registers, adds five shifted copies with bit-sliced adders (the number of T's in [q, q + 5) for every
// q: three planes), three shifted copies of that sum (T's in [q, q + 15): four planes), and the loop's conditions become masks:
//     entry(pos)  = T[pos] & (>= 3 T's in [pos, pos + 5)) & (>= thr_first T's in [pos + 1, pos + 16))        first = lowest set bit below `window`
//     go(p)       = p < window & (>= thr_adv T's in [p + 1, p + 16))                                          the extension's probe
//     back(e)     = T[e] & T[e - 1] & (two of T[e - 2], T[e - 3], T[e - 4])                                   the walk back: highest set bit <= end
// (back: with T[e] and T[e - 1] set, ">= 3 T's in the last four" is "one of T[e-2], T[e-3]", ">= 4 in the last five" is "two of T[e-2 .. e-4]",
// which implies it.)  For polya_len = 15 with the thresholds 12 and 10 the shipped fractions give; anything else takes the loop above.
// The generic kernels keep the loop, so the parity suite (shipped against generic against the oracle) compares the two finders as well.
__device__ __forceinline__ void full_add(uint32_t a, uint32_t b, uint32_t c, uint32_t &sum, uint32_t &carry) {
    sum = a ^ b ^ c;
    carry = (a & b) | (c & (a ^ b));
}
// 64 bits from bit position p of a 192-bit mask kept in the lane's own three 64-bit slots of the candidate-mask area ([slot][lane]: the four
// waves of a block run independently, so a lane may only touch the columns of its own wave); bits past the mask read as 0
__device__ __forceinline__ uint64_t mask64(const uint64_t *m, int tid, int p) {
    const int q = p >> 6, sh = p & 63;
    const uint64_t a = q < 3 ? m[q * kBlock + tid] : 0ull, b = q + 1 < 3 ? m[(q + 1) * kBlock + tid] : 0ull;
    uint64_t r = a >> sh;
    if (sh) r |= b << (64 - sh);
    return r;
}
__device__ __forceinline__ bool find_polyt_bits(const uint32_t *planes, uint64_t *go_lds, int tid, const ScanParams &P, const uint32_t (&tw)[8], int &begin1,
                                                int &end1) {
    constexpr int ML = 15;
    const int n = P.window + ML + 10;
    // T's in [q, q + 5) for every q: planes c5[0..2], seven words (the eighth reads as 0)
    uint32_t c5[3][8];
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const uint32_t s1 = __builtin_amdgcn_alignbit(tw[k + 1], tw[k], 1), s2 = __builtin_amdgcn_alignbit(tw[k + 1], tw[k], 2),
                       s3 = __builtin_amdgcn_alignbit(tw[k + 1], tw[k], 3), s4 = __builtin_amdgcn_alignbit(tw[k + 1], tw[k], 4);
        uint32_t x, k1, k2;
        full_add(tw[k], s1, s2, x, k1);
        full_add(x, s3, s4, c5[0][k], k2);
        c5[1][k] = k1 ^ k2;
        c5[2][k] = k1 & k2;
    }
    c5[0][7] = c5[1][7] = c5[2][7] = 0u;
    // T's in [q, q + 15) = the three five-counts at q, q + 5, q + 10: >= 12 and >= 10 as masks, six words (+ a zero word)
    uint32_t ge12[7], ge10[7];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint32_t b[3], c[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            b[j] = __builtin_amdgcn_alignbit(c5[j][k + 1], c5[j][k], 5);
            c[j] = __builtin_amdgcn_alignbit(c5[j][k + 1], c5[j][k], 10);
        }
        uint32_t r0, k0, t1, k1a, t2, k2a, r2, k2b;
        full_add(c5[0][k], b[0], c[0], r0, k0);
        full_add(c5[1][k], b[1], c[1], t1, k1a);
        const uint32_t r1 = t1 ^ k0, k1b = t1 & k0;
        full_add(c5[2][k], b[2], c[2], t2, k2a);
        full_add(t2, k1a, k1b, r2, k2b);
        const uint32_t r3 = k2a | k2b;  // (the count is at most 15: the two carries never come together)
        (void)r0;
        ge12[k] = r3 & r2;
        ge10[k] = r3 & (r2 | r1);
    }
    ge12[6] = ge10[6] = 0u;
    // entry positions; the extension's probe mask goes to LDS (it is read at run-time offsets)
    int first = -1;
    uint32_t gom[6];
#pragma unroll
    for (int k = 5; k >= 0; k--) {
        const int keep = P.window - 32 * k;  // positions below `window` (uniform)
        const uint32_t below = keep >= 32 ? 0xFFFFFFFFu : (keep <= 0 ? 0u : ((1u << keep) - 1u));
        const uint32_t entry = __builtin_amdgcn_alignbit(ge12[k + 1], ge12[k], 1) & tw[k] & (c5[2][k] | (c5[1][k] & c5[0][k])) & below;
        gom[k] = __builtin_amdgcn_alignbit(ge10[k + 1], ge10[k], 1) & below;
        if (entry) first = 32 * k + __builtin_ctz(entry);
    }
#pragma unroll
    for (int q = 0; q < 3; q++) go_lds[q * kBlock + tid] = ((uint64_t)gom[2 * q + 1] << 32) | gom[2 * q];
    if (first < 0) return false;
    int start = first;
    {
        int wbase = first;
        uint64_t gw = mask64(go_lds, tid, first);
        auto go = [&](int p) -> bool {  // start <= p <= start + 20, and start only grows: the window is taken from `start`
            if (p >= wbase + 64) {
                wbase = start;
                gw = mask64(go_lds, tid, start);
            }
            return (gw >> (p - wbase)) & 1ull;
        };
        const int INC[8] = {20, 15, 10, 5, 4, 3, 2, 1};  // L223-230
#pragma unroll
        for (int k = 0; k < 8; k++)
            while (go(start + INC[k])) start += INC[k];
    }
    int endpos = start + ML - 1;  // L231
    {
        // lambda$findpolyAT$3 L122-142: the last position <= endpos (and > 4) that passes; 4 when there is none
        int e = 4;
        bool found = false;
#pragma unroll
        for (int k = 5; k >= 0; k--) {
            const uint32_t lo = k ? tw[k - 1] : 0u;
            const uint32_t l1 = __builtin_amdgcn_alignbit(tw[k], lo, 31), l2 = __builtin_amdgcn_alignbit(tw[k], lo, 30), l3 = __builtin_amdgcn_alignbit(tw[k], lo, 29),
                           l4 = __builtin_amdgcn_alignbit(tw[k], lo, 28);
            uint32_t back = tw[k] & l1 & ((l2 & l3) | (l4 & (l2 ^ l3)));
            if (k == 0) back &= ~31u;
            const int rel = endpos - 32 * k;
            const uint32_t upto = rel >= 31 ? 0xFFFFFFFFu : (rel < 0 ? 0u : ((2u << rel) - 1u));
            back &= upto;
            if (!found && back) {
                e = 32 * k + 31 - __builtin_clz(back);
                found = true;
            }
        }
        endpos = e;
    }
    // the three forward steps look at a handful of positions: the exact-T bits through a 64-bit register window, as in the loop above
    int wbase = 0;
    uint64_t wbits = 0;
    auto look = [&](int p) -> uint32_t {  // 32 exact-T bits from position p on
        if (p < wbase || p + 32 > wbase + 64) {
            wbase = p;
            wbits = get64(planes + 3 * kLdsWords * kBlock, tid, p) & ~get64(planes, tid, p);
        }
        return (uint32_t)(wbits >> (p - wbase));
    };
    wbase = endpos + 1;
    wbits = get64(planes + 3 * kLdsWords * kBlock, tid, wbase) & ~get64(planes, tid, wbase);
    while (n > endpos + 6 && __popc(look(endpos + 1) & 31u) > 3) endpos += 5;  // L145-147
    while (n > endpos + 4 && __popc(look(endpos + 1) & 7u) > 1) endpos += 3;   // L156-158
    while (endpos < n - 1 && (look(endpos + 1) & 1u)) endpos++;                // L171-172
    begin1 = first + 1;
    end1 = endpos + 1;
    return true;
}

// TSO "AACGCAGAGTACATGG" (Jar/config.xml:155) as 4-bit codes A=1 G=2 C=4 T=8, base i in bits [4i+3:4i]
__host__ __device__ __forceinline__ uint32_t tso4(int i) {
    constexpr uint64_t TSO = 0x2281418212142411ull;
    return (uint32_t)(TSO >> (4 * i)) & 15u;
}

// The adapters the reference ships (Jar/config.xml:111-113: `sequence` "CTTCCGATCT" and `sequence_complete`
// "CTACACGACGCTCTTCCGATCT" of adapter_for3pBarcoding / the 5' adapter) as compile-time 4-bit codes, base i in bits [4i+3:4i] of
// word i / 8: a kernel compiled for one of them (SHIP = true) selects the column planes of an alignment and the gate planes by
// constants instead of by uniform bit tests of ScanParams (80 SGPR select masks at AD = 10, spilled to VGPR lanes and read back per
// use).  Any other adapter runs the generic kernels (SHIP = false).
template <int AD>
__host__ __device__ constexpr uint32_t shipped_a4(int i) {
    constexpr const char *S = AD == 10 ? "CTTCCGATCT" : "CTACACGACGCTCTTCCGATCT";
    return S[i] == 'A' ? 1u : S[i] == 'G' ? 2u : S[i] == 'C' ? 4u : 8u;
}

// 4-mer gate (Kmers.nKmersMatching_4mer > 1) for 64 scan positions starting at bit b of the owner's planes
template <int N, typename F>
__device__ __forceinline__ uint64_t gate64(const uint32_t *planes, int owner, int b, F code) {
    uint64_t any = 0, two = 0;
    uint64_t m0 = match64(planes, owner, code(0), b), m1 = match64(planes, owner, code(1), b + 1),
             m2 = match64(planes, owner, code(2), b + 2);
#pragma unroll
    for (int i = 0; i + 3 < N; i++) {
        const uint64_t m3 = match64(planes, owner, code(i + 3), b + i + 3);
        const uint64_t k = m0 & m1 & m2 & m3;
        two |= any & k;
        any |= k;
        m0 = m1;
        m1 = m2;
        m2 = m3;
    }
    return two;
}

// the same gate for 32 scan positions (the second chunk of the TSO scan: positions 65 .. 90 of its window)
template <int N, typename F>
__device__ __forceinline__ uint32_t gate32(const uint32_t *planes, int owner, int b, F code) {
    uint32_t any = 0, two = 0;
    uint32_t m0 = match32(planes, owner, code(0), b), m1 = match32(planes, owner, code(1), b + 1), m2 = match32(planes, owner, code(2), b + 2);
#pragma unroll
    for (int i = 0; i + 3 < N; i++) {
        const uint32_t m3 = match32(planes, owner, code(i + 3), b + i + 3);
        const uint32_t k = m0 & m1 & m2 & m3;
        two |= any & k;
        any |= k;
        m0 = m1;
        m1 = m2;
        m2 = m3;
    }
    return two;
}

__device__ __forceinline__ uint64_t keep_low(uint64_t m, int n_bits) {
    return n_bits <= 0 ? 0ull : (n_bits >= 64 ? m : (m & ((1ull << n_bits) - 1ull)));
}

// k-th set bit (k < popcount) of a 64-bit mask
__device__ __forceinline__ int kth_bit(uint64_t m, int k) {
    for (; k > 0; k--) m &= m - 1;
    return __builtin_ctzll(m);
}

// result of one aligned candidate, as the owner lane needs it (5 words per entry in LDS, odd stride)
struct Entry {
    float ne, end5, endn;
    uint32_t a;  // nmis | ins << 8 | del(+128) << 16 | term6 << 24
    uint32_t b;  // consec | best_two << 8 | pos << 16
};

// wave-wide exclusive prefix sum of one int per lane; returns the wave total through `total`
__device__ __forceinline__ int wave_exscan(int v, int lane, int &total) {
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(inc, o);
        if (lane >= o) inc += y;
    }
    total = __shfl(inc, 63);
    return inc - v;
}

// ---------------------------------------------------------------------------------------------------------------
// One wavefront = 64 read ends = 32 reads (a block is four independent wavefronts: no block-level barrier).  Phases:
//   A (lane = end)        stage planes, polyT finder, 4-mer gates of the adapter (positions up to polyT end) and of
//                         the TSO (positions 1..90): candidate bit masks
//   B (lane = candidate)  the wave's candidates are numbered through a prefix sum and aligned 64 at a time, so
//                         Needleman-Wunsch always runs with full waves whatever the per-read candidate counts are
//   C (lane = end)        owners fold their candidates' statistics in scan order (best score / first best, the TSO
//                         skip rule), lanes 2i/2i+1 exchange for the strand decision and the TSO rules, one lane
//                         writes the record and the barcode window
// ---------------------------------------------------------------------------------------------------------------
template <int AD, bool FP, bool SHIP>
__global__ __launch_bounds__(kBlock, (scan_waves<AD, SHIP>())) void k_scan(const uint32_t *__restrict__ ends, const int32_t *__restrict__ read_len,
                                                 const uint8_t *__restrict__ qtail, const uint32_t *__restrict__ qsum,
                                                 size_t n_reads, ScanParams P, smi_scan_result *__restrict__ out,
                                                 smi_bc_window *__restrict__ windows) {
    // Every aligned candidate passed gate64 on the slice that is aligned (>= 2 matching 4-mers = >= 5 matching bases on the main diagonal),
    // which bounds how far an optimal path can leave the diagonal (smi_nw.h "Band"): 3 cells for the 10-mer, 12 for the 22-mer, 7 for the
    // TSO.  The kernels of the shipped adapters fill the band only; the generic kernels fill the whole matrix, and the parity tests run both.
    constexpr int kBandAd = (SHIP || !SMI_SCAN_GENERIC_FULL) ? nw_band<AD, 5>() : AD, kBandTso = (SHIP || !SMI_SCAN_GENERIC_FULL) ? nw_band<16, 5>() : 16;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *planes = lds;                                                           // [4][kLdsWords][kBlock]
    uint64_t *cmask = reinterpret_cast<uint64_t *>(planes + 4 * kLdsWords * kBlock);  // [3][kBlock] candidate bits
    uint32_t *coff = reinterpret_cast<uint32_t *>(cmask + 3 * kBlock);                // [kBlock] candidate offsets
    uint32_t *ent = coff + kBlock;                                                    // [kBlock][5]
    const int tid = threadIdx.x;
    const size_t n_ends = 2 * n_reads;
    for (size_t e0 = (size_t)blockIdx.x * kBlock; e0 < n_ends; e0 += (size_t)gridDim.x * kBlock) {
        const size_t e = e0 + tid;
        const bool active = e < n_ends;
        const size_t read = e >> 1;
        const int side = (int)(e & 1);  // 0 = head (forward scan), 1 = reverse-complemented tail
        // ---- phase A ---------------------------------------------------------------------------------------
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int w = 0; w < kLdsWords; w++)
                planes[(c * kLdsWords + w) * kBlock + tid] = active ? ends[(size_t)(c * kPlaneWords + w) * n_ends + e] : 0u;
        const int len = active ? read_len[read] : 0;
        const bool long_enough = len >= P.min_read_length;  // testReadLength L131-137
        int pb = 0, pe = 0;
        bool has_t = false;
        if (active && long_enough && !(FP && P.dont_polya) && !SMI_ABLATED(4)) {
            if (SHIP) {  // (the kernels of the shipped adapters carry the bit-parallel finder only -- both finders inlined pushed the 5' kernel past 64 KB of code and cost it 13 %; launch_scan sends parameters it is not built for to the generic kernels)
                // exact T (the T plane without the A plane) of the lane's end, read back from its LDS columns: only the path that runs the finder pays
                // for them
                // (the empty asm keeps the loads, and with them the branch-free front of the finder, INSIDE this branch: hoisted above it -- the compiler
                // did -- they run for every wave of the 5' --noPolyARequired kernel, which never takes the branch: 1.44 -> 1.63 ms per 10 M reads)
                asm volatile("" ::: "memory");
                uint32_t tw[8];
#pragma unroll
                for (int w = 0; w < kLdsWords; w++) tw[w] = planes[(3 * kLdsWords + w) * kBlock + tid] & ~planes[w * kBlock + tid];
                tw[7] = 0u;
                has_t = find_polyt_bits(planes, cmask, tid, P, tw, pb, pe);
            } else
                has_t = find_polyt(planes, tid, P, pb, pe);
        }
        uint64_t am[3] = {0, 0, 0};
        // 5' barcoding scans an end when the polyT was found at the OTHER end (or no polyA is asked for):
        // PolyATadapterAnalyzer_5pBCUMI.java:L51-68
        const bool scan5 = FP && active && long_enough && (P.dont_polya || __shfl_xor((int)has_t, 1));
        if (FP ? scan5 : has_t) {
            // 3': positions 1 .. min(pe - AD, pe - 12)  (seqTilPolyAend has length pe; L49-61, AdapterTSOanalyzer L87)
            // 5': positions 1 .. AdapterSearchWindow of the first window + AD + maxMM + 5 bases
            const int last = FP ? P.window5 : min(pe - AD, pe - 12);
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                if (last > ch * 64 && !SMI_ABLATED(64))  // (the third chunk is needed by the few ends whose polyT ends beyond position 138: most waves skip it)
                    am[ch] = keep_low(gate64<AD>(planes, tid, ch * 64, [&](int i) { return SHIP ? shipped_a4<AD>(i) : P.a4(i); }), last - ch * 64);
        }
        const int n_ad = __popcll(am[0]) + __popcll(am[1]) + __popcll(am[2]);
        const int lane = tid & 63, wbase = tid & ~63;  // this wave's lanes are wbase .. wbase+63
        int tot_ad;
        // count (<= 192) and offset (<= 64 * 192) of a lane's candidates share a register while the alignments run
        const uint32_t no_ad = (uint32_t)n_ad | ((uint32_t)wave_exscan(n_ad, lane, tot_ad) << 8);
        const uint32_t pbe = (uint32_t)pb | ((uint32_t)pe << 16);

        // ---- phases B + C(fold): adapter candidates, then TSO candidates -----------------------------------------
        // adapter fold state (AdapterScanRslt best key + getMatchList L275-319)
        // The fold state rides through the alignment loops in few registers: a_pack / t_pack = pos | nmis << 8 | ins << 16 | (del + 128) << 24
        // (positions are <= 192), t_aux = consec | best_two << 8 | skip << 16
        float a_best = 3.4028234663852886e+38f, a_key = 0.0f;
        bool a_have = false;
        uint32_t a_pack = 0;
        float a_endn = 0.0f;
        // TSO fold state (scanForAdapterOrTSOseq with maxErrors = 5, L96-104; scanForTSO takes the first best position)
        float t_best = 3.4028234663852886e+38f;
        uint32_t t_pack = 0, t_aux = 0;
#pragma unroll 1
        for (int kind = 0; kind < 2; kind++) {
            // this kind's candidate masks and offsets go to LDS (the previous kind's are dead: wave_sync below).  The TSO gates are
            // computed here, after the adapter alignments, so that their masks are not live across them.
            uint32_t no = no_ad;
            int total = tot_ad;
            if (kind == 0) {
                cmask[0 * kBlock + tid] = am[0];
                cmask[1 * kBlock + tid] = am[1];
                cmask[2 * kBlock + tid] = am[2];
            } else {
                uint64_t tm[2] = {0, 0};
                if (active && long_enough && !FP && !SMI_ABLATED(8)) {
                    // TSO: positions 1 .. min(116 - 16, windowForTSOsearch = 90)  (scanForTSO L325); the 5' analyzer has no TSO scan
                    if (SHIP) {  // window 90: 64 positions + 26 of the next 32 (a 64-position gate for those costs twice the operations)
                        tm[0] = gate64<16>(planes, tid, 0, [](int i) { return tso4(i); });
                        tm[1] = (uint64_t)(gate32<16>(planes, tid, 64, [](int i) { return tso4(i); }) & ((1u << (90 - 64)) - 1u));
                    } else {
#pragma unroll
                        for (int ch = 0; ch < 2; ch++)
                            tm[ch] = keep_low(gate64<16>(planes, tid, ch * 64, [&](int i) { return P.t4(i); }), P.tso_window - ch * 64);
                    }
                }
                if (SHIP && !SMI_ABLATED(16)) {
                    // Exact pre-filter of the ISOLATED TSO candidates.  A candidate's alignment matters in two ways only: it may be
                    // accepted (Math.round(ne) <= 5, AdapterTSOanalyzer L96-104), or its error count makes the scan jump over the next
                    // candidates (delta = round(ne - 5) - 1 <= 26: an alignment of two 16-mers has at most 32 columns).  A candidate
                    // without another candidate of its end among the next 26 positions can only matter by being accepted, and it
                    // cannot be when the bit-vector bound min_s Levenshtein(TSO, slice[s ..]) (s <= 5 leading template gaps, which is
                    // all an alignment with ne < 5.5 can have) exceeds 5 -- smi_nw.h myers_bound_tso16, 16 two-cycle operations per
                    // pattern base instead of a 184-cell fill and a walk.  About a third of the TSO candidates go this way (the
                    // chance hits on the end that has no TSO, and behind the true site on the end that has one); the generic
                    // kernels align every candidate, and the parity suite runs both.
                    auto shr128 = [](uint64_t &lo, uint64_t &hi, int k) {
                        lo = (lo >> k) | (hi << (64 - k));
                        hi >>= k;
                    };
                    uint64_t slo = tm[0], shi = tm[1];
                    shr128(slo, shi, 1);  // a candidate 1 position further on
#pragma unroll
                    for (int k = 1; k <= 8; k <<= 1) {  // ... 1..2, 1..4, 1..8, 1..16
                        uint64_t a = slo, b = shi;
                        shr128(a, b, k);
                        slo |= a;
                        shi |= b;
                    }
                    {
                        uint64_t a = slo, b = shi;
                        shr128(a, b, 10);  // ... 1..26
                        slo |= a;
                        shi |= b;
                    }
                    const uint64_t iso0 = tm[0] & ~slo, iso1 = tm[1] & ~shi;
                    int tot_iso;
                    const int off_iso = wave_exscan(__popcll(iso0) + __popcll(iso1), lane, tot_iso);
                    if (tot_iso) {  // wave-uniform
                        cmask[0 * kBlock + tid] = iso0;
                        cmask[1 * kBlock + tid] = iso1;
                        coff[tid] = (uint32_t)off_iso;
                        uint32_t *drop = ent + tid * 5;  // three words of dropped positions per end
                        drop[0] = drop[1] = drop[2] = 0u;
                        wave_sync();
                        for (int base = 0; base < tot_iso; base += 64) {
                            const int en = base + lane;
                            if (en < tot_iso) {
                                int lo = 0, hi = 64;
                                while (hi - lo > 1) {
                                    const int mid = (lo + hi) >> 1;
                                    if ((int)coff[wbase + mid] <= en)
                                        lo = mid;
                                    else
                                        hi = mid;
                                }
                                const int owner = wbase + lo;
                                const int k = en - (int)coff[owner];
                                const uint64_t mm = cmask[0 * kBlock + owner];
                                const int c0 = __popcll(mm);
                                const int bit = k < c0 ? kth_bit(mm, k) : 64 + kth_bit(cmask[1 * kBlock + owner], k - c0);  // scan position - 1
                                uint32_t V[4];
#pragma unroll
                                for (int c = 0; c < 4; c++) V[c] = __brev(get32(planes + c * kLdsWords * kBlock, owner, bit)) >> 16;
                                if (myers_bound_tso16(V, 5) > 5) atomicOr(&ent[owner * 5 + (bit >> 5)], 1u << (bit & 31));
                            }
                        }
                        wave_sync();
                        tm[0] &= ~((uint64_t)drop[0] | ((uint64_t)drop[1] << 32));
                        tm[1] &= ~(uint64_t)drop[2];
                        wave_sync();  // cmask / coff / ent are written again below
                    }
                }
                const int n_ts = __popcll(tm[0]) + __popcll(tm[1]);
                no = (uint32_t)n_ts | ((uint32_t)wave_exscan(n_ts, lane, total) << 8);
                cmask[0 * kBlock + tid] = tm[0];
                cmask[1 * kBlock + tid] = tm[1];
                cmask[2 * kBlock + tid] = 0ull;
            }
            coff[tid] = no >> 8;
            wave_sync();
            const uint32_t *offs = coff;
            const int my_off = (int)(no >> 8), my_n = (int)(no & 0xFFu);
            for (int base = 0; base < total; base += 64) {
                const int en = base + lane;
                if (en < total) {
                    // owner: last lane of this wave with offs[lane] <= en
                    int lo = 0, hi = 64;
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if ((int)offs[wbase + mid] <= en)
                            lo = mid;
                        else
                            hi = mid;
                    }
                    const int owner = wbase + lo;
                    int k = en - (int)offs[owner];
                    int pos = 0;
                    for (int ch = 0; ch < 3; ch++) {
                        const uint64_t m = cmask[ch * kBlock + owner];
                        const int c = __popcll(m);
                        if (k < c) {
                            pos = ch * 64 + kth_bit(m, k) + 1;
                            break;
                        }
                        k -= c;
                    }
                    AlnStats st;
                    st.ne = 9.0f, st.end5 = st.endn = 0.0f, st.nmis = 9, st.ins = st.del = st.consec = st.best_two = 0, st.term6 = false;
                    // the column masks of the alignment are the four base planes of the read slice, selected per pattern
                    // base: four window fetches per candidate, not one per pattern base
                    uint32_t W[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) W[c] = get32(planes + c * kLdsWords * kBlock, owner, pos - 1);
                    auto col_of = [&](uint32_t a4) -> uint32_t {
                        return ((a4 & 1u) ? W[0] : 0u) | ((a4 & 2u) ? W[1] : 0u) | ((a4 & 4u) ? W[2] : 0u) | ((a4 & 8u) ? W[3] : 0u);
                    };
                    if (kind == 0 && !SMI_ABLATED(2)) {
                        uint32_t col[AD];
#pragma unroll
                        for (int c = 0; c < AD; c++) col[c] = col_of(SHIP ? shipped_a4<AD>(c) : P.a4(c)) & ((1u << AD) - 1u);
                        nw_full<AD, true, false, kBandAd>(col, P.min_3p, st);  // the adapter fold reads ne, nmis, ins, del, end5, endn, term6
                    } else if (kind == 1 && !SMI_ABLATED(1)) {
                        uint32_t col[16];
#pragma unroll
                        for (int c = 0; c < 16; c++) col[c] = col_of(SHIP ? tso4(c) : P.t4(c)) & 0xFFFFu;
                        nw_full<16, false, true, kBandTso>(col, 0, st);  // the TSO rules read ne, nmis, ins, del, consec, best_two
                    }
                    uint32_t *o = ent + tid * 5;
                    o[0] = __float_as_uint(st.ne);
                    o[1] = __float_as_uint(st.end5);
                    o[2] = __float_as_uint(st.endn);
                    o[3] = (uint32_t)(st.nmis & 0xFF) | ((uint32_t)(st.ins & 0xFF) << 8) | ((uint32_t)((st.del + 128) & 0xFF) << 16) |
                           ((uint32_t)st.term6 << 24);
                    o[4] = (uint32_t)(st.consec & 0xFF) | ((uint32_t)(st.best_two & 0xFF) << 8) | ((uint32_t)pos << 16);
                }
                wave_sync();
                // owners fold their entries of this tile, in scan order
                const int f0 = max(my_off, base), f1 = SMI_ABLATED(128) ? f0 : min(my_off + my_n, base + 64);
                for (int x = f0; x < f1; x++) {
                    const uint32_t *o = ent + (wbase + x - base) * 5;
                    const float ne = __uint_as_float(o[0]);
                    const int nmis = (int)(o[3] & 0xFF);
                    const int pos = (int)(o[4] >> 16);
                    const uint32_t packed = (uint32_t)pos | (o[3] << 8);  // the term6 bit falls off the top
                    if (kind == 0) {
                        if (ne < a_best) {
                            a_best = ne;
                            a_have = false;
                        }
                        if (ne == a_best) {
                            // createNeedlemanMatch L242-247: accepted with <= maxMM errors or 6 terminal matches
                            const bool ok = nmis <= P.max_mm || ((o[3] >> 24) & 1u);
                            const float key = __uint_as_float(o[1]);
                            // several best positions: smallest countIndelsMismatchesEndOfRead(5), first of its group
                            if (ok && (!a_have || key < a_key)) {
                                a_have = true;
                                a_key = key;
                                a_pack = packed;
                                a_endn = __uint_as_float(o[2]);
                            }
                        }
                    } else if (pos >= (int)(t_aux >> 16)) {  // positions jumped over by deltaPos are never aligned (L100-106)
                        const float t_max = SHIP ? 5.0f : (float)P.tso_max_mm;  // maxErrors = maxNeedlemanMismatches of the TSO
                        if (!((float)(int)floorf(__fadd_rn(ne, 0.5f)) > t_max) && ne < t_best) {  // Math.round(ne) <= maxErrors
                            t_best = ne;
                            t_pack = packed;
                            t_aux = (t_aux & 0xFFFF0000u) | (o[4] & 0xFFFFu);
                        }
                        if (t_max < ne) {
                            int d = (int)floorf(__fadd_rn(__fsub_rn(ne, t_max), 0.5f)) - 1;
                            t_aux = (t_aux & 0xFFFFu) | ((uint32_t)(pos + (d < 1 ? 1 : d)) << 16);
                        }
                    }
                }
                wave_sync();
            }
        }

        // ---- phase C: strand decision (PolyATadapterAnalyzerBase.analyze L145-163): lanes 2i and 2i+1 exchange ----
        const int o_has_t = __shfl_xor((int)has_t, 1);
        const int n_ad_c = (int)(no_ad & 0xFFu);
        const int o_n_all = __shfl_xor(n_ad_c, 1);
        const float o_best = __shfl_xor(a_best, 1);
        const bool f_has = side == 0 ? has_t : (bool)o_has_t, r_has = side == 0 ? (bool)o_has_t : has_t;
        const int f_n = side == 0 ? n_ad_c : o_n_all, r_n = side == 0 ? o_n_all : n_ad_c;
        const float f_best = side == 0 ? a_best : o_best, r_best = side == 0 ? o_best : a_best;
        uint32_t flags = 0;
        int use_fwd = -1;
        if (!long_enough) {
            flags |= SMI_F_READ_TOO_SHORT | SMI_F_FAILED;
        } else {
            if (!(FP && P.dont_polya))
                flags |= (!f_has && !r_has) ? SMI_F_POLY_A_NOT_FOUND
                         : (f_has && !r_has) ? SMI_F_POLY_T_5P
                         : (!f_has && r_has) ? SMI_F_POLY_A_3P
                                             : SMI_F_POLY_T_5P_POLY_A_3P;
            // an end that was not scanned has no candidates, so "scan result present and not empty" is n > 0
            const bool f_ne = f_n > 0, r_ne = r_n > 0;
            if (f_ne && r_ne) {
                if (fabsf(__fsub_rn(f_best, r_best)) < 2.0f)
                    flags |= SMI_F_ADAPTER_5P_AND_3P;
                else {
                    flags |= SMI_F_ADAPTER_SELECTED_DESP_BOTH;
                    use_fwd = f_best < r_best ? 1 : 0;
                }
            } else if (f_ne)
                use_fwd = 1;
            else if (r_ne)
                use_fwd = 0;
            if (use_fwd < 0) flags |= SMI_F_FAILED;
        }
        const bool chosen = active && use_fwd >= 0 && side == (use_fwd ? 0 : 1);
        const int a_pos = (int)(a_pack & 0xFF), a_nmis = (int)((a_pack >> 8) & 0xFF), a_ins = (int)((a_pack >> 16) & 0xFF),
                  a_del = (int)(a_pack >> 24) - 128;
        const int t_pos = (int)(t_pack & 0xFF), t_nmis = (int)((t_pack >> 8) & 0xFF), t_ins = (int)((t_pack >> 16) & 0xFF),
                  t_del = (int)(t_pack >> 24) - 128, t_consec = (int)(t_aux & 0xFF), t_two = (int)((t_aux >> 8) & 0xFF);

        smi_scan_result res;
        res.flags = flags;
        res.adapter_end = 0;
        res.adapter_start = 0;
        res.polya_start = 0;
        res.polya_end = 0;
        res.scan_end = 0;
        res.adapter_nmis = 0;
        res.found = 0;
        res.reverse = 0;
        res.pass1_ok = 0;
        res.reserved = 0;
        res.tso_start = 0;
        res.tso_end = 0;
        smi_bc_window win;
        win.bases = 0;
        win.nmask = 0;
        win.flags = 0;
        const int pb_c = (int)(pbe & 0xFFFFu), pe_c = (int)(pbe >> 16);
        const int o_pe = __shfl_xor(pe_c, 1), o_pb = __shfl_xor(pb_c, 1);
        if (chosen) {
            // polyA coordinates are set as soon as a side is chosen (analyze L169-171); 5' barcoding takes the polyT
            // result of the other end, and none with --noPolyARequired
            if (!FP) {
                res.polya_start = len - (pe_c - 1);
                res.polya_end = len - (pb_c - 1);
            } else if (!P.dont_polya) {
                res.polya_start = len - (o_pe - 1);
                res.polya_end = len - (o_pb - 1);
            }
            if (!a_have) {
                res.flags |= SMI_F_FAILED;  // L217
            } else {
                const int s_end = a_pos + AD - 1 + a_ins - a_del;  // createNeedlemanMatch L251
                res.found = 1;
                res.scan_end = (int16_t)s_end;
                res.adapter_nmis = (int16_t)a_nmis;
                if (!FP) {
                    res.adapter_start = len - (a_pos - 1);  // ReadScanResult.java:L446-447
                    res.adapter_end = len - (s_end - 1);
                    res.reverse = use_fwd ? 1 : 0;
                    res.flags |= use_fwd ? (SMI_F_ADAPTER_5P | SMI_F_PASSED_REV) : (SMI_F_ADAPTER_3P | SMI_F_PASSED_FWD);
                } else {
                    res.adapter_start = a_pos;  // L449-450: scan coordinates = stranded coordinates
                    res.adapter_end = s_end;
                    res.reverse = use_fwd ? 0 : 1;
                    res.flags |= use_fwd ? (SMI_F_ADAPTER_5P | SMI_F_PASSED_FWD) : (SMI_F_ADAPTER_3P | SMI_F_PASSED_REV);
                }
                // barcode window.  3': stranded[AE-22 .. AE+1] = reverse complement of scan[s_end-1 .. s_end+22];
                // 5': stranded[AE-1 .. AE+23] = scan[s_end-1 .. s_end+23] as it stands (Parser.java:L205-221)
                const int n_win = FP ? SMI_WIN_BASES_5P : SMI_WIN_BASES_3P;
                const int hi_sp = s_end - 2 + n_win, lo_sp = s_end - 1;
                if (lo_sp >= 1 && hi_sp <= len && hi_sp <= kEndBases) {
                    // bit-parallel on one 32-bit window per plane (bit i = scan position lo_sp + i): a base is kept when
                    // exactly one plane has it (N and '-' go to the mask); 2-bit codes A0 G1 C2 T3, complemented for 3'
                    const uint32_t wm = (1u << n_win) - 1u;
                    const uint32_t va = get32(planes + 0 * kLdsWords * kBlock, tid, lo_sp - 1);
                    const uint32_t vg = get32(planes + 1 * kLdsWords * kBlock, tid, lo_sp - 1);
                    const uint32_t vc = get32(planes + 2 * kLdsWords * kBlock, tid, lo_sp - 1);
                    const uint32_t vt = get32(planes + 3 * kLdsWords * kBlock, tid, lo_sp - 1);
                    const uint32_t single = (va ^ vg ^ vc ^ vt) & ~((va & vg & (vc | vt)) | (vc & vt & (va | vg))) & wm;
                    uint32_t hi = (FP ? (vc | vt) : (vg | va)) & single, lo = (FP ? (vg | vt) : (vc | va)) & single;
                    uint32_t nmask = ~single & wm;
                    // window base j sits at bit n_win-1-j of the 3' window (the reverse complement reads it backwards)
                    // and at bit j of the 5' one; base 0 is the most significant 2-bit group of `bases`
                    if (FP) {
                        hi = __brev(hi) >> (32 - n_win);
                        lo = __brev(lo) >> (32 - n_win);
                    } else {
                        nmask = __brev(nmask) >> (32 - n_win);
                    }
                    auto spread = [](uint32_t x) {
                        uint64_t v = x;
                        v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
                        v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
                        v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
                        v = (v | (v << 2)) & 0x3333333333333333ull;
                        v = (v | (v << 1)) & 0x5555555555555555ull;
                        return v;
                    };
                    const uint64_t bases = (spread(hi) << 1) | spread(lo);
                    win.bases = bases;
                    win.nmask = nmask;
                    win.flags = SMI_WIN_VALID | (FP ? SMI_WIN_5P : 0u);
                }
                // pass-1 quality filter (short-circuit && chain; the UNSTRANDED quality string is indexed with
                // stranded coordinates, UsedCellBCListGenerator.java:L201)
                if (qtail != nullptr && a_endn == 0.0f) {
                    const int ae = res.adapter_end;
                    int sum = 0;
                    bool in_range = ae - 16 >= 1;
                    {
                        // 3': the last kEndBases qualities, right-aligned; 5': the first kEndBases, left-aligned.  The sixteen
                        // qualities are consecutive bytes of the tail: one 16-byte load summed with v_sad_u8 (a loop of sixteen
                        // dependent byte loads per chosen lane before)
                        const int p0 = ae - 16;
                        const int idx0 = FP ? p0 - 1 : kEndBases - 1 - (len - p0);
                        if (idx0 < 0 || idx0 + 15 >= kEndBases) in_range = false;
                        if (in_range) {
                            uint32_t w[4];
                            __builtin_memcpy(w, qtail + (size_t)read * kEndBases + idx0, 16);
                            uint32_t acc = 0;
#pragma unroll
                            for (int k = 0; k < 4; k++) acc = __builtin_amdgcn_sad_u8(w[k], 0u, acc);
                            sum = (int)acc - 16 * 33;
                        }
                    }
                    if (in_range) {
                        const float q_bc = (float)((double)sum / 16.0);
                        if (!(q_bc < (float)P.min_bc_qv)) {
                            const float q_read = (float)((double)qsum[read] / (double)len);
                            res.pass1_ok = !(q_read < (float)P.min_read_qv) ? 1 : 0;
                        }
                    } else {
                        res.reserved = 1;  // the reference would throw (IntStream.skip(negative)); never seen
                    }
                }
            }
        }
        // ---- TSO rules on both ends (PolyATadapterAnalyzer_3pBCUMI.scanReadForTSOs L122-190) ----------------------
        {
            struct Tm {
                int present, passed, nmis, end_scan, consec, two;
            };
            Tm mine;
            mine.present = t_pos != 0;
            mine.nmis = t_nmis;
            mine.passed = mine.present && t_nmis <= (SHIP ? 5 : P.tso_max_mm);  // scanForTSO L350
            mine.end_scan = t_pos + 15 + t_ins - t_del;
            mine.consec = t_consec;
            mine.two = t_two;
            Tm o;
            o.present = __shfl_xor(mine.present, 1);
            o.passed = __shfl_xor(mine.passed, 1);
            o.nmis = __shfl_xor(mine.nmis, 1);
            o.end_scan = __shfl_xor(mine.end_scan, 1);
            o.consec = __shfl_xor(mine.consec, 1);
            o.two = __shfl_xor(mine.two, 1);
            Tm f = side == 0 ? mine : o, r = side == 0 ? o : mine;
            auto found = [](const Tm &x) { return x.present && x.passed; };
            if (!found(f) && !found(r)) {  // L146-153: rescue by >= 8 consecutive matches (config.xml:161)
                const int min_consec = SHIP ? 8 : P.tso_min_consec, min_two = SHIP ? 12 : P.tso_min_two;
                if (f.present) f.passed = f.consec >= min_consec;
                if (r.present) r.passed = r.consec >= min_consec;
                if (!found(f) && !found(r)) {  // L155-162: rescue by the two stretches >= 12 (config.xml:164)
                    if (f.present) f.passed = f.two >= min_two;
                    if (r.present) r.passed = r.two >= min_two;
                }
            }
            if (found(f) && found(r) && abs(f.nmis - r.nmis) > 3) {  // L167-172
                if (f.nmis > r.nmis)
                    f.present = 0;
                else
                    r.present = 0;
            }
            const bool ff = found(f), rf = found(r);
            res.tso_start = ff ? (int16_t)f.end_scan : (int16_t)0;  // TSOresult.start/.end = end of the match in scan
            res.tso_end = rf ? (int16_t)r.end_scan : (int16_t)0;    // coordinates (L177-181)
            if (long_enough) res.flags |= (ff && !rf) ? SMI_F_TSO_5P : (!ff && rf) ? SMI_F_TSO_3P : (ff && rf) ? SMI_F_TSO_5P_AND_3P : 0u;
        }
        // one record per read: written by the chosen lane, else by the even lane
        const int partner_chosen = __shfl_xor((int)chosen, 1);
        if (active && (chosen || (side == 0 && !partner_chosen))) {
            out[read] = res;
            if (windows) windows[read] = win;
        }
        wave_sync();  // the next iteration overwrites this wave's columns
    }
}

static size_t scan_lds_bytes() {
    size_t bytes = (size_t)4 * kLdsWords * kBlock * 4 + (size_t)3 * kBlock * 8 + (size_t)kBlock * 4 + (size_t)kBlock * 5 * 4;
#ifdef SMI_MEASURE
    // measurement builds only: extra (unused) LDS per block, to hold K-SCAN to fewer blocks per CU when another kernel should run beside it
    if (const char *pad = std::getenv("SMI_SCAN_LDS_PAD")) bytes += (size_t)std::atoi(pad);
#endif
    return bytes;
}

static int thr_for(int len, float limit_f, bool use_double, double limit_d) {
    for (int k = 0; k <= len; k++) {
        const float v = (float)k / (float)len;
        if (use_double ? ((double)v >= limit_d) : (v >= limit_f)) return k;
    }
    return len + 1;
}

int launch_scan(smi_ctx *ctx, const uint32_t *d_ends, const int32_t *d_len, const uint8_t *d_qtail,
                const uint32_t *d_qsum, size_t n, const smi_scan_config *cfg, smi_scan_result *d_out,
                smi_bc_window *d_win, hipStream_t s) {
    if (!n) return SMI_OK;
    ScanParams P;
    P.min_read_length = cfg->min_read_length;
    P.polya_len = cfg->polya_len;
    P.window = cfg->window_polya;
    P.thr_first = thr_for(cfg->polya_len, cfg->polya_frac, false, 0.0);
    P.thr_adv = thr_for(cfg->polya_len, 0.0f, true, (double)cfg->polya_frac - 0.1);
    P.max_mm = cfg->max_mismatches;
    P.min_3p = cfg->min_adapter_3p_matches;
    P.min_bc_qv = cfg->min_mean_bc_qv;
    P.min_read_qv = cfg->min_mean_read_qv;
    const int ad = cfg->adapter_len;
    P.adapter_nib[0] = P.adapter_nib[1] = P.adapter_nib[2] = 0u;
    for (int i = 0; i < 22 && i < ad; i++) P.adapter_nib[i >> 3] |= (cfg->adapter4[i] & 15u) << ((i & 7) * 4);
    P.dont_polya = cfg->dont_search_polya;
    P.window5 = cfg->adapter_search_window;
    // the TSO of the read scan: tso_window == 0 (a configuration from before the knob) = the shipped parameters
    const bool tso_given = cfg->tso_window != 0;
    P.tso_nib[0] = P.tso_nib[1] = 0u;
    for (int i = 0; i < 16; i++) P.tso_nib[i >> 3] |= ((tso_given ? cfg->tso4[i] : tso4(i)) & 15u) << ((i & 7) * 4);
    P.tso_window = tso_given ? cfg->tso_window : 90;
    P.tso_max_mm = tso_given ? cfg->tso_max_mismatches : 5;
    P.tso_min_consec = tso_given ? cfg->tso_min_consec : 8;
    P.tso_min_two = tso_given ? cfg->tso_min_two_best : 12;
    // the kernels of the shipped adapters carry the bit-parallel finder, built for the shipped window length and the thresholds the shipped fractions give;
    // SMI_SCAN_FINDER_LOOP: cross-check switch (the generic kernels, which keep the loop)
    P.finder_bits = cfg->polya_len == 15 && P.thr_first == 12 && P.thr_adv == 10 && cfg->window_polya >= 1 && cfg->window_polya <= 160 && !std::getenv("SMI_SCAN_FINDER_LOOP");
    // SMI_SCAN_ABLATE switches parts of the kernel off to time them (tools/gpu_scan_ablate.sh): the results are wrong by construction, so
    // only a measurement build (make MEASURE=1 -> -DSMI_MEASURE) honours it; the shipped library refuses to run with it set
    const char *abl = std::getenv("SMI_SCAN_ABLATE");
#ifdef SMI_MEASURE
    P.ablate = abl ? std::atoi(abl) : 0;
#else
    P.ablate = 0;
    if (abl && std::atoi(abl) != 0) {
        set_error("smi_scan_device: SMI_SCAN_ABLATE is set, which yields wrong results by construction; it is honoured by measurement builds only (make MEASURE=1)");
        return SMI_ERR_INVALID;
    }
#endif
    const size_t n_ends = 2 * n;
    const unsigned grid = (unsigned)std::min<size_t>((n_ends + kBlock - 1) / kBlock, 256 * 64);  // measured: x4 8.2, x16 7.5, x64 7.2, uncapped 7.7 ms
    if (int rc = time_begin(ctx, SMI_K_SCAN, s)) return rc;
    // the dynamic LDS size (40 KiB) is below the 64 KiB that needs an opt-in
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), scan_lds_bytes(), s, d_ends, d_len, d_qtail, d_qsum, n, P, d_out,
                           d_win);
    };
    bool ship = (ad == 10 || ad == 22) && P.finder_bits;
    for (int i = 0; ship && i < ad; i++) ship = P.a4(i) == (ad == 10 ? shipped_a4<10>(i) : shipped_a4<22>(i));
    // (the shipped kernels carry the TSO, its window and its limits as constants, and the isolated-candidate pre-filter is derived for them)
    ship = ship && P.tso_window == 90 && P.tso_max_mm == 5 && P.tso_min_consec == 8 && P.tso_min_two == 12;
    for (int i = 0; ship && i < 16; i++) ship = P.t4(i) == tso4(i);
    if (std::getenv("SMI_SCAN_GENERIC")) ship = false;  // tests run both builds against the oracle
    if (ad == 10) {
        if (cfg->five_prime)
            ship ? launch(k_scan<10, true, true>) : launch(k_scan<10, true, false>);
        else
            ship ? launch(k_scan<10, false, true>) : launch(k_scan<10, false, false>);
    } else {
        if (cfg->five_prime)
            ship ? launch(k_scan<22, true, true>) : launch(k_scan<22, true, false>);
        else
            ship ? launch(k_scan<22, false, true>) : launch(k_scan<22, false, false>);
    }
    SMI_HIP(hipGetLastError());
    if (int rc = time_end(ctx, SMI_K_SCAN, s)) return rc;
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// K-PACK: ASCII reads -> scan-orientation bit-plane ends (+ read length, tail qualities, quality sum).
// Bases: one lane per read end (below); qualities: one wave per read.
// ---------------------------------------------------------------------------------------------------------------
// one lane = one read end (lanes 2i / 2i+1 = head / reverse-complemented tail of read i): every lane walks its 224
// bases in 8-byte pieces and builds the four plane words in registers, so the stores of a wave are 256-B rows of the
// [plane-word][end] layout and no cross-lane operation is needed
// starts != nullptr: record r's bases begin at reads[starts[r]] (reads = the FASTQ text itself, no gathered copy); lengths always come
// from the offsets prefix array
// (kStarts: the bases sit in the FASTQ text at starts[r] -- a template parameter, so that the load of starts[r] is issued beside the one of the
// offsets instead of behind a test that waits for them)
template <bool kStarts>
__global__ __launch_bounds__(256) void k_pack_ends(const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets,
                                                   const uint64_t *__restrict__ starts, size_t n, uint32_t *__restrict__ ends,
                                                   int32_t *__restrict__ read_len) {
    const size_t n_ends = 2 * n;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n_ends; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e >> 1;
        const int side = (int)(e & 1);
        const uint64_t at = kStarts ? starts[r] : 0;
        const uint64_t beg = offsets[r];
        const int64_t len = (int64_t)(offsets[r + 1] - beg);
        if (side == 0) read_len[r] = (int32_t)len;
        const uint8_t *src = reads + (kStarts ? at : beg);
        // 16 bases per piece: one 16-byte load (unaligned), four enc4x4 -- the tail end is read backwards and complemented, which is a
        // byte swap of the piece and the complement table
        if (len >= kEndBases) {
            // the usual case, a read of 224 bases or more: its fourteen pieces are requested together (one round trip to the text instead
            // of seven), then encoded
            uint32_t vv[2 * kPlaneWords][4];
#pragma unroll
            for (int k = 0; k < 2 * kPlaneWords; k++)
                __builtin_memcpy(vv[k], side == 0 ? src + 16 * k : src + (len - 16 - 16 * k), 16);
#pragma unroll
            for (int w = 0; w < kPlaneWords; w++) {
                uint32_t pl[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 2; k++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t code = side ? enc4x4<true>(__builtin_bswap32(vv[2 * w + k][3 - j])) : enc4x4<false>(vv[2 * w + k][j]);
#pragma unroll
                        for (int c = 0; c < 4; c++) pl[c] |= plane_nibble(code, c) << (16 * k + 4 * j);
                    }
#pragma unroll
                for (int c = 0; c < 4; c++) ends[(size_t)(c * kPlaneWords + w) * n_ends + e] = pl[c];
            }
            continue;
        }
#pragma unroll
        for (int w = 0; w < kPlaneWords; w++) {
            uint32_t pl[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int p0 = 32 * w + 16 * k;  // scan positions p0 .. p0 + 15
                if (p0 >= len) break;
                uint32_t v[4] = {0, 0, 0, 0};    // v[j] byte i = character at scan position p0 + 4j + i
                if (p0 + 16 <= len) {
                    if (side == 0) {
                        __builtin_memcpy(v, src + p0, 16);
                    } else {
                        uint32_t t[4];
                        __builtin_memcpy(t, src + (len - 16 - p0), 16);
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] = __builtin_bswap32(t[3 - j]);
                    }
                } else {  // the read ends inside this piece
                    for (int i = 0; i < (int)(len - p0); i++)
                        v[i >> 2] |= (uint32_t)src[side == 0 ? p0 + i : len - 1 - p0 - i] << (8 * (i & 3));
                }
                const int nb = (int)min((int64_t)16, len - p0);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t code = side ? enc4x4<true>(v[j]) : enc4x4<false>(v[j]);
                    if (4 * j + 4 > nb) code &= 4 * j >= nb ? 0u : (0xFFFFFFFFu >> (8 * (4 * j + 4 - nb)));
#pragma unroll
                    for (int c = 0; c < 4; c++) pl[c] |= plane_nibble(code, c) << (16 * k + 4 * j);
                }
            }
            ends[(size_t)(0 * kPlaneWords + w) * n_ends + e] = pl[0];
            ends[(size_t)(1 * kPlaneWords + w) * n_ends + e] = pl[1];
            ends[(size_t)(2 * kPlaneWords + w) * n_ends + e] = pl[2];
            ends[(size_t)(3 * kPlaneWords + w) * n_ends + e] = pl[3];
        }
    }
}

// qualities (pass 1 only): one wave per read sums the whole string (coalesced) and copies the 224 qualities the
// pass-1 filter may look at (3': the last ones, right-aligned; 5': the first ones)
__global__ __launch_bounds__(256) void k_pack_quals(const uint8_t *__restrict__ quals, const uint64_t *__restrict__ offsets,
                                                    size_t n, int head_quals, uint8_t *__restrict__ qtail,
                                                    uint32_t *__restrict__ qsum) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < n; r += n_waves) {
        const uint64_t beg = offsets[r];
        const int64_t len = (int64_t)(offsets[r + 1] - beg);
        // sum of the characters, 16 per lane and step (v_sad_u8 adds the four bytes of a dword), minus 33 per character
        uint32_t s = 0;
        const int64_t body = len & ~(int64_t)15;
        for (int64_t i = 16 * (int64_t)lane; i < body; i += 1024) {
            uint32_t w[4];
            __builtin_memcpy(w, quals + beg + i, 16);
#pragma unroll
            for (int k = 0; k < 4; k++) s = __builtin_amdgcn_sad_u8(w[k], 0u, s);
        }
        if (lane < (int)(len - body)) s += (uint32_t)quals[beg + body + lane];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) qsum[r] = s - 33u * (uint32_t)len;
        if (len >= kEndBases) {  // the 224 qualities the filter may look at: fourteen lanes, 16 bytes each
            if (lane < kEndBases / 16) {
                uint32_t w[4];
                __builtin_memcpy(w, quals + beg + (head_quals ? 0 : len - kEndBases) + 16 * lane, 16);
                __builtin_memcpy(qtail + r * kEndBases + 16 * lane, w, 16);
            }
        } else {
            for (int i = lane; i < kEndBases; i += 64) {
                const int64_t p = head_quals ? i : len - kEndBases + i;
                qtail[r * kEndBases + i] = (p >= 0 && p < len) ? quals[beg + p] : (uint8_t)33;
            }
        }
    }
}

// K-PACK for the packed boundary: the read ends cut out of read PLANES (smi_pack_reads_host / K-PACKR) instead of ASCII.  One lane = one
// end, as in k_pack_ends.  Record i = bases [rec_offsets[i], rec_offsets[i+1]) of the chunk inside read src (frag_src[i] >> 2, or i):
// bit b of word w of a plane = base 32 w + b of the read, so the head is a 224-bit funnel shift out of each plane and the reverse-
// complemented tail the same bits mirrored (v_bfrev) with the planes swapped A <-> T, G <-> C (N = all four bits stays N).
__device__ __forceinline__ uint32_t plane_bits32(const uint32_t *pl, int64_t bitpos) {  // 32 bits from bit position bitpos >= 0
    const int64_t w = bitpos >> 5;
    const int sh = (int)(bitpos & 31);
    const uint64_t two = ((uint64_t)pl[w + 1] << 32) | pl[w];
    return (uint32_t)(two >> sh);
}
__global__ __launch_bounds__(256) void k_ends_from_planes(const uint32_t *__restrict__ planes, size_t stride, const uint32_t *__restrict__ pstart,
                                                          const uint64_t *__restrict__ read_offsets,
                                                          const uint64_t *__restrict__ rec_offsets, const uint32_t *__restrict__ frag_src, size_t m,
                                                          uint32_t *__restrict__ ends, int32_t *__restrict__ read_len) {
    const size_t n_ends = 2 * m;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n_ends; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e >> 1;
        const int side = (int)(e & 1);
        const size_t src = frag_src ? (size_t)(frag_src[i] >> 2) : i;
        const uint64_t rbeg = read_offsets[src], beg = rec_offsets[i];
        const int64_t len = (int64_t)(rec_offsets[i + 1] - beg);
        const int64_t fb = (int64_t)(beg - rbeg);  // first base of the record inside its read
        if (side == 0) read_len[i] = (int32_t)len;
        const uint32_t *p0 = planes + (pstart ? (size_t)pstart[src] : plane_start(rbeg, src));
        if (len >= 32 * kPlaneWords) {
            // whole end: the 224 bases sit in eight consecutive words of each plane -- two 16-byte loads per plane instead of fourteen
            // 4-byte ones (a lane's addresses are its own: the loads of a wave do not coalesce, so their number is what costs)
            struct __attribute__((aligned(4))) W4 {
                uint32_t x, y, z, w;
            };
            const int64_t b0 = side == 0 ? fb : fb + len - 32 * kPlaneWords;  // first base of the end inside the read
            const uint32_t *q0 = p0 + (b0 >> 5);
            const int sh = (int)(b0 & 31);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const W4 lo = *reinterpret_cast<const W4 *>(q0 + c * stride), hi = *reinterpret_cast<const W4 *>(q0 + c * stride + 4);
                const uint32_t r[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int w = 0; w < kPlaneWords; w++) {
                    if (side == 0)
                        ends[(size_t)(c * kPlaneWords + w) * n_ends + e] = (uint32_t)((((uint64_t)r[w + 1] << 32) | r[w]) >> sh);
                    else  // scan position 32 w + b = base b0 + 223 - 32 w - b: the word at 6 - w mirrored, planes swapped
                        ends[(size_t)((3 - c) * kPlaneWords + w) * n_ends + e] =
                            __builtin_bitreverse32((uint32_t)((((uint64_t)r[7 - w] << 32) | r[6 - w]) >> sh));
                }
            }
            continue;
        }
#pragma unroll
        for (int w = 0; w < kPlaneWords; w++) {
            const int64_t nb = len - 32 * w;  // bases of this end word that exist
            uint32_t pl[4] = {0, 0, 0, 0};
            if (nb > 0) {
                const uint32_t keep = nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u);
                if (side == 0) {
#pragma unroll
                    for (int c = 0; c < 4; c++) pl[c] = plane_bits32(p0 + c * stride, fb + 32 * w) & keep;
                } else {
                    // scan position p = 32 w + b is read base fb + len - 1 - p: the 32 bases ending at fb + len - 1 - 32 w, mirrored; a partial
                    // word starts at the record's first base and is shifted down after the mirror
                    const int64_t q = nb >= 32 ? fb + len - 32 * (w + 1) : fb;
                    const int down = nb >= 32 ? 0 : (int)(32 - nb);
#pragma unroll
                    for (int c = 0; c < 4; c++) pl[3 - c] = (__builtin_bitreverse32(plane_bits32(p0 + c * stride, q) & keep)) >> down;
                }
            }
#pragma unroll
            for (int c = 0; c < 4; c++) ends[(size_t)(c * kPlaneWords + w) * n_ends + e] = pl[c];
        }
    }
}

int launch_ends_from_planes(smi_ctx *, const uint32_t *d_planes, size_t stride, const uint64_t *d_read_offsets, const uint64_t *d_rec_offsets,
                            const uint32_t *d_frag_src, size_t m, uint32_t *d_ends, int32_t *d_len, hipStream_t s, const uint32_t *d_pstart) {
    if (!m) return SMI_OK;
    const unsigned grid = (unsigned)std::min<size_t>((2 * m + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(k_ends_from_planes, dim3(grid), dim3(256), 0, s, d_planes, stride, d_pstart, d_read_offsets, d_rec_offsets, d_frag_src, m, d_ends, d_len);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

int launch_pack_ends(smi_ctx *, const uint8_t *d_reads, const uint8_t *d_quals, const uint64_t *d_offsets, const uint64_t *d_starts,
                     size_t n, int head_quals, uint32_t *d_ends, int32_t *d_len, uint8_t *d_qtail, uint32_t *d_qsum, hipStream_t s) {
    if (!n) return SMI_OK;
    const unsigned grid = (unsigned)std::min<size_t>((2 * n + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(d_starts ? k_pack_ends<true> : k_pack_ends<false>, dim3(grid), dim3(256), 0, s, d_reads, d_offsets, d_starts, n, d_ends, d_len);
    if (d_quals) {
        const unsigned gq = (unsigned)std::min<size_t>((n + 3) / 4, 256 * 32);
        hipLaunchKernelGGL(k_pack_quals, dim3(gq), dim3(256), 0, s, d_quals, d_offsets, n, head_quals, d_qtail, d_qsum);
    }
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

}  // namespace smi
