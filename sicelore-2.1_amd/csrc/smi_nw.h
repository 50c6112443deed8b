// smi_nw.h -- Needleman-Wunsch on gfx950 registers, shared by K-SCAN (smi_scan.hip) and K-CHIM (smi_chimera.hip).
//
// Reference units (bytecode; citation form in DESIGN.md):
//   NeedlemanWunsch.fillInCell / init            TB!nuc/alignment/needleman/NeedlemanWunsch.java:L55-80,L106-122
//   SequenceAlignment.getTraceback               TB!nuc/alignment/needleman/SequenceAlignment.java:L102-151
//   Match.countErrorsInNeedleman, NeedlemanMatch FJ!nanopore/analyzers/{Match,NeedlemanMatch}.java
#pragma once
#include <type_traits>

#include "smi_internal.h"

namespace smi {

// ---- Needleman-Wunsch with 2-bit moves in LDS and a walk from the end ---------------------------------------
// Scores (-4,-5,.,.,-5,-5,+5): NeedlemanParameters.java:L36-38.  Tie-breaks: NeedlemanWunsch.java:L55-80.
// The walk yields everything the reference reads off the alignment strings (SequenceAlignment.getTraceback L102-151):
//   Match.countErrorsInNeedleman L31-34 (#x - 0.9f * leading template gaps), NeedlemanMatch.countNeedlemanErrorsInRead
//   L68-86, countIndelsMismatchesEndOfRead L109-123, getNconsecutiveMatchesNeedleman L160-173,
//   getSumOfBestTwoMatchStretchesNeedleman L183-196, Match.hasN3pConsecutiveMatchesInNeedleman L41-50.
struct AlnStats {
    float ne;         // countErrorsInNeedleman
    float end5, endn; // countIndelsMismatchesEndOfRead(5) / (minAdapter3pMatches)
    int nmis, ins, del;
    int consec, best_two;
    bool term6;
};

// col[c] bit r = read base r of the slice matches pattern base c.
//
// Fill: one cell = 7 VALU ops.  A cell is kept as U = 4*score + 2 + 20*r (r = row): with the move as a 2-bit tag in
// the low bits (3 diag match, 2 diag mismatch, 1 up, 0 left) one v_max3_i32 over
//     diag' = U[r-1][c-1] + 41*m     up' = U[r-1][c] - 1     left' = U[r][c-1] - 22
// yields 4*best + tag + 20*r -- the tag order IS the reference's tie order (diag >= up >= left) -- and
// (v & ~3) | 2 is the stored cell again.  The tags of a row are shifted into one register (v_alignbit), the rows
// stay in registers (row loop fully unrolled), so the walk needs no LDS.
// Walk: rows N..1 unrolled; inside a row only consecutive left moves loop.
// kEnds: end5 / endn / term6 are wanted (the adapter fold); kRuns: consec / best_two are wanted (the TSO rules).  The walk
// keeps only the bookkeeping of the statistics its caller reads.
//
// Band.  Every caller aligns a candidate that passed the reference's 4-mer gate (> 1 matching 4-mers on the main diagonal of the very
// slice that is aligned), so at least 5 pattern bases match on the diagonal and the diagonal path scores >= 10 * 5 - 5N.  A path through
// a cell d off the diagonal has >= d gaps before and >= d after it and <= N - d diagonal steps: <= 5(N - d) - 9d (leading template gaps
// cost 4, every other gap 5).  For 14 d > 10 (N - 5) such a path is strictly worse than the diagonal, so no optimal path enters those
// cells -- and neither can a tie: a predecessor that ties with the chosen one lies on an optimal path itself.  Cells of the band whose
// best predecessor was cut off hold a smaller value than the reference's matrix, but they are not on an optimal path either; the cells
// on optimal paths, the moves out of them and their tie order are the reference's.  nw_band<N, MIN_DIAG>() is the largest |r - c| kept.
template <int N, int MIN_DIAG>
__host__ __device__ constexpr int nw_band() {
    int w = 0;
    while (w + 1 < N && 14 * (w + 1) <= 10 * (N - MIN_DIAG)) w++;
    return w;
}

// ---- the walk, bit-parallel ------------------------------------------------------------------------------------
// The path from (N, N) back to (0, 0) is at most 2N steps; step t is recorded as two bits in two masks (T1 T0: 11 match, 10 mismatch,
// 01 up = a base of the read against a template gap, 00 left = a read gap) and every statistic the reference reads off its alignment
// strings is a popcount, a count of trailing / leading bits or a short loop over runs of those masks.  The walk itself is one
// find-highest-bit per row: the left moves of a row are the zero tags between the current column and the next non-zero tag below
// it, so a row costs the same whatever the number of its left moves, and the rows need no inner loop.  (Until round 5 every step went
// through ~40 operations of bookkeeping, and a row with left moves ran them once per move for the whole wave.)
//   u = 2 * steps + 2 * column is unchanged by left and diagonal moves and grows by 2 on an up move: step index of a tag at column c2
//   = (u - 2 c2) / 2.  Rows walked = non-left steps, so the rows still above the path when it reaches column 0 (`lead`) = N - popcount.
template <int N>
struct NwWalkWord {
    using mask_t = typename std::conditional<(N <= 16), uint32_t, uint64_t>::type;
};
template <class W>
__device__ __forceinline__ W nw_lowmask(int n) {  // n in [0, bits of W]
    constexpr int B = (int)sizeof(W) * 8;
    return n >= B ? ~(W)0 : (((W)1 << n) - (W)1);
}
__device__ __forceinline__ int nw_popc(uint32_t x) { return __popc(x); }
__device__ __forceinline__ int nw_popc(uint64_t x) { return __popcll(x); }
__device__ __forceinline__ int nw_ctz(uint32_t x) { return x ? __builtin_ctz(x) : 32; }
__device__ __forceinline__ int nw_ctz(uint64_t x) { return x ? __builtin_ctzll(x) : 64; }
__device__ __forceinline__ int nw_top(uint32_t x) { return 31 - __builtin_clz(x); }   // x != 0
__device__ __forceinline__ int nw_top(uint64_t x) { return 63 - __builtin_clzll(x); }  // x != 0
// position of the k-th set bit of x (k >= 1), or the width of the word when x has fewer
template <class W>
__device__ __forceinline__ int nw_kth_bit(W x, int k) {
    for (int i = 1; i < k; i++) x &= x - (W)1;
    return nw_ctz(x);
}
// float sum of the reference's end-of-read error weights: k12 times (float)((double)e + 1.2), then k10 times e + 1.0f (the x-columns that
// lie over the first two read bases come first on the way back, countIndelsMismatchesEndOfRead L109-123)
__device__ __forceinline__ float nw_end_errors(int k12, int k10) {
    float e = 0.0f;
    for (int i = 0; i < k12; i++) e = (float)((double)e + 1.2);
    for (int i = 0; i < k10; i++) e = __fadd_rn(e, 1.0f);
    return e;
}

template <int N, bool kEnds, bool kRuns, int NM>
__device__ __forceinline__ void nw_walk_bits(const uint32_t (&mlo)[N], const uint32_t (&mhi)[NM], int n_end, AlnStats &out) {
    using MW = typename NwWalkWord<N>::mask_t;  // a row's tags (2 bits per column) and the step masks (<= 2N steps)
    constexpr int B = (int)sizeof(MW) * 8;
    MW T0 = 0, T1 = 0;
    int cc = 2 * N;  // 2 * current column
    int u = 2 * N;
    if constexpr (N <= 16) {
        // 32-bit rows.  sh = 32 - 2 * column: the row's tags shifted left by sh have the current column's tag on top, the leading zero bits
        // (rounded down to a pair) are the left moves, and every quantity of the step is an add or a shift of them -- no selects.
        int sh = 32 - 2 * N, up2 = 2 * N - 32;  // up2 = u - 32
#pragma unroll
        for (int R = N; R >= 1; R--) {
            if (sh < 32) {
                const uint32_t y = mlo[R - 1] << sh;
                const int k2 = __clz((int)y) & ~1;       // 2 * left moves (32: nothing but left moves down to column 0)
                const int sh2 = min(sh + k2, 32);
                const uint32_t tag = (y << (k2 & 31)) >> 30;  // y == 0: 0
                const uint32_t lo = tag & 1u, hi = tag >> 1;
                const int t = (up2 + sh2) >> 1;          // step index = (u - 2 * column of the tag) / 2
                T0 |= lo << (t & 31);
                T1 |= hi << (t & 31);
                up2 += (int)((lo & ~hi) << 1);           // an up move
                sh = sh2 + (int)(hi << 1);               // diagonal moves leave the column
            }
        }
        cc = 32 - sh;
        u = up2 + 32;
    } else {
#pragma unroll
        for (int R = N; R >= 1; R--) {
            if (cc > 0) {
                MW z = mlo[R - 1];
                z |= (MW)mhi[R - 1] << (B / 2);
                const MW below = z & nw_lowmask<MW>(cc);  // tags of the columns <= c
                // the first non-left tag at or below the column (none: the rest of the row is left moves down to column 0)
                const int pos = below ? (nw_top(below) & ~1) : -2;
                const uint32_t tag = below ? (uint32_t)(below >> pos) & 3u : 0u;
                const int cc2 = pos + 2;
                const int t = ((u - cc2) >> 1) & (B - 1);
                T0 |= (MW)(tag & 1u) << t;
                T1 |= (MW)(tag >> 1) << t;
                u += tag == 1u ? 2 : 0;
                cc = cc2 - (int)(tag & 2u);  // diagonal moves leave the column, an up move stays in it
            }
        }
    }
    // the rest of the path runs along the first column (`lead` up moves = leading template gaps) or the first row (left moves)
    const int lead = N - nw_popc((MW)(T0 | T1));
    int t_end = (u - cc) >> 1;
    T0 |= nw_lowmask<MW>(lead) << (t_end & (B - 1));
    t_end += lead + (cc >> 1);
    const MW valid = nw_lowmask<MW>(t_end);
    const MW Mt = T0 & T1, I = T0 & ~T1, S = T1 & ~T0, D = valid & ~(T0 | T1), X = valid & ~Mt;
    const int ins = nw_popc(I), sub = nw_popc(S), nx = nw_popc(X);
    const int trail = nw_ctz((MW)~D);  // read gaps at the end of the read (the start of the walk) are not deletions
    const int del = (int)(int8_t)(nw_popc(D) - trail);  // (byte arithmetic, L84)
    out.ins = ins;
    out.del = del;
    out.nmis = ins + del + sub;
    out.ne = __fsub_rn((float)nx, __fmul_rn(0.9f, (float)lead));  // Match.countErrorsInNeedleman: two roundings
    out.term6 = false;
    out.end5 = out.endn = 0.0f;
    out.consec = out.best_two = 0;
    if (kEnds) {
        out.term6 = (X & (MW)0x3F) == 0 && t_end >= 6;
        // cb of a step = read bases consumed before it = non-left steps before it: "cb <= 1" are the steps up to the 2nd non-left one
        const MW ND = T0 | T1;
        const MW m12 = nw_lowmask<MW>(nw_kth_bit(ND, 2) + 1);
        const MW m5 = nw_lowmask<MW>(nw_kth_bit(ND, 5) + 1);
        const MW mn = n_end > 0 ? nw_lowmask<MW>(nw_kth_bit(ND, n_end) + 1) : (MW)0;
        out.end5 = nw_end_errors(nw_popc((MW)(X & m12 & m5)), nw_popc((MW)(X & m5 & ~m12)));
        out.endn = nw_end_errors(nw_popc((MW)(X & m12 & mn)), nw_popc((MW)(X & mn & ~m12)));
    }
    if (kRuns) {
        // runs of matches in walk order; a run counts iff an 'x' was met before it, i.e. unless it starts the walk
        MW m = Mt & (Mt + (MW)1);
        int consec = 0, s1 = 0, s2 = 0, n_runs = 0;
        while (m) {
            const MW lowbit = m & (~m + (MW)1);
            const MW nxt = m + lowbit;  // the carry runs through the lowest run
            const int run = nw_ctz(nxt) - nw_ctz(m);
            m &= nxt;
            consec = max(consec, run);
            if (run > 4) {  // getSumOfBestTwoMatchStretchesNeedleman L183-196 as it is written
                if (n_runs == 0 || run < s1) {
                    s2 = s1;
                    s1 = run;
                } else if (n_runs == 1 || run < s2) {
                    s2 = run;
                }
                n_runs++;
            }
        }
        out.consec = consec;
        out.best_two = (n_runs >= 1 ? s1 : 0) + (n_runs >= 2 ? s2 : 0);
    }
}

template <int N, bool kEnds = true, bool kRuns = true, int W = N>
__device__ __forceinline__ void nw_full(const uint32_t (&col)[N], int n_end, AlnStats &out) {
    constexpr int NLO = N < 16 ? N : 16, NHI = N - NLO;
    static_assert(N <= 32, "two 32-bit move words per row");
    uint32_t mlo[N], mhi[NHI > 0 ? N : 1];
    {
        int U[N + 1];
#pragma unroll
        for (int c = 0; c <= N; c++) U[c] = -20 * c + 2;  // row 0 (columns beyond the band are never read)
#pragma unroll
        for (int r = 1; r <= N; r++) {
            const int clo = r - W > 1 ? r - W : 1, chi = r + W < N ? r + W : N;
            int diag = U[clo - 1];
            if (r <= W) U[0] = 4 * r + 2;  // 4 * (-4r) + 2 + 20r
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int c = clo; c <= chi; c++) {
                int m, d;  // d = diag + 41 * match bit (asm: the compiler's own choice is and/cmp/cndmask/add)
                asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(m) : "v"(col[c - 1]), "n"(r - 1));
                asm("v_mad_u32_u24 %0, %1, 41, %2" : "=v"(d) : "v"(m), "v"(diag));
                int v;
                if (c - r == W) {  // the cell above lies outside the band
                    v = max(d, U[c - 1] - 22);
                } else if (r - c == W) {  // the cell to the left lies outside the band
                    v = max(d, U[c] - 1);
                } else {
                    const int up = U[c] - 1, left = U[c - 1] - 22;
                    v = max(max(d, up), left);
                }
                diag = U[c];
                U[c] = (v & ~3) | 2;
                // {v, word} >> 2: the tag enters at the top.  Inline asm: as an intrinsic the chain is re-associated
                // into 16 masks/shifts/ors per row and every v stays live until then
                if (c <= 16)
                    asm("v_alignbit_b32 %0, %1, %2, 2" : "=v"(lo) : "v"(v), "v"(lo));
                else
                    asm("v_alignbit_b32 %0, %1, %2, 2" : "=v"(hi) : "v"(v), "v"(hi));
            }
            // the tag of column c belongs at bits 2(c-1) of lo / 2(c-17) of hi; the last one pushed sits at the top
            const int cl = chi < 16 ? chi : 16;
            mlo[r - 1] = cl < 16 ? lo >> (32 - 2 * cl) : lo;
            if (NHI > 0) mhi[r - 1] = chi > 16 ? hi >> (32 - 2 * (chi > 16 ? chi - 16 : 1)) : 0u;
            __builtin_amdgcn_sched_barrier(0);  // keep rows apart: interleaving them only costs registers
        }
    }
#ifdef SMI_NW_WALK_LOOP
    int c = N, lead = 0;
    int ins = 0, del = 0, sub = 0, trail = 0, cb = 0, t = 0, nx = 0;
    bool trailing = true, term = true;
    float e5 = 0.0f, en = 0.0f;
    // runs of '.', met in reverse: a run is closed (counts) iff an 'x' was met before it on the way back
    int run = 0, consec = 0, s1 = 0, s2 = 0, n_runs = 0;
    bool seen_x = false, run_closed = false;
    auto close_run = [&]() {
        if (kRuns && run > 0 && run_closed) {
            consec = max(consec, run);
            if (run > 4) {
                if (n_runs == 0 || run < s1) {
                    s2 = s1;
                    s1 = run;
                } else if (n_runs == 1 || run < s2) {
                    s2 = run;
                }
                n_runs++;
            }
        }
        run = 0;
    };
#pragma unroll
    for (int R = N; R >= 1; R--) {
        // every row above contributed one column and at least one step: cb = N - R and t >= N - R while row R is walked, so
        // the "first 5 columns" / "first 6 steps" bookkeeping is dead code in the lower rows (R is a constant after unrolling)
        const bool first5 = N - R < 5, first6 = N - R < 6;
        if (c > 0) {
            int tag;
            do {
                tag = (int)(((NHI > 0 && c > 16) ? (mhi[R - 1] >> (2 * (c - 17))) : (mlo[R - 1] >> (2 * (c - 1)))) & 3u);
                const bool x = tag != 3;
                const bool read_gap = tag == 0;
                ins += tag == 1;
                del += read_gap;
                sub += tag == 2;
                nx += x;
                if (trailing && read_gap)
                    trail++;
                else
                    trailing = false;
                if (kEnds && first6 && t < 6 && x) term = false;
                if (x) {
                    close_run();
                    seen_x = true;
                    if (kEnds && first5 && cb < 5) e5 = cb <= 1 ? (float)((double)e5 + 1.2) : __fadd_rn(e5, 1.0f);
                    if (kEnds && cb < n_end) en = cb <= 1 ? (float)((double)en + 1.2) : __fadd_rn(en, 1.0f);
                } else {
                    if (kRuns) {
                        if (run == 0) run_closed = seen_x;
                        run++;
                    }
                }
                if (!read_gap) cb++;
                if (tag != 1) c--;
                t++;
            } while (tag == 0 && c > 0);
            if (c == 0) lead = tag == 0 ? R : R - 1;  // rows still above the path when it reaches the first column
        }
    }
    // the rest of the path runs along the first column (`lead` up-moves = leading template gaps) or the first row
    // (c left-moves = read gaps); every such column is an 'x'
    int r = lead;
    if (r > 0 || c > 0) {
        close_run();
        seen_x = true;
    }
    for (; r > 0; r--) {  // up moves
        ins++;
        nx++;
        if (kEnds && t < 6) term = false;
        if (kEnds && cb < 5) e5 = cb <= 1 ? (float)((double)e5 + 1.2) : __fadd_rn(e5, 1.0f);
        if (kEnds && cb < n_end) en = cb <= 1 ? (float)((double)en + 1.2) : __fadd_rn(en, 1.0f);
        trailing = false;
        cb++;
        t++;
    }
    for (; c > 0; c--) {  // left moves (read gaps)
        del++;
        nx++;
        if (trailing) trail++;
        if (kEnds && t < 6) term = false;
        if (kEnds && cb < 5) e5 = cb <= 1 ? (float)((double)e5 + 1.2) : __fadd_rn(e5, 1.0f);
        if (kEnds && cb < n_end) en = cb <= 1 ? (float)((double)en + 1.2) : __fadd_rn(en, 1.0f);
        t++;
    }
    close_run();
    del = (int)(int8_t)(del - trail);  // trailing read gaps are not deletions (byte arithmetic, L84)
    out.ins = ins;
    out.del = del;
    out.nmis = ins + del + sub;
    out.term6 = term && t >= 6;
    out.end5 = e5;
    out.endn = en;
    out.consec = consec;
    out.best_two = (n_runs >= 1 ? s1 : 0) + (n_runs >= 2 ? s2 : 0);
    // Match.countErrorsInNeedleman: (float)#x - 0.9f * (float)lead, two roundings
    out.ne = __fsub_rn((float)nx, __fmul_rn(0.9f, (float)lead));
#else
    nw_walk_bits<N, kEnds, kRuns>(mlo, mhi, n_end, out);
#endif
}

// Error count only (Match.countErrorsInNeedleman = #x - 0.9f * leading template gaps), no moves kept: the two
// statistics ride in the low bits of the cell, below the score and the move tag, where they cannot influence the
// max (the three candidates of a cell always differ in (score, tag)).
//   cell = (4*score + 2 + 20*r) << 11 | q << 5 | lead      q = 32 + #up - #match on the best path
// On an N x N alignment #x = N + #up - #match (#up = #left, #diag = N - #up), so one counter is enough; 6 VALU ops
// per cell.
template <int N, int W = N>
__device__ __forceinline__ float nw_errors(const uint32_t (&col)[N]) {
    static_assert(N <= 27, "field widths: lead 5 bits, q 6 bits");
    constexpr int SH = 11;
    const uint32_t kmatch = (41u << SH) - 32u;  // diag with a match: +41 in the tagged score, q - 1
    int U[N + 1];
#pragma unroll
    for (int c = 0; c <= N; c++) U[c] = ((-20 * c + 2) << SH) + (32 << 5);
#pragma unroll
    for (int r = 1; r <= N; r++) {
        const int clo = r - W > 1 ? r - W : 1, chi = r + W < N ? r + W : N;  // the band of nw_full
        int diag = U[clo - 1];
        if (r <= W) U[0] = ((4 * r + 2) << SH) + ((32 + r) << 5) + r;
#pragma unroll
        for (int c = clo; c <= chi; c++) {
            int m, d;
            asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(m) : "v"(col[c - 1]), "n"(r - 1));
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(m), "s"(kmatch), "v"(diag));
            int v;
            if (c - r == W) {
                v = max(d, U[c - 1] - (22 << SH));
            } else if (r - c == W) {
                v = max(d, U[c] - (1 << SH) + 32);
            } else {
                const int up = U[c] - (1 << SH) + 32, left = U[c - 1] - (22 << SH);
                v = max(max(d, up), left);
            }
            diag = U[c];
            U[c] = (v & ~(3 << SH)) | (2 << SH);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const int lead = U[N] & 31, nx = N + ((U[N] >> 5) & 63) - 32;
    return __fsub_rn((float)nx, __fmul_rn(0.9f, (float)lead));
}

// ---- Myers / Hyyro bit-vector step shared by the exact pre-filters of K-CHIM-A and K-SCAN (derivation: smi_chimera.hip "Exact
// pre-filter of the Needleman-Wunsch acceptance tests").  Sixteen two-cycle VALU operations per pattern base in one asm block.
#define SMI_MYERS_STEP(EQ)                                                                                          \
    asm volatile("v_or_b32 %2, %5, %1\n\t"  /* Xv = Eq | Mv                  */                                      \
                 "v_and_b32 %3, %5, %0\n\t" /* t  = Eq & Pv                  */                                      \
                 "v_add_u32 %3, %3, %0\n\t" /* t += Pv                       */                                      \
                 "v_xor_b32 %3, %3, %0\n\t" /* t ^= Pv                       */                                      \
                 "v_or_b32 %3, %3, %5\n\t"  /* Xh = t | Eq                   */                                      \
                 "v_or_b32 %4, %3, %0\n\t"  /* u  = Xh | Pv                  */                                      \
                 "v_not_b32 %4, %4\n\t"     /* u  = ~u                       */                                      \
                 "v_or_b32 %4, %1, %4\n\t"  /* Ph = Mv | u                   */                                      \
                 "v_and_b32 %3, %0, %3\n\t" /* Mh = Pv & Xh                  */                                      \
                 "v_add_u32 %4, %4, %4\n\t" /* Ph <<= 1                      */                                      \
                 "v_or_b32 %4, 1, %4\n\t"   /* Ph |= 1  (D[0][j] = j)        */                                      \
                 "v_add_u32 %3, %3, %3\n\t" /* Mh <<= 1                      */                                      \
                 "v_or_b32 %0, %2, %4\n\t"  /* w  = Xv | Ph                  */                                      \
                 "v_not_b32 %0, %0\n\t"     /* w  = ~w                       */                                      \
                 "v_or_b32 %0, %3, %0\n\t"  /* Pv = Mh | w                   */                                      \
                 "v_and_b32 %1, %4, %2\n\t" /* Mv = Ph & Xv                  */                                      \
                 : "+v"(pv), "+v"(mv), "=&v"(t_xv), "=&v"(t_a), "=&v"(t_b)                                           \
                 : "v"(EQ))


// The bound for the 16-mer TSO of K-SCAN ("AACGCAGAGTACATGG", Jar/config.xml:155), pattern consumed from its end, planes as compile-time
// constants: V[b] = plane b (A G C T) of the 16-base read slice bit-reversed into the low 16 bits.
//   -> min over s <= lead_max of Levenshtein(TSO, slice[s ..]), a lower bound of Match.countErrorsInNeedleman whenever that is < lead_max + 0.5
__device__ __forceinline__ int myers_bound_tso16(const uint32_t (&V)[4], int lead_max) {
    constexpr int M = 16;
    // planes of "AACGCAGAGTACATGG" read backwards: G G T A C A T G A G A C G C A A
    constexpr int PL[M] = {1, 1, 3, 0, 2, 0, 3, 1, 0, 1, 0, 2, 1, 2, 0, 0};
    uint32_t pv = 0xFFFFFFFFu, mv = 0u, t_xv, t_a, t_b;
#pragma unroll
    for (int j = 0; j < M; j++) {
        const uint32_t eqv = V[PL[j]];
        SMI_MYERS_STEP(eqv);
    }
    const uint32_t mask = (1u << M) - 1u;
    int d = M + __popc(pv & mask) - __popc(mv & mask);  // D[M][M]
    int best = d;
    for (int k = M - 1; k >= M - lead_max && k >= 0; k--) {
        d -= (int)((pv >> k) & 1u) - (int)((mv >> k) & 1u);
        best = min(best, d);
    }
    return best;
}

}  // namespace smi
