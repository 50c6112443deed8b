// smi_umi.hip -- K-UMI: pairwise UMI distances of assignumis on gfx950 (hand-written HIP).
//
// Reference: ClusteringEditDistanceBase.generateDistanceMatrix / calcEditDistances / calcBestEditDistance
// (FJ!clustering/ClusteringEditDistanceBase.java:L168-259, L297-350, L67-80) with the bounded Levenshtein of
// FJ!nanopore/analyzers/apachemod/LevenshteinDistance.java:L220-283 (threshold 4, "-1" stored as 5).
// For every pair of reads of a (cell barcode, genomic region) group: nine distances between the 12-mers that start
// -1/0/+1 bases after the barcode end of each read, the first strict minimum in the order ZERO, PLUSONE, MINUSONE
// (enum ordinal order, PlusMinusOnePosData.java:L20-22) and the two winning offsets.
//
// MI355X mapping: one lane per unordered pair (i <= v) of a group, pairs of all groups flattened so that a launch
// fills the chip whatever the group sizes are.  A 14-base window is one 64-bit register (4-bit codes); each
// distance is Myers' bit-parallel global edit distance on 12-bit vectors -- the banded DP of the reference returns
// exactly min(Levenshtein, "> 4"), which is what the bit-vector algorithm's last-row score gives.  Integer/bitwise
// only: no MFMA, no LDS.
#include <hipcub/hipcub.hpp>
#include <type_traits>

#include "smi_internal.h"
#include "smi_umi_stage.h"

namespace smi {

// The window's four code bits as 14-bit planes: plane c, bit k = bit c of nibble k.  Every fourth bit of a dword is drawn together in three
// or-shift steps (8 nibbles of the low dword, 6 of the high one); the masks "nibble == A / G / C / T / N" are then a dozen logic operations on
// the planes.  (The first version tested each code with a SWAR compare and moved its 14 flag bits one by one: 350 operations per pair of
// 2,200.)
__device__ __forceinline__ uint32_t every_fourth_bit(uint32_t t) {  // bits 0, 4, .., 28 -> bits 0 .. 7
    t = (t | (t >> 3)) & 0x03030303u;
    t = (t | (t >> 6)) & 0x000F000Fu;
    return (t | (t >> 12)) & 0xFFu;
}
struct EqMasks {
    uint32_t a, g, c, t, n;
};
__device__ __forceinline__ EqMasks eq_masks(uint64_t w) {
    const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
    uint32_t p[4];
#pragma unroll
    for (int c = 0; c < 4; c++) p[c] = every_fourth_bit((lo >> c) & 0x11111111u) | (every_fourth_bit((hi >> c) & 0x00111111u) << 8);
    EqMasks m;
    m.a = p[0] & ~(p[1] | p[2] | p[3]);  // code 1
    m.g = p[1] & ~(p[0] | p[2] | p[3]);  // code 2
    m.c = p[2] & ~(p[0] | p[1] | p[3]);  // code 4
    m.t = p[3] & ~(p[0] | p[1] | p[2]);  // code 8
    m.n = p[0] & p[1] & p[2] & p[3];     // code 15
    return m;
}

// ---- the nine distances of a pair (round 5) ----------------------------------------------------------------------------------------------
// Round 4's form ran nine separate 12-bit Myers recurrences, 19 compiler-chosen instructions per step, nearly all three-operand forms
// (v_bitop3, v_and_or, v_lshl_or: four issue cycles each, profiles/r02/valu_peak.json): 2,090 instructions = ~ 8,400 issue cycles per pair.
// Three changes, none of which alters a result:
//   * TWO recurrences per register: 12-bit fields at bits 0 .. 11 and 16 .. 27.  The only operation that crosses bit positions upwards is
//     the addition (Eq & Pv) + Pv; its carry out of a field stops at bit 12 / 28 as long as Pv is zero there, so one extra AND per step
//     (Pv &= 0x0FFF0FFF) keeps the fields apart (Eq and Mv may hold anything in the guard bits).  The "+1 per text base" of the boundary
//     row is OR 0x00010001 (and that also overwrites whatever the shift moved from the low field's guard bits into bit 16).  Nine distances = five runs: (i = 0, 1) x v for v = 0, 1, 2;
//     (i = 2) x (v = 0, 1); (i = 2, v = 2).
//   * the distance is read off the last column at the end, D[12][12] = 12 + popcount(Pv) - popcount(Mv), instead of being followed step
//     by step (two extractions and two adds per step and field).
//   * a step is one asm block of seventeen two-operand instructions (and / or / xor / add / not on VGPRs and literals): those issue in two
//     cycles instead of four as long as they come in long runs.  Five runs of 12 x 17 instructions = 2,040 issue cycles; the ~ 450
//     instructions around them (masks, look-ups, scores, the tile's shuffles) are four-cycle forms: 1,800 cycles.  Measured: 4,340 cycles
//     per wave of pairs, 89 % of that issue bound.
// The match masks: eq[j] = positions of read a equal to base j of read b is a 5-way select per text base -- ~ 100 compare / select
// instructions per pair in round 4.  Now every lane keeps a 16-entry table
// indexed by the 4-bit code in LDS ([code][thread]: conflict-free, no other lane ever touches the column, so no barrier): the five live
// entries are written per pair already in the packed form the runs consume, codes that match nothing stay zero from the kernel's start
// (the flat kernel; the tiled kernel shares one table per tile row: UmiRowTable).
#ifndef SMI_UMI_TILE_THREADS
#define SMI_UMI_TILE_THREADS 256
#endif
constexpr int kUmiEqCols = SMI_UMI_TILE_THREADS > 256 ? SMI_UMI_TILE_THREADS : 256;  // a column per thread of the larger workgroup
struct UmiEqTable {
    // entry = m | m << 15 for the 14-bit mask m: bits 0 .. 11 = m[0 .. 11] (pattern offset 0), bits 16 .. 27 = m[1 .. 12] (offset 1); what
    // lands in the guard bits 12 .. 15 / 28 .. 31 is harmless in an Eq word (it never reaches Pv, and the scores mask it off Mv);
    // entry >> 2 has m[2 .. 13] (offset 2) in its low field
    uint32_t e[16][kUmiEqCols];
};

#define SMI_M2_STEP(E)                                                                    \
    "v_or_b32 %2, " E ", %1\n\t"       /* Xv = Eq | Mv                                  */ \
    "v_and_b32 %3, " E ", %0\n\t"      /* t  = Eq & Pv                                  */ \
    "v_add_u32 %3, %3, %0\n\t"         /* t += Pv       (carry stops at bit 12 / 28)    */ \
    "v_xor_b32 %3, %3, %0\n\t"         /* t ^= Pv                                       */ \
    "v_or_b32 %3, %3, " E "\n\t"       /* Xh = t | Eq                                   */ \
    "v_or_b32 %4, %3, %0\n\t"          /* u  = Xh | Pv                                  */ \
    "v_not_b32 %4, %4\n\t"             /* u  = ~u                                       */ \
    "v_or_b32 %4, %1, %4\n\t"          /* Ph = Mv | u                                   */ \
    "v_and_b32 %3, %0, %3\n\t"         /* Mh = Pv & Xh  (clean: Pv is)                  */ \
    "v_add_u32 %4, %4, %4\n\t"         /* Ph <<= 1                                      */ \
    "v_or_b32 %4, 0x10001, %4\n\t"     /* Ph |= 1 in both fields (D[0][j] = j)          */ \
    "v_add_u32 %3, %3, %3\n\t"         /* Mh <<= 1                                      */ \
    "v_or_b32 %0, %2, %4\n\t"          /* w  = Xv | Ph                                  */ \
    "v_not_b32 %0, %0\n\t"             /* w  = ~w                                       */ \
    "v_or_b32 %0, %3, %0\n\t"          /* Pv = Mh | w                                   */ \
    "v_and_b32 %0, 0xfff0fff, %0\n\t"  /* the guard bits of Pv stay zero                */ \
    "v_and_b32 %1, %4, %2\n\t"         /* Mv = Ph & Xv  (clean: Xv is)                  */
// A statement per step (the default) lets the compiler put the waits for the table look-ups and a three-operand instruction of the next
// run's masks between the steps; twelve steps in ONE statement keep the run pure and were SLOWER (27.5 against 25.9 ms: every look-up must
// have arrived before the first step, and what the compiler would have slipped in runs afterwards in a block of its own) -- an odd
// instruction per seventeen does not cost the two-cycle rate, a block of them does (NOTES R5.1).
#define SMI_MYERS2_ONE(X) asm volatile(SMI_M2_STEP("%5") : "+v"(pv), "+v"(mv), "=&v"(t_xv), "=&v"(t_a), "=&v"(t_b) : "v"(X))
// one run = UL steps (UL = umi_length, config.xml:264: 12 as shipped, 10 for the 10x v2 chemistry; 8 .. 12 built), fully unrolled so that E[O + k] are registers.
// (Round 5 also measured the twelve steps as ONE asm statement: 27.5 against 25.9 ms per 0.94 G pairs, NOTES R5.1 -- not kept.)
#define SMI_MYERS2_RUN(E, O)                                    \
    do {                                                        \
        _Pragma("unroll") for (int k_ = 0; k_ < UL; k_++) SMI_MYERS2_ONE(E[(O) + k_]); \
    } while (0)

// distance << 12 of both fields, ready to take the enumeration rank and the offsets in the low bits (the clamp to 5 comes once, at the end)
// (a pattern of UL < 12 bases lives in the low UL bits of its field: nothing in the recurrence moves information downwards, so the rows above
// the pattern's last -- whatever the table holds there -- change no bit below them, and the last column is read off rows 0 .. UL - 1 only)
template <int UL>
__device__ __forceinline__ void myers2_scores(uint32_t pv, uint32_t mv, uint32_t &lo, uint32_t &hi) {
    constexpr uint32_t M = (1u << UL) - 1u;
    lo = (uint32_t)(__popc(pv & M) + UL - __popc(mv & M)) << 12;
    hi = (uint32_t)(__popc((pv >> 16) & M) + UL - __popc((mv >> 16) & M)) << 12;  // (Pv's guard bits are zero, Mv's are not)
}

// calcBestEditDistance L67-80 visits (i, v) in the order (1,2,0) x (1,2,0) and keeps the first strict minimum: the least of
// distance << 12 | rank << 8 | i << 4 | v << 6
__host__ __device__ constexpr uint32_t umi_rank_code(int i, int v) {
    const int ri = i == 1 ? 0 : i == 2 ? 1 : 2, rv = v == 1 ? 0 : v == 2 ? 1 : 2;
    return (uint32_t)((3 * ri + rv) << 8) | (uint32_t)(i << 4) | (uint32_t)(v << 6);
}

// the five runs on the UL + 2 looked-up words of a pair: w[j] = table entry of read a for the code of base j of read b
template <int UL>
__device__ __forceinline__ uint32_t umi_pair_runs(const uint32_t (&w)[UL + 2]) {
    constexpr uint32_t kInit = ((1u << UL) - 1u) * 0x00010001u;  // Pv = D[i][0] - D[i-1][0] = +1 over the pattern's UL rows, both fields
    uint32_t w2[UL + 2];
#pragma unroll
    for (int j = 0; j < UL + 2; j++) w2[j] = w[j] >> 2;
    uint32_t g[UL];  // run 3: pattern offset 2 against text offsets 0 (low field) and 1 (high field): the low halves of w2[t] and w2[t + 1]
#pragma unroll
    for (int t = 0; t < UL; t++) g[t] = __builtin_amdgcn_perm(w2[t + 1], w2[t], 0x05040100u);
    uint32_t key = 0xFFFFFFFFu, lo, hi, t_xv, t_a, t_b;
#pragma unroll
    for (int v = 0; v < 3; v++) {  // pattern offsets 0 / 1 against text offset v
        uint32_t pv = kInit, mv = 0u;
        SMI_MYERS2_RUN(w, v);
        myers2_scores<UL>(pv, mv, lo, hi);
        key = min(key, min(lo | umi_rank_code(0, v), hi | umi_rank_code(1, v)));
    }
    {
        uint32_t pv = kInit, mv = 0u;
        SMI_MYERS2_RUN(g, 0);
        myers2_scores<UL>(pv, mv, lo, hi);
        key = min(key, min(lo | umi_rank_code(2, 0), hi | umi_rank_code(2, 1)));
    }
    {
        uint32_t pv = kInit, mv = 0u;
        SMI_MYERS2_RUN(w2, 2);  // (the high field holds leftovers: its result is not read)
        myers2_scores<UL>(pv, mv, lo, hi);
        key = min(key, lo | umi_rank_code(2, 2));
    }
    // limitedCompare: -1 above the threshold 4, stored as 5 (L343).  Distances of 5 and more are all 5 to the strict-minimum scan, so when
    // the least is one of them the first pair visited keeps its place: (i, v) = (1, 1)
    key = min(key, (5u << 12) | umi_rank_code(1, 1));
    return (key >> 12) | (key & 0xF0u);
}

__device__ __forceinline__ uint32_t umi_word(uint32_t m) { return m | (m << 15); }  // a table entry from a 14-bit mask (UmiEqTable)

// the flat kernel's pair: every lane writes the five live entries of its own column and looks its words up there
template <int UL>
__device__ __forceinline__ uint32_t umi_pair(uint64_t a, uint64_t b, UmiEqTable &T, int tid) {
    const EqMasks ma = eq_masks(a);
    T.e[1][tid] = umi_word(ma.a);
    T.e[2][tid] = umi_word(ma.g);
    T.e[4][tid] = umi_word(ma.c);
    T.e[8][tid] = umi_word(ma.t);
    T.e[15][tid] = umi_word(ma.n);
    uint32_t w[UL + 2];
#pragma unroll
    for (int j = 0; j < UL + 2; j++) w[j] = T.e[(uint32_t)(b >> (4 * j)) & 15u][tid];
    return umi_pair_runs<UL>(w);
}

// the tiled kernel's: the 64 rows of a tile share their read a with the 64 columns of that row, so the entries of a row are written ONCE per
// tile row (by 64 threads, before the tile's pairs start) into a [row][code] table -- the masks of read a were 75 of the ~ 450 instructions
// around the runs when every pair recomputed them.  Rows are 17 words apart: the eight rows a wave reads at once start in different banks.
constexpr int kUmiTileMin = 64;   // groups above this many reads are tiled
constexpr int kUmiTile = 64;
struct UmiRowTable {
    uint32_t e[kUmiTile][17];
};
__device__ __forceinline__ void umi_row_fill(UmiRowTable &R, int row, uint64_t a) {
    const EqMasks ma = eq_masks(a);
#pragma unroll
    for (int c = 0; c < 16; c++) R.e[row][c] = 0u;  // codes that match nothing
    R.e[row][1] = umi_word(ma.a);
    R.e[row][2] = umi_word(ma.g);
    R.e[row][4] = umi_word(ma.c);
    R.e[row][8] = umi_word(ma.t);
    R.e[row][15] = umi_word(ma.n);
}
template <int UL>
__device__ __forceinline__ uint32_t umi_pair_row(const UmiRowTable &R, int row, uint64_t b) {
    uint32_t w[UL + 2];
#pragma unroll
    for (int j = 0; j < UL + 2; j++) w[j] = R.e[row][(uint32_t)(b >> (4 * j)) & 15u];
    return umi_pair_runs<UL>(w);
}

// every lane's column of the table: the codes that match nothing (everything but A, G, C, T, N) are zero and stay zero
__device__ __forceinline__ void umi_table_init(UmiEqTable &T, int tid) {
#pragma unroll
    for (int c = 0; c < 16; c++) T.e[c][tid] = 0u;
}

// ---- two mappings of pairs to lanes ---------------------------------------------------------------------------------------------------
// A pair (i, v) is stored twice: m[i][v] and, with its offsets swapped, m[v][i].  With one lane per pair in row-major order (the flat kernel)
// the first store of a wave is 64 consecutive bytes, the second 64 bytes a row apart: every one of them dirties a 128-byte line of L2 by one
// byte, the rest of that line is written by other workgroups -- on other XCDs, with their own L2 -- much later, and the line goes out to HBM
// many times: 5.8 x the matrix bytes (profiles/r03/umi_pmc.json).  Groups above 64 reads (which hold nearly all matrix bytes: they grow
// with n^2) are therefore cut into 64 x 64 tiles, a workgroup per tile, a wave per 8 x 8 block of it: the eight results of a row of the block
// travel to one lane (three shuffles) and leave as one 8-byte store, and so do the eight of a column for the mirrored copy.  Both copies of
// the tile's 64-byte row pieces are complete when the workgroup ends, in ONE L2.  Blocks that the diagonal or the matrix edge cuts store
// byte by byte, lanes below the diagonal idle there (the reference computes i <= v only, and its tie-break is not symmetric).  Small groups
// keep the flat mapping: a tile over a 3-read group would be 98 % idle lanes.

struct UmiPlan {
    uint64_t small_pairs, tiles;
};
struct UmiPlanAdd {
    __host__ __device__ UmiPlan operator()(const UmiPlan &a, const UmiPlan &b) const { return UmiPlan{a.small_pairs + b.small_pairs, a.tiles + b.tiles}; }
};

// plan[g] = what group g contributes to the two kernels (plan[n_groups] = nothing: the exclusive scan leaves the totals there)
__global__ void k_umi_plan(const uint32_t *__restrict__ group_off, uint32_t n_groups, UmiPlan *__restrict__ plan) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_groups) return;
    UmiPlan p{0, 0};
    if (g < n_groups) {
        const uint64_t n = group_off[g + 1] - group_off[g];
        if (n > (uint64_t)kUmiTileMin) {
            const uint64_t nb = (n + 4 * kUmiTile - 1) / (4 * kUmiTile);  // macro-tiles of 4 x 4 tiles (k_umi_dist_tiles)
            p.tiles = nb * (nb + 1) / 2;
        } else
            p.small_pairs = n * (n + 1) / 2;
    }
    plan[g] = p;
}

// row i of an upper triangle of side n (rows have n, n-1, ... entries) that holds entry `local`: first(i) = i*n - i*(i-1)/2
__device__ __forceinline__ uint64_t tri_row(uint64_t n, uint64_t local) {
    uint64_t i = (uint64_t)(((double)(2 * n + 1) - sqrt((double)(2 * n + 1) * (double)(2 * n + 1) - 8.0 * (double)local)) * 0.5);
    if (i >= n) i = n - 1;
    while (i > 0 && i * n - i * (i - 1) / 2 > local) i--;
    while ((i + 1) * n - (i + 1) * i / 2 <= local) i++;
    return i;
}

// the flat kernel: one lane per pair (i <= v) of the groups of at most kUmiTileMin reads, pairs of all those groups in one index space
// (plan[g].small_pairs = pairs in front of group g; a tiled group has none, so "the last g with plan[g].small_pairs <= t" never lands on one)
template <int UL>
__global__ __launch_bounds__(256) void k_umi_dist(const uint64_t *__restrict__ windows, const uint32_t *__restrict__ group_off,
                                                  const UmiPlan *__restrict__ plan, const uint64_t *__restrict__ mat_off,
                                                  uint32_t n_groups, uint8_t *__restrict__ out) {
    __shared__ UmiEqTable T;
    umi_table_init(T, threadIdx.x);
    const uint64_t total_pairs = plan[n_groups].small_pairs;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < total_pairs; t += (uint64_t)gridDim.x * blockDim.x) {
        // group of this pair: last g with off[g] <= t.  The pairs of a wave are consecutive, so the binary search
        // runs once for the wave's first pair (scalar) and every lane walks forward from there
        const uint64_t t_first = __builtin_amdgcn_readfirstlane((uint32_t)(t >> 32)) * 0x100000000ull +
                                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)t);
        uint32_t lo = 0, hi = n_groups;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (plan[mid].small_pairs <= t_first)
                lo = mid;
            else
                hi = mid;
        }
        uint32_t g = lo;
        while (g + 1 < n_groups && plan[g + 1].small_pairs <= t) g++;
        const uint64_t local = t - plan[g].small_pairs;
        const uint32_t r0 = group_off[g];
        const uint64_t n = group_off[g + 1] - r0;
        const uint64_t i = tri_row(n, local);
        const uint64_t v = i + (local - (i * n - i * (i - 1) / 2));
        const uint32_t r = umi_pair<UL>(windows[r0 + i], windows[r0 + v], T, threadIdx.x);
        uint8_t *m = out + mat_off[g];
        m[i * n + v] = (uint8_t)r;
        // transposed copy for the lower triangle (getTransposedEditDistance L133, L213-216): offsets swapped
        m[v * n + i] = (uint8_t)((r & 15u) | (((r >> 6) & 3u) << 4) | (((r >> 4) & 3u) << 6));
    }
}

// the tiled kernel: a workgroup per 256 x 256 macro-tile (mi <= mv) of a group above kUmiTileMin reads; it goes through the macro-tile's
// 4 x 4 tiles of 64 x 64 one after the other, a wave per 8 x 8 block of the tile, lane = (di, dv).  A tile's results are collected in LDS --
// A = m[i0 ..][v0 ..], B = the mirrored m[v0 ..][i0 ..] (one and the same array for a tile on the diagonal, where a cell (a, a) ends up with
// the mirrored value, as the two stores of the flat kernel leave it) -- and leave as 16-byte pieces of rows when the tile is done: every
// store instruction of a wave covers whole 64-byte row pieces.  Macro-tiles are taken from a counter: they hold between 1 and 16 tiles.
// What is left of the write amplification (profiles/r04/umi_pmc.json: 1.5 x the matrix bytes, from 5.8 x): a row of the matrix starts at
// any byte (n is not a multiple of 64), so the first and last 64-byte block of a row piece is shared with the tile beside it, which writes
// its part 160 us later (256 threads) -- L2 has written the block back by then.  Eight waves per workgroup halve that distance and merge
// more (1.37 x) but run 10 % slower (56.8 against 51.5 ms per 0.94 G pairs: fewer waves per SIMD, a barrier per 80 us); whole 256-byte row
// pieces per store would need the mirrored copy of a macro-tile in LDS (64 KB).  SMI_UMI_TILE_THREADS=512 builds the other variant.
constexpr int kUmiLdsRow = kUmiTile + 16;  // bytes per staged row (80: rows start 16-byte aligned, neighbouring rows in different banks)
constexpr int kUmiMacro = 4;               // tiles per macro-tile edge
#ifndef SMI_UMI_TILE_THREADS
#define SMI_UMI_TILE_THREADS 256
#endif
constexpr int kUmiTileThreads = SMI_UMI_TILE_THREADS;

template <int UL>
__global__ __launch_bounds__(kUmiTileThreads) void k_umi_dist_tiles(const uint64_t *__restrict__ windows, const uint32_t *__restrict__ group_off,
                                                                    const UmiPlan *__restrict__ plan, const uint64_t *__restrict__ mat_off,
                                                                    uint32_t n_groups, uint32_t *__restrict__ next_unit, uint8_t *__restrict__ out, int padded) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[2][kUmiTile][kUmiLdsRow];
    __shared__ uint32_t s_unit;
    __shared__ UmiRowTable R;
    const uint64_t total_units = plan[n_groups].tiles;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int di = lane >> 3, dv = lane & 7;
    for (;;) {
        if (threadIdx.x == 0) s_unit = atomicAdd(next_unit, 1u);
        __syncthreads();
        const uint64_t unit = s_unit;
        __syncthreads();  // (s_unit is written again at the top of the next turn)
        if (unit >= total_units) return;
        uint32_t lo = 0, hi = n_groups;  // last g with plan[g].tiles <= unit (uniform)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (plan[mid].tiles <= unit)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t g = lo;
        const uint64_t local = unit - plan[g].tiles;
        const uint32_t r0 = group_off[g];
        const uint64_t n = group_off[g + 1] - r0;
        const uint64_t nbm = (n + kUmiMacro * kUmiTile - 1) / (kUmiMacro * kUmiTile);
        const uint64_t mi = tri_row(nbm, local), mv = mi + (local - (mi * nbm - mi * (mi - 1) / 2));
        uint8_t *m = out + mat_off[g];
        const uint64_t ld = umi_ld(n, padded != 0);  // row stride: n, or n rounded up to whole 64-byte lines (smi_umi_stage.h)
        const uint64_t *win = windows + r0;
        uint64_t rows_of = ~0ull;  // first row of the tile row the table holds
        for (int tt = 0; tt < kUmiMacro * kUmiMacro; tt++) {
            const uint64_t bi = kUmiMacro * mi + tt / kUmiMacro, bv = kUmiMacro * mv + tt % kUmiMacro;
            const uint64_t i0 = bi * kUmiTile, v0 = bv * kUmiTile;
            if (i0 >= n || v0 >= n || bi > bv) continue;  // (uniform)
            uint8_t(*A)[kUmiLdsRow] = stage[0];
            uint8_t(*B)[kUmiLdsRow] = stage[bi == bv ? 0 : 1];
            if (i0 != rows_of) {  // (uniform) the four tiles of a macro-tile row share their reads a: their table is built once
                // nobody reads the table any more: every wave has passed the barrier behind the previous tile's pairs
                if (threadIdx.x < kUmiTile && i0 + threadIdx.x < n) umi_row_fill(R, (int)threadIdx.x, win[i0 + threadIdx.x]);
                rows_of = i0;
                __syncthreads();
            }
            int live = 0;  // blocks of this tile that hold pairs, dealt to the eight waves in turn
            for (int sb = 0; sb < 64; sb++) {
                const int si = sb >> 3, sv = sb & 7;
                const uint64_t bi0 = i0 + 8 * si, bv0 = v0 + 8 * sv;  // the block's first row / column
                if (bi0 >= n || bv0 >= n || bv0 + 7 < bi0) continue;  // outside the matrix, or wholly below the diagonal
                if ((live++ & (kUmiTileThreads / 64 - 1)) != wv) continue;
                const uint64_t i = bi0 + di, v = bv0 + dv;
                const bool valid = i < n && v < n && i <= v;
                uint32_t r = 0;
                if (valid) r = umi_pair_row<UL>(R, 8 * si + di, win[v]);
                const uint32_t rt = (r & 15u) | (((r >> 6) & 3u) << 4) | (((r >> 4) & 3u) << 6);  // getTransposedEditDistance L133, L213-216
                const bool whole = bi0 + 7 < bv0 && bv0 + 7 < n;  // every lane holds a pair above the diagonal
                if (whole) {
                    // a row of the block -> lane dv = 0 of that row
                    uint32_t x = r | ((uint32_t)__shfl_down((int)r, 1) << 8);
                    x |= (uint32_t)__shfl_down((int)x, 2) << 16;
                    const uint32_t xh = (uint32_t)__shfl_down((int)x, 4);
                    // a column of the block (the mirrored copy's row piece) -> lane di = 0 of that column
                    uint32_t y = rt | ((uint32_t)__shfl_down((int)rt, 8) << 8);
                    y |= (uint32_t)__shfl_down((int)y, 16) << 16;
                    const uint32_t yh = (uint32_t)__shfl_down((int)y, 32);
                    if (dv == 0) *reinterpret_cast<uint2 *>(&A[8 * si + di][8 * sv]) = make_uint2(x, xh);
                    if (di == 0) *reinterpret_cast<uint2 *>(&B[8 * sv + dv][8 * si]) = make_uint2(y, yh);
                } else if (valid) {
                    A[8 * si + di][8 * sv + dv] = (uint8_t)r;
                    B[8 * sv + dv][8 * si + di] = (uint8_t)rt;  // (the same cell when i == v: the mirrored value stays)
                }
            }
            __syncthreads();
            // out: region A = rows i0 .., columns v0 ..; region B (tiles off the diagonal) = rows v0 .., columns i0 ..: threads 0 .. 255 take A,
            // the others B, thread = (row, 16-byte piece)
            for (int reg = threadIdx.x >> 8; reg < 2; reg += kUmiTileThreads / 256) {
                const int row = (threadIdx.x & 255) >> 2, piece = threadIdx.x & 3;
                const uint64_t rr = (reg ? v0 : i0) + row, c0 = (reg ? i0 : v0) + 16 * piece, c_end = min((reg ? i0 : v0) + (uint64_t)kUmiTile, n);
                if (!(reg == 1 && bi == bv) && rr < n && c0 < c_end) {
                    const uint8_t *src = &stage[reg][row][16 * piece];
                    uint8_t *dst = m + rr * ld + c0;
                    if (c0 + 16 <= c_end) {
                        uint4 q = *reinterpret_cast<const uint4 *>(src);
                        __builtin_memcpy(dst, &q, 16);
                    } else
                        for (int k = 0; k < (int)(c_end - c0); k++) dst[k] = src[k];
                }
            }
            __syncthreads();
        }
    }
}

int launch_umi_dist(smi_ctx *ctx, const uint64_t *d_windows, const uint32_t *d_group_off, const uint64_t *d_pair_off,
                    const uint64_t *d_mat_off, uint32_t n_groups, uint64_t total_pairs, uint8_t *d_out, hipStream_t s, int umi_len, bool padded) {
    (void)d_pair_off;  // (the flat index space of round 1; the two kernels take theirs from the plan below)
    if (!total_pairs || !n_groups) return SMI_OK;
    // plan: per group its pairs in the flat kernel or its tiles in the tiled one, prefix sums of both in one scan
    size_t cub_bytes = 0;
    SMI_HIP(hipcub::DeviceScan::ExclusiveScan(nullptr, cub_bytes, (UmiPlan *)nullptr, (UmiPlan *)nullptr, UmiPlanAdd(), UmiPlan{0, 0}, (int)n_groups + 1, s));
    const size_t arr = (((size_t)n_groups + 1) * sizeof(UmiPlan) + 255) & ~(size_t)255;
    const size_t need = 2 * arr + 256 + cub_bytes;
    if (ctx->umi_plan_bytes < need) {
        if (ctx->umi_plan) (void)hipFree(ctx->umi_plan);
        ctx->umi_plan = nullptr;
        ctx->umi_plan_bytes = 0;
        SMI_HIP(hipMalloc(&ctx->umi_plan, need + need / 4));
        ctx->umi_plan_bytes = need + need / 4;
    }
    char *base = static_cast<char *>(ctx->umi_plan);
    UmiPlan *d_cnt = reinterpret_cast<UmiPlan *>(base), *d_plan = reinterpret_cast<UmiPlan *>(base + arr);
    uint32_t *d_next = reinterpret_cast<uint32_t *>(base + 2 * arr);
    if (int rc = time_begin(ctx, SMI_K_UMI, s)) return rc;
    SMI_HIP(hipMemsetAsync(d_next, 0, 4, s));
    hipLaunchKernelGGL(k_umi_plan, dim3((n_groups + 1 + 255) / 256), dim3(256), 0, s, d_group_off, n_groups, d_cnt);
    SMI_HIP(hipcub::DeviceScan::ExclusiveScan(base + 2 * arr + 256, cub_bytes, d_cnt, d_plan, UmiPlanAdd(), UmiPlan{0, 0}, (int)n_groups + 1, s));
    // the grids from what the host knows: all pairs bound the flat kernel's; a tiled group has at least 2,145 pairs per macro-tile (65 reads)
    const unsigned grid = (unsigned)std::min<uint64_t>((total_pairs + 255) / 256, 256ull * 64);
    const unsigned grid_t = (unsigned)std::min<uint64_t>(total_pairs / 2145 + 1, 256ull * 2048 / kUmiTileThreads);
    auto launch = [&](auto ul) {
        constexpr int UL = decltype(ul)::value;
        hipLaunchKernelGGL(k_umi_dist<UL>, dim3(grid), dim3(256), 0, s, d_windows, d_group_off, d_plan, d_mat_off, n_groups, d_out);
        hipLaunchKernelGGL(k_umi_dist_tiles<UL>, dim3(grid_t), dim3(kUmiTileThreads), 0, s, d_windows, d_group_off, d_plan, d_mat_off, n_groups, d_next, d_out, padded ? 1 : 0);
    };
    switch (umi_len) {  // umis/umi_length (config.xml:264)
    case 12: launch(std::integral_constant<int, 12>()); break;
    case 11: launch(std::integral_constant<int, 11>()); break;
    case 10: launch(std::integral_constant<int, 10>()); break;
    case 9: launch(std::integral_constant<int, 9>()); break;
    case 8: launch(std::integral_constant<int, 8>()); break;
    default: set_error("smi_umi_dist_device: umi_length must be 8 .. 12 in this build"); return SMI_ERR_INVALID;
    }
    SMI_HIP(hipGetLastError());
    if (int rc = time_end(ctx, SMI_K_UMI, s)) return rc;
    return SMI_OK;
}

}  // namespace smi
