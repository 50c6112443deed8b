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

namespace smi {

constexpr int kEndBases = SMI_END_BASES;     // 208
constexpr int kPlaneWords = SMI_PLANE_WORDS;  // 7
constexpr int kLdsWords = 8;                  // per plane in LDS (word 7 = 0 so 64-bit fetches never run off)
constexpr int kBlock = 256;

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
    uint32_t adapter4[22];  // 4-bit codes of the adapter, padded
};

// ---- LDS plane access -----------------------------------------------------------------------------------------
// planes: [5][kLdsWords][kBlock] u32 (A, G, C, T, exact-T); the lane's column is `tid`
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

// ---- polyT finder (PolyATSearcher.java:L56-252) ---------------------------------------------------------------
// tex: exact-T plane.  Entry `pos` of the reference's score list is the T fraction of bases [pos+1, pos+15].
__device__ __forceinline__ bool find_polyt(const uint32_t *tex, int tid, const ScanParams &P, int &begin1, int &end1) {
    const int ML = P.polya_len;
    const uint32_t wmask = (1u << ML) - 1u;
    const int n = P.window + ML + 10;  // sub-sequence length (175)
    int first = -1;
    for (int pos = 0; pos < P.window; pos++) {
        const uint32_t x = get32(tex, tid, pos);
        const int cnt = __popc((x >> 1) & wmask);                       // L199-200
        if (cnt >= P.thr_first && (x & 1u) && __popc(x & 31u) > 2) {    // L217-218, lambda$2 L98-101
            first = pos;
            break;
        }
    }
    if (first < 0) return false;
    int start = first;
    const int INC[8] = {20, 15, 10, 5, 4, 3, 2, 1};  // L223-230
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int inc = INC[k];
        while (start + inc < P.window && __popc((get32(tex, tid, start + inc) >> 1) & wmask) >= P.thr_adv) start += inc;
    }
    int endpos = start + ML - 1;  // L231
    // lambda$findpolyAT$3 L122-142: walk back until T at endpos, >=2 T in the last 2, >=3 in 4, >=4 in 5
    while (endpos > 4) {
        const uint32_t x = get32(tex, tid, endpos - 4);  // bit i = base endpos-4+i
        const bool ok = ((x >> 4) & 1u) && __popc((x >> 3) & 3u) >= 2 && __popc((x >> 1) & 15u) >= 3 && __popc(x & 31u) >= 4;
        if (ok) break;
        endpos--;
    }
    while (n > endpos + 6 && __popc(get32(tex, tid, endpos + 1) & 31u) > 3) endpos += 5;  // L145-147
    while (n > endpos + 4 && __popc(get32(tex, tid, endpos + 1) & 7u) > 1) endpos += 3;   // L156-158
    while (endpos < n - 1 && (get32(tex, tid, endpos + 1) & 1u)) endpos++;                // L171-172
    begin1 = first + 1;
    end1 = endpos + 1;
    return true;
}

// ---- scan-phase Needleman-Wunsch with forward-carried statistics ----------------------------------------------
// cell = score << 16 | nx << 8 | lead  (nx = number of 'x' columns on the traceback path, lead = leading template
// gaps).  Scores (-4,-5,.,.,-5,-5,+5): NeedlemanParameters.java:L36-38.  Tie-breaks: NeedlemanWunsch.java:L55-80.
template <int AD>
__device__ __forceinline__ float nw_errors(const uint32_t (&col)[AD]) {
    // col[c] bit r = read base r of the slice matches adapter base c
    int prev[AD + 1], cur[AD + 1];
#pragma unroll
    for (int c = 0; c <= AD; c++) prev[c] = (-5 * c) * 65536 + (c << 8);  // row 0: c left moves = c x's
    for (int r = 1; r <= AD; r++) {
        cur[0] = (-4 * r) * 65536 + (r << 8) + r;  // column 0: r up moves
#pragma unroll
        for (int c = 1; c <= AD; c++) {
            const int up = prev[c] - 5 * 65536 + (1 << 8);
            const int left = cur[c - 1] - 5 * 65536 + (1 << 8);
            const bool m = (col[c - 1] >> (r - 1)) & 1u;
            const int diag = prev[c - 1] + (m ? 5 * 65536 : (-5 * 65536 + (1 << 8)));
            const int su = up >> 16, sl = left >> 16, sd = diag >> 16;
            int v;
            if (su >= sl)
                v = sd >= su ? diag : up;
            else
                v = sd >= sl ? diag : left;
            cur[c] = v;
        }
#pragma unroll
        for (int c = 0; c <= AD; c++) prev[c] = cur[c];
    }
    const int nx = (prev[AD] >> 8) & 0xFF, lead = prev[AD] & 0xFF;
    // Match.countErrorsInNeedleman (Match.java:L31-34): (float)#x - 0.9f * (float)lead, two roundings
    return __fsub_rn((float)nx, __fmul_rn(0.9f, (float)lead));
}

struct FinalAln {
    int ins, del, sub, nmis;
    bool term6;
    float end5, endn;
    int consec;   // NeedlemanMatch.getNconsecutiveMatchesNeedleman L160-173 (a run counts once a non-'.' follows it)
    int best_two; // getSumOfBestTwoMatchStretchesNeedleman L183-196 (the two SMALLEST closed runs > 4)
};

// full DP with 2-bit moves in LDS + walk from the end (SequenceAlignment.getTraceback L102-151,
// NeedlemanMatch.countNeedlemanErrorsInRead L68-86, countIndelsMismatchesEndOfRead L109-123,
// Match.hasN3pConsecutiveMatchesInNeedleman L41-50).  move: 0 diag match, 1 diag mismatch, 2 up, 3 left.
template <int AD>
__device__ __forceinline__ void nw_final(const uint32_t (&col)[AD], uint64_t *dirs, int tid, int n_end, FinalAln &out) {
    int prev[AD + 1], cur[AD + 1];
#pragma unroll
    for (int c = 0; c <= AD; c++) prev[c] = -5 * c;
    for (int r = 1; r <= AD; r++) {
        cur[0] = -4 * r;
        uint64_t row = 0;
#pragma unroll
        for (int c = 1; c <= AD; c++) {
            const int up = prev[c] - 5, left = cur[c - 1] - 5;
            const bool m = (col[c - 1] >> (r - 1)) & 1u;
            const int diag = prev[c - 1] + (m ? 5 : -5);
            int v;
            uint64_t mv;
            if (up >= left) {
                if (diag >= up) {
                    v = diag;
                    mv = m ? 0 : 1;
                } else {
                    v = up;
                    mv = 2;
                }
            } else {
                if (diag >= left) {
                    v = diag;
                    mv = m ? 0 : 1;
                } else {
                    v = left;
                    mv = 3;
                }
            }
            cur[c] = v;
            row |= mv << (2 * (c - 1));
        }
        dirs[(r - 1) * (kBlock / 2) + (tid >> 1)] = row;  // one lane of each pair reaches this
#pragma unroll
        for (int c = 0; c <= AD; c++) prev[c] = cur[c];
    }
    int r = AD, c = AD;
    int ins = 0, del = 0, sub = 0, trail = 0, cb = 0, t = 0;
    bool trailing = true, term = true;
    float e5 = 0.0f, en = 0.0f;
    // runs of '.', met in reverse: a run is closed (counts) iff an 'x' was met before it on the way back
    int run = 0, consec = 0, s1 = 0, s2 = 0, n_runs = 0;
    bool seen_x = false, run_closed = false;
    auto close_run = [&]() {
        if (run > 0 && run_closed) {
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
    while (r > 0 || c > 0) {
        int mv;
        if (r == 0)
            mv = 3;  // first row points left
        else if (c == 0)
            mv = 2;  // first column points up
        else
            mv = (int)((dirs[(r - 1) * (kBlock / 2) + (tid >> 1)] >> (2 * (c - 1))) & 3u);
        const bool x = mv != 0;
        const bool read_gap = mv == 3;
        ins += mv == 2;
        del += read_gap;
        sub += mv == 1;
        if (trailing && read_gap)
            trail++;
        else
            trailing = false;
        if (t < 6 && x) term = false;
        if (x) {
            close_run();
            seen_x = true;
        } else {
            if (run == 0) run_closed = seen_x;
            run++;
        }
        if (x) {
            if (cb < 5) e5 = cb <= 1 ? (float)((double)e5 + 1.2) : __fadd_rn(e5, 1.0f);
            if (cb < n_end) en = cb <= 1 ? (float)((double)en + 1.2) : __fadd_rn(en, 1.0f);
        }
        if (!read_gap) cb++;
        if (mv <= 1) {
            r--;
            c--;
        } else if (mv == 2)
            r--;
        else
            c--;
        t++;
    }
    close_run();
    out.consec = consec;
    out.best_two = (n_runs >= 1 ? s1 : 0) + (n_runs >= 2 ? s2 : 0);
    del = (int)(int8_t)(del - trail);
    out.ins = ins;
    out.del = del;
    out.sub = sub;
    out.nmis = ins + del + sub;
    out.term6 = term && t >= 6;
    out.end5 = e5;
    out.endn = en;
}

template <int AD>
__device__ __forceinline__ void load_cols(const uint32_t *planes, int tid, const ScanParams &P, int pos1,
                                          uint32_t (&col)[AD]) {
#pragma unroll
    for (int c = 0; c < AD; c++) col[c] = match32(planes, tid, P.adapter4[c], pos1 - 1) & ((1u << AD) - 1u);
}

// TSO "AACGCAGAGTACATGG" (Jar/config.xml:155) as 4-bit codes A=1 G=2 C=4 T=8
__device__ __forceinline__ uint32_t tso4(int i) {
    // packed nibbles, base i in bits [4i+3:4i]
    constexpr uint64_t TSO = 0x2281418212142411ull;  // A A C G C A G A G T A C A T G G  (low nibble first)
    return (uint32_t)(TSO >> (4 * i)) & 15u;
}
__device__ __forceinline__ void load_cols_tso(const uint32_t *planes, int tid, int pos1, uint32_t (&col)[16]) {
#pragma unroll
    for (int c = 0; c < 16; c++) col[c] = match32(planes, tid, tso4(c), pos1 - 1) & 0xFFFFu;
}

struct TsoMatch {
    int present, passed, nmis, end_scan, consec, best_two;
};

// PolyATadapterAnalyzerBase.scanForTSO (L324-369) on this lane's end: AdapterTSOanalyzer.scanForAdapterOrTSOseq with
// maxErrors = 5 (L84-110: candidates kept when Math.round(nErrors) <= 5, positions skipped by round(nErrors - 5) - 1
// after a bad candidate), then the final alignment of the first best position.
__device__ __forceinline__ int scan_tso_positions(const uint32_t *planes, int tid) {
    float best = 3.4028234663852886e+38f;
    int best_pos = 0, skip_until = 0;
    const int last = 90;  // min(116 - 16, windowForTSOsearch = 90)
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
        const int b = ch * 64;
        uint64_t any = 0, two = 0;
        uint64_t m0 = match64(planes, tid, tso4(0), b), m1 = match64(planes, tid, tso4(1), b + 1),
                 m2 = match64(planes, tid, tso4(2), b + 2);
#pragma unroll
        for (int i = 0; i + 3 < 16; i++) {
            const uint64_t m3 = match64(planes, tid, tso4(i + 3), b + i + 3);
            const uint64_t k = m0 & m1 & m2 & m3;
            two |= any & k;
            any |= k;
            m0 = m1;
            m1 = m2;
            m2 = m3;
        }
        uint64_t cand = two;
        const int hi = last - b;
        cand = hi <= 0 ? 0 : (hi >= 64 ? cand : (cand & ((1ull << hi) - 1ull)));
        while (cand) {
            const int i = __builtin_ctzll(cand);
            cand &= cand - 1;
            const int pos = b + i + 1;
            if (pos < skip_until) continue;  // jumped over by deltaPos
            uint32_t col[16];
            load_cols_tso(planes, tid, pos, col);
            const float ne = nw_errors<16>(col);
            if (!((float)(int)floorf(__fadd_rn(ne, 0.5f)) > 5.0f)) {  // Math.round(nErrors) <= maxErrors
                if (ne < best) {
                    best = ne;
                    best_pos = pos;
                }
            }
            if (5.0f < ne) {
                int d = (int)floorf(__fadd_rn(__fsub_rn(ne, 5.0f), 0.5f)) - 1;
                if (d < 1) d = 1;
                skip_until = pos + d;
            }
        }
    }
    return best_pos;  // 0 = AdapterScanRslt empty
}

__device__ __forceinline__ void tso_final(const uint32_t *planes, uint64_t *dirs, int tid, int best_pos, TsoMatch &m) {
    uint32_t col[16];
    load_cols_tso(planes, tid, best_pos, col);
    FinalAln a;
    nw_final<16>(col, dirs, tid, 0, a);
    m.present = 1;
    m.nmis = a.nmis;
    m.passed = a.nmis <= 5;
    m.end_scan = best_pos + 15 + a.ins - a.del;
    m.consec = a.consec;
    m.best_two = a.best_two;
}

// ---------------------------------------------------------------------------------------------------------------
template <int AD>
__global__ __launch_bounds__(kBlock) void k_scan(const uint32_t *__restrict__ ends, const int32_t *__restrict__ read_len,
                                                 const uint8_t *__restrict__ qtail, const uint32_t *__restrict__ qsum,
                                                 size_t n_reads, ScanParams P, smi_scan_result *__restrict__ out,
                                                 smi_bc_window *__restrict__ windows) {
    extern __shared__ uint32_t lds[];
    uint32_t *planes = lds;                                                   // [5][kLdsWords][kBlock]
    uint64_t *dirs = reinterpret_cast<uint64_t *>(lds + 5 * kLdsWords * kBlock);  // [AD][kBlock / 2]
    const int tid = threadIdx.x;
    const size_t n_ends = 2 * n_reads;
    for (size_t e0 = (size_t)blockIdx.x * kBlock; e0 < n_ends; e0 += (size_t)gridDim.x * kBlock) {
        const size_t e = e0 + tid;
        const bool active = e < n_ends;
        const size_t read = e >> 1;
        const int side = (int)(e & 1);  // 0 = head (forward scan), 1 = reverse-complemented tail
        // ---- stage the bit-planes ---------------------------------------------------------------------------
        uint32_t ta[kLdsWords];
#pragma unroll
        for (int w = 0; w < kLdsWords; w++) ta[w] = 0xFFFFFFFFu;
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int w = 0; w < kLdsWords; w++) {
                uint32_t v = 0;
                if (active && w < kPlaneWords) v = ends[(size_t)(c * kPlaneWords + w) * n_ends + e];
                planes[(c * kLdsWords + w) * kBlock + tid] = v;
                ta[w] = c == 3 ? (ta[w] & v) : (ta[w] & ~v);
            }
#pragma unroll
        for (int w = 0; w < kLdsWords; w++) planes[(4 * kLdsWords + w) * kBlock + tid] = ta[w];  // exact T = T & ~A & ~G & ~C
        // (each lane only ever reads its own column: no barrier needed)
        const int len = active ? read_len[read] : 0;
        const bool long_enough = len >= P.min_read_length;  // testReadLength L131-137
        const uint32_t *tex = planes + 4 * kLdsWords * kBlock;

        // ---- polyT + adapter scan on this end ----------------------------------------------------------------
        int pb = 0, pe = 0;
        const bool has_t = active && long_enough && find_polyt(tex, tid, P, pb, pe);
        float best = 3.4028234663852886e+38f;
        uint64_t bm[3] = {0, 0, 0};
        int n_all = 0;
        if (has_t) {
            // scan positions 1 .. min(pe - AD, pe - 12)  (seqTilPolyAend has length pe; L49-61, L87)
            const int last = min(pe - AD, pe - 12);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const int b = ch * 64;
                uint64_t any = 0, two = 0;
                uint64_t m0 = match64(planes, tid, P.adapter4[0], b), m1 = match64(planes, tid, P.adapter4[1], b + 1),
                         m2 = match64(planes, tid, P.adapter4[2], b + 2);
#pragma unroll
                for (int i = 0; i + 3 < AD; i++) {
                    const uint64_t m3 = match64(planes, tid, P.adapter4[i + 3], b + i + 3);
                    const uint64_t k = m0 & m1 & m2 & m3;
                    two |= any & k;
                    any |= k;
                    m0 = m1;
                    m1 = m2;
                    m2 = m3;
                }
                // bit i <-> pos = b + i + 1 ; keep 1 <= pos <= last
                uint64_t cand = two;
                const int hi = last - b;  // number of valid bits in this chunk
                cand = hi <= 0 ? 0 : (hi >= 64 ? cand : (cand & ((1ull << hi) - 1ull)));
                while (cand) {
                    const int i = __builtin_ctzll(cand);
                    cand &= cand - 1;
                    uint32_t col[AD];
                    load_cols<AD>(planes, tid, P, b + i + 1, col);
                    const float ne = nw_errors<AD>(col);
                    n_all++;
                    if (ne < best) {
                        best = ne;
                        bm[0] = bm[1] = bm[2] = 0;
                    }
                    if (ne == best) bm[ch] |= 1ull << i;
                }
            }
        }
        // ---- strand decision (PolyATadapterAnalyzerBase.analyze L145-163): lanes 2i and 2i+1 exchange -------------
        const int o_has_t = __shfl_xor((int)has_t, 1);
        const int o_n_all = __shfl_xor(n_all, 1);
        const float o_best = __shfl_xor(best, 1);
        const bool f_has = side == 0 ? has_t : (bool)o_has_t, r_has = side == 0 ? (bool)o_has_t : has_t;
        const int f_n = side == 0 ? n_all : o_n_all, r_n = side == 0 ? o_n_all : n_all;
        const float f_best = side == 0 ? best : o_best, r_best = side == 0 ? o_best : best;
        uint32_t flags = 0;
        int use_fwd = -1;
        if (!long_enough) {
            flags |= SMI_F_READ_TOO_SHORT | SMI_F_FAILED;
        } else {
            flags |= (!f_has && !r_has) ? SMI_F_POLY_A_NOT_FOUND
                     : (f_has && !r_has) ? SMI_F_POLY_T_5P
                     : (!f_has && r_has) ? SMI_F_POLY_A_3P
                                         : SMI_F_POLY_T_5P_POLY_A_3P;
            const bool f_ne = f_has && f_n > 0, r_ne = r_has && r_n > 0;
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

        // ---- accepted alignment (getMatchList L275-319, createNeedlemanMatch L237-253) ------------------------------
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
        if (chosen) {
            const int n_best = __popcll(bm[0]) + __popcll(bm[1]) + __popcll(bm[2]);
            bool have = false;
            float best_key = 0.0f;
            FinalAln fa;
            int f_pos = 0;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                uint64_t m = bm[ch];
                while (m) {
                    const int i = __builtin_ctzll(m);
                    m &= m - 1;
                    const int pos = ch * 64 + i + 1;
                    uint32_t col[AD];
                    load_cols<AD>(planes, tid, P, pos, col);
                    FinalAln a;
                    nw_final<AD>(col, dirs, tid, P.min_3p, a);
                    // MIN_3P_CONSEC_MATCHES_TO_OVERRIDE_PASS = 6 (AdapterParameters.java:L22)
                    const bool ok = a.nmis <= P.max_mm || a.term6;
                    if (!ok) continue;
                    // one offset: taken as is; several: smallest countIndelsMismatchesEndOfRead(5) group, first of it
                    if (!have || (n_best > 1 && a.end5 < best_key)) {
                        have = true;
                        best_key = a.end5;
                        fa = a;
                        f_pos = pos;
                    }
                }
            }
            // polyA coordinates are set as soon as a side is chosen (analyze L169-171)
            res.polya_start = len - (pe - 1);
            res.polya_end = len - (pb - 1);
            if (!have) {
                res.flags |= SMI_F_FAILED;  // L217
            } else {
                const int s_end = f_pos + AD - 1 + fa.ins - fa.del;  // L251
                res.found = 1;
                res.scan_end = (int16_t)s_end;
                res.adapter_start = len - (f_pos - 1);  // ReadScanResult.java:L446-447
                res.adapter_end = len - (s_end - 1);
                res.adapter_nmis = (int16_t)fa.nmis;
                res.reverse = use_fwd ? 1 : 0;
                res.flags |= use_fwd ? (SMI_F_ADAPTER_5P | SMI_F_PASSED_REV) : (SMI_F_ADAPTER_3P | SMI_F_PASSED_FWD);
                // barcode window: stranded[AE-22 .. AE+1] = reverse complement of scan[s_end-1 .. s_end+22]
                const int hi_sp = s_end + 22, lo_sp = s_end - 1;
                if (lo_sp >= 1 && hi_sp <= len && hi_sp <= kEndBases) {
                    uint64_t bases = 0;
                    uint32_t nmask = 0;
#pragma unroll 4
                    for (int j = 0; j < 24; j++) {
                        const int bit = hi_sp - j - 1;  // scan position hi_sp - j, 0-based bit
                        const uint32_t a = get32(planes + 0 * kLdsWords * kBlock, tid, bit) & 1u;
                        const uint32_t g = get32(planes + 1 * kLdsWords * kBlock, tid, bit) & 1u;
                        const uint32_t c = get32(planes + 2 * kLdsWords * kBlock, tid, bit) & 1u;
                        const uint32_t t = get32(planes + 3 * kLdsWords * kBlock, tid, bit) & 1u;
                        // complement: A<->T, G<->C ; 2-bit code A0 G1 C2 T3
                        const uint32_t single = (a + g + c + t) == 1u;
                        const uint32_t code = t ? 0u : (c ? 1u : (g ? 2u : 3u));
                        bases = (bases << 2) | (single ? code : 0u);
                        nmask |= (single ? 0u : 1u) << j;
                    }
                    win.bases = bases;
                    win.nmask = nmask;
                    win.flags = SMI_WIN_VALID;
                }
                // pass-1 quality filter (short-circuit && chain; the UNSTRANDED quality string is indexed with
                // stranded coordinates, UsedCellBCListGenerator.java:L201)
                if (qtail != nullptr && fa.endn == 0.0f) {
                    const int ae = res.adapter_end;
                    // raw 1-based positions ae-16 .. ae-1 ; qtail is right-aligned: index = kEndBases - 1 - (len - p)
                    int sum = 0;
                    bool in_range = ae - 16 >= 1;
                    for (int p = ae - 16; p <= ae - 1; p++) {
                        const int idx = kEndBases - 1 - (len - p);
                        if (idx < 0 || idx >= kEndBases) {
                            in_range = false;
                            break;
                        }
                        sum += (int)qtail[(size_t)read * kEndBases + idx] - 33;
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
        // ---- TSO scan on both ends (PolyATadapterAnalyzer_3pBCUMI.scanReadForTSOs L122-190) -------------------------------
        TsoMatch tm = {0, 0, 0, 0, 0, 0};
        const int tso_pos = (active && long_enough) ? scan_tso_positions(planes, tid) : 0;
#pragma unroll
        for (int sd = 0; sd < 2; sd++)  // the two lanes of a pair share one direction slot: one side at a time
            if (side == sd && tso_pos) tso_final(planes, dirs, tid, tso_pos, tm);
        {
            TsoMatch o;
            o.present = __shfl_xor(tm.present, 1);
            o.passed = __shfl_xor(tm.passed, 1);
            o.nmis = __shfl_xor(tm.nmis, 1);
            o.end_scan = __shfl_xor(tm.end_scan, 1);
            o.consec = __shfl_xor(tm.consec, 1);
            o.best_two = __shfl_xor(tm.best_two, 1);
            TsoMatch f = side == 0 ? tm : o, r = side == 0 ? o : tm;
            auto found = [](const TsoMatch &x) { return x.present && x.passed; };
            if (!found(f) && !found(r)) {  // L146-153: rescue by >= 8 consecutive matches (config.xml:161)
                if (f.present) f.passed = f.consec >= 8;
                if (r.present) r.passed = r.consec >= 8;
                if (!found(f) && !found(r)) {  // L155-162: rescue by the two stretches >= 12 (config.xml:164)
                    if (f.present) f.passed = f.best_two >= 12;
                    if (r.present) r.passed = r.best_two >= 12;
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
    }
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
    for (int i = 0; i < 22; i++) P.adapter4[i] = i < ad ? cfg->adapter4[i] : 0u;
    const size_t n_ends = 2 * n;
    const unsigned grid = (unsigned)std::min<size_t>((n_ends + kBlock - 1) / kBlock, 256 * 16);
    if (int rc = time_begin(ctx, SMI_K_SCAN, s)) return rc;
    if (ad == 10) {
        const size_t lds = 5 * kLdsWords * kBlock * 4 + 16 * (kBlock / 2) * 8;  // 16 rows: the TSO alignment
        hipLaunchKernelGGL(k_scan<10>, dim3(grid), dim3(kBlock), lds, s, d_ends, d_len, d_qtail, d_qsum, n, P, d_out, d_win);
    } else {
        const size_t lds = 5 * kLdsWords * kBlock * 4 + 22 * (kBlock / 2) * 8;
        hipLaunchKernelGGL(k_scan<22>, dim3(grid), dim3(kBlock), lds, s, d_ends, d_len, d_qtail, d_qsum, n, P, d_out, d_win);
    }
    SMI_HIP(hipGetLastError());
    if (int rc = time_end(ctx, SMI_K_SCAN, s)) return rc;
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// K-PACK: ASCII reads -> scan-orientation bit-plane ends (+ read length, tail qualities, quality sum).
// One wave per read: lanes stride over the read for the quality sum (coalesced), then build the planes with
// ballots (lane = base position).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t enc4(uint8_t c) {
    // NucleicAcidByteCodeBase.ENCODE_MATRIX (TB!nuc/encoding/NucleicAcidByteCodeBase.java:L45-78), ACGTN subset;
    // every other character is treated as N
    switch (c) {
    case 'A': case 'a': return 1;
    case 'G': case 'g': return 2;
    case 'C': case 'c': return 4;
    case 'T': case 't': return 8;
    default: return 15;
    }
}
__device__ __forceinline__ uint32_t comp4(uint32_t b) {
    return ((b & 1u) << 3) | ((b & 8u) >> 3) | ((b & 2u) << 1) | ((b & 4u) >> 1);
}

__global__ __launch_bounds__(256) void k_pack_ends(const uint8_t *__restrict__ reads, const uint8_t *__restrict__ quals,
                                                   const uint64_t *__restrict__ offsets, size_t n,
                                                   uint32_t *__restrict__ ends, int32_t *__restrict__ read_len,
                                                   uint8_t *__restrict__ qtail, uint32_t *__restrict__ qsum) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_ends = 2 * n;
    for (size_t r = wave; r < n; r += n_waves) {
        const uint64_t beg = offsets[r];
        const int64_t len = (int64_t)(offsets[r + 1] - beg);
        if (lane == 0) read_len[r] = (int32_t)len;
        if (quals) {
            uint32_t s = 0;
            for (int64_t i = lane; i < len; i += 64) s += (uint32_t)quals[beg + i] - 33u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) qsum[r] = s;
            for (int i = lane; i < kEndBases; i += 64) {
                const int64_t p = len - kEndBases + i;  // right-aligned
                qtail[r * kEndBases + i] = p >= 0 ? quals[beg + p] : (uint8_t)33;
            }
        }
        for (int side = 0; side < 2; side++) {
            for (int w = 0; w < kPlaneWords; w += 2) {
                // 64 base positions per step: position p = 32*w + lane
                const int p = 32 * w + lane;
                uint32_t code = 0;  // '-' (matches nothing) beyond the read / the stored end
                if (p < kEndBases && p < len) {
                    if (side == 0)
                        code = enc4(reads[beg + p]);
                    else
                        code = comp4(enc4(reads[beg + (len - 1 - p)]));
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const unsigned long long b = __ballot((code >> c) & 1u);
                    if (lane == 0) {
                        ends[(size_t)(c * kPlaneWords + w) * n_ends + 2 * r + side] = (uint32_t)b;
                        if (w + 1 < kPlaneWords) ends[(size_t)(c * kPlaneWords + w + 1) * n_ends + 2 * r + side] = (uint32_t)(b >> 32);
                    }
                }
            }
        }
    }
}

int launch_pack_ends(smi_ctx *, const uint8_t *d_reads, const uint8_t *d_quals, const uint64_t *d_offsets, size_t n,
                     uint32_t *d_ends, int32_t *d_len, uint8_t *d_qtail, uint32_t *d_qsum, hipStream_t s) {
    if (!n) return SMI_OK;
    const unsigned grid = (unsigned)std::min<size_t>((n + 3) / 4, 256 * 32);
    hipLaunchKernelGGL(k_pack_ends, dim3(grid), dim3(256), 0, s, d_reads, d_quals, d_offsets, n, d_ends, d_len, d_qtail, d_qsum);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

}  // namespace smi
