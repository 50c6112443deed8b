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
#include "smi_internal.h"

namespace smi {

// bit k of the result = nibble k of w equals code c (k = 0..13)
__device__ __forceinline__ uint32_t eq_mask(uint64_t w, uint32_t c) {
    uint64_t x = w ^ (0x1111111111111111ull * c);
    // nibble == 0  <=>  (((x & 0x7..7) + 0x7..7) | x) has bit 3 clear
    uint64_t z = ~((((x & 0x7777777777777777ull) + 0x7777777777777777ull) | x)) & 0x8888888888888888ull;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) m |= (uint32_t)((z >> (4 * k + 3)) & 1ull) << k;
    return m;
}

// Myers / Hyyro global edit distance of the 12-base pattern that starts at base `pi` of read a against the 12 text
// bases that start at base `tj` of read b.  eq[j] = pattern positions (14 bits, of read a) matching text base j of read b.
__device__ __forceinline__ int myers12(const uint32_t (&eq)[14], int pi, int tj) {
    const uint32_t M = 0xFFFu, TOP = 0x800u;
    uint32_t Pv = M, Mv = 0;
    int score = 12;
#pragma unroll
    for (int t = 0; t < 12; t++) {
        const uint32_t Eq = (eq[tj + t] >> pi) & M;
        const uint32_t Xv = Eq | Mv;
        const uint32_t Xh = ((((Eq & Pv) + Pv) ^ Pv) | Eq) & M;
        uint32_t Ph = (Mv | ~(Xh | Pv)) & M;
        uint32_t Mh = Pv & Xh;
        score += (Ph & TOP) ? 1 : 0;
        score -= (Mh & TOP) ? 1 : 0;
        Ph = ((Ph << 1) | 1u) & M;  // global distance: the boundary row grows by one per text character
        Mh = (Mh << 1) & M;
        Pv = (Mh | ~(Xv | Ph)) & M;
        Mv = Ph & Xv;
    }
    return score;
}

__device__ __forceinline__ uint32_t umi_pair(uint64_t a, uint64_t b) {
    const uint32_t mA = eq_mask(a, 1), mG = eq_mask(a, 2), mC = eq_mask(a, 4), mT = eq_mask(a, 8), mN = eq_mask(a, 15);
    // match masks per text base, shared by the nine alignments (a code outside A, G, C, T, N matches nothing, as equals() would)
    uint32_t eq[14];
#pragma unroll
    for (int j = 0; j < 14; j++) {
        const uint32_t c = (uint32_t)(b >> (4 * j)) & 15u;
        eq[j] = c == 1u ? mA : c == 2u ? mG : c == 4u ? mC : c == 8u ? mT : c == 15u ? mN : 0u;
    }
    // calcBestEditDistance L67-80: start (127, MINUSONE, MINUSONE); visit values 1,2,0 x 1,2,0; strict <
    int best = 127, b1 = 0, b2 = 0;
    const int ORDER[3] = {1, 2, 0};
#pragma unroll
    for (int x = 0; x < 3; x++)
#pragma unroll
        for (int y = 0; y < 3; y++) {
            const int i = ORDER[x], v = ORDER[y];
            int d = myers12(eq, i, v);
            d = d > 4 ? 5 : d;  // limitedCompare: -1 above the threshold, stored as 5 (L343)
            if (d < best) {
                best = d;
                b1 = i;
                b2 = v;
            }
        }
    return (uint32_t)best | ((uint32_t)b1 << 4) | ((uint32_t)b2 << 6);
}

__global__ __launch_bounds__(256) void k_umi_dist(const uint64_t *__restrict__ windows, const uint32_t *__restrict__ group_off,
                                                  const uint64_t *__restrict__ pair_off, const uint64_t *__restrict__ mat_off,
                                                  uint32_t n_groups, uint64_t total_pairs, uint8_t *__restrict__ out) {
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < total_pairs; t += (uint64_t)gridDim.x * blockDim.x) {
        // group of this pair: last g with pair_off[g] <= t.  The pairs of a wave are consecutive, so the binary search
        // runs once for the wave's first pair (scalar) and every lane walks forward from there
        const uint64_t t_first = __builtin_amdgcn_readfirstlane((uint32_t)(t >> 32)) * 0x100000000ull +
                                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)t);
        uint32_t lo = 0, hi = n_groups;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pair_off[mid] <= t_first)
                lo = mid;
            else
                hi = mid;
        }
        uint32_t g = lo;
        while (g + 1 < n_groups && pair_off[g + 1] <= t) g++;
        const uint64_t local = t - pair_off[g];
        const uint32_t r0 = group_off[g];
        const uint64_t n = group_off[g + 1] - r0;
        // row i of the upper triangle (rows have n, n-1, ... entries): first(i) = i*n - i*(i-1)/2
        uint64_t i = (uint64_t)(((double)(2 * n + 1) - sqrt((double)(2 * n + 1) * (double)(2 * n + 1) - 8.0 * (double)local)) * 0.5);
        if (i >= n) i = n - 1;
        while (i > 0 && i * n - i * (i - 1) / 2 > local) i--;
        while ((i + 1) * n - (i + 1) * i / 2 <= local) i++;
        const uint64_t v = i + (local - (i * n - i * (i - 1) / 2));
        const uint32_t r = umi_pair(windows[r0 + i], windows[r0 + v]);
        uint8_t *m = out + mat_off[g];
        m[i * n + v] = (uint8_t)r;
        // transposed copy for the lower triangle (getTransposedEditDistance L133, L213-216): offsets swapped
        m[v * n + i] = (uint8_t)((r & 15u) | (((r >> 6) & 3u) << 4) | (((r >> 4) & 3u) << 6));
    }
}

int launch_umi_dist(smi_ctx *ctx, const uint64_t *d_windows, const uint32_t *d_group_off, const uint64_t *d_pair_off,
                    const uint64_t *d_mat_off, uint32_t n_groups, uint64_t total_pairs, uint8_t *d_out, hipStream_t s) {
    if (!total_pairs) return SMI_OK;
    const unsigned grid = (unsigned)std::min<uint64_t>((total_pairs + 255) / 256, 256ull * 64);
    if (int rc = time_begin(ctx, SMI_K_UMI, s)) return rc;
    hipLaunchKernelGGL(k_umi_dist, dim3(grid), dim3(256), 0, s, d_windows, d_group_off, d_pair_off, d_mat_off, n_groups,
                       total_pairs, d_out);
    SMI_HIP(hipGetLastError());
    if (int rc = time_end(ctx, SMI_K_UMI, s)) return rc;
    return SMI_OK;
}

}  // namespace smi
