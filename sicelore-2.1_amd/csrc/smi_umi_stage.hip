// smi_umi_stage.hip -- the UMI stage of `assignumis` on the device (gfx950): what smi_assignumis_chunk runs between the upload of a
// BamReader chunk's read names / flags / positions / CIGARs and the download of the per-record UMI tags.
//
// Reference units (bytecode; citation form in DESIGN.md):
//   FastqRecordExt.getScanDatFromReadName                FJ!nanoporereadscanner/readerwriter/FastqRecordExt.java:L395-496   -> K-UPARSE
//   NanoporeRead$ReadScanData.generateReadScanData / getGenomePosition / getReferencePositionAtReadPosition
//                                                        FJ!umifinder/reads/nanopore/NanoporeRead$ReadScanData.java:L86-153 -> K-UPARSE
//   ClusteringEditDistanceBase.lambda$static$7 (UMI window) FJ!clustering/ClusteringEditDistanceBase.java:L297-350          -> K-UPARSE
//   UmiClustering.groupDataByCellAndRegion               FJ!umifinder/analyzers/clustering/UmiClustering.java:L105         -> key sort
//   ClusterOneHierarchical.call + LingPipe CompleteLinkClusterer / Dendrogram.partitionDistance
//                                                        ...ClusterOneHierarchical.java:L66-217, AL!cluster/CompleteLinkClusterer.java:L146-237 -> K-UCLUST
//   OneUmiCluster.setClusterCenter*, ClusterOneBase.setSamflagsAndStatsForClustered
//                                                        FJ!clustering/OneUmiCluster.java:L49-65, ...ClusterOneBase.java:L118-168 -> K-UCLUST, K-UTAG
// The host keeps two steps: the genomic-region grouping (ReadGrouper: a sequential refinement over position-sorted reads, run per strand
// on two threads) and the clusterer of groups of more than 100 reads (ClusterOne_MyClustering), for which the group's matrix comes back.
//
// MI355X mapping.  K-UPARSE: one lane per record, the name staged through LDS by the wave (coalesced 16-byte loads), one pass over its
// characters.  Grouping: a 64-bit key (region, 2-bit barcode) per eligible record, hipcub radix sort (stable: members stay in input
// order) + run-length encoding + scans.  K-UCLUST: one wave per (cell, region) group; the pair queue of LingPipe's clusterer lives in
// LDS as a score and an insertion-id matrix, the queue's order (score ascending, then the LATEST insertion first) is a wave-wide arg-min,
// and a merge re-issues the insertion ids in the order the reference offers the new pairs.  Integer / byte work: no MFMA.
#include <hipcub/hipcub.hpp>

#include <cmath>

#include "smi_internal.h"
#include "smi_umi_stage.h"

namespace smi {

// ---------------------------------------------------------------------------------------------------------------------------
// K-UPARSE
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kNameStage = 320;  // characters of a name staged in LDS per record; longer names (never seen from scanfastq) take global loads

struct NameView {
    const char *lds;     // staged copy (its first `staged` characters)
    const char *glob;    // the name in global memory
    int len, staged;
    __device__ __forceinline__ char at(int i) const { return i < staged ? lds[i] : glob[i]; }
};

// Integer.parseInt of v[a, b): digits with an optional sign, at most 10 digits; *nonstd is set for anything the host's parser might read
// differently (leading blanks, '+', overflow) so that the chunk takes the host path instead
__device__ __forceinline__ bool parse_int(const NameView &v, int a, int b, long *out, bool *nonstd) {
    if (b <= a) return false;
    int i = a;
    bool neg = false;
    if (v.at(i) == '-') {
        neg = true;
        i++;
    }
    if (i >= b || b - i > 10) {
        *nonstd = true;
        return false;
    }
    long x = 0;
    for (; i < b; i++) {
        const char c = v.at(i);
        if (c < '0' || c > '9') {
            if (c == ' ' || c == '+' || c == '\t') *nonstd = true;  // strtol would have skipped / accepted these
            return false;
        }
        x = x * 10 + (c - '0');
    }
    if (x > 2147483647L + (neg ? 1 : 0)) {
        *nonstd = true;
        return false;
    }
    *out = neg ? -x : x;
    return true;
}

__device__ __forceinline__ uint32_t ucode4(char c) { return c == 'A' ? 1u : c == 'G' ? 2u : c == 'C' ? 4u : c == 'T' ? 8u : 15u; }
__device__ __forceinline__ uint32_t ucomp4(uint32_t c) { return c == 1 ? 8u : c == 8 ? 1u : c == 2 ? 4u : c == 4 ? 2u : 15u; }

// NanoporeRead$ReadScanData.getReferencePositionAtReadPosition on a BAM CIGAR (smi_ref_position_at_read_position)
__device__ __forceinline__ bool ref_position(const uint32_t *cigar, int n_cigar, int alignment_start, int position, int *out) {
    if (position == 0) return false;
    int last_ref_end = 1, last_read_end = 1, read_at = 1, ref_at = alignment_start;
    for (int i = 0; i < n_cigar; i++) {
        const uint32_t op = cigar[i] & 15u;
        const int len = (int)(cigar[i] >> 4);
        if (op == 1 || op == 4)
            read_at += len;
        else if (op == 2 || op == 3)
            ref_at += len;
        else if (op == 0 || op == 7 || op == 8) {
            const int block_read = read_at, block_ref = ref_at;
            read_at += len;
            ref_at += len;
            if (block_read + len - 1 < position) {
                last_ref_end = block_ref + len - 1;
                last_read_end = block_read + len - 1;
                continue;
            }
            *out = position < block_read ? block_ref - abs(block_ref - last_ref_end) / 2 : block_ref + position - block_read;
            return true;
        }
    }
    if (position - last_read_end < 300) {
        *out = last_ref_end;
        return true;
    }
    return false;
}

__global__ __launch_bounds__(64) void k_umi_parse(const char *__restrict__ names, const uint32_t *__restrict__ name_off, const uint16_t *__restrict__ flags,
                                                  const int32_t *__restrict__ pos0, const uint32_t *__restrict__ cigars,
                                                  const uint32_t *__restrict__ cigar_off, int n, int five, int grouping_distance, int bc_edit_limit,
                                                  int umi_len, uint64_t random_umi_seed, UmiParsed *__restrict__ out) {
    // The names of the wave's 64 records are consecutive in memory: one coalesced copy of that byte range into LDS (16 bytes per lane and
    // step), every lane then reads its own name from there.  (Staging row by row -- 64 rows x 4 dependent rounds -- made this kernel
    // latency-bound: 0.83 ms per 120 k records.)  A block of unusually long names falls back to rows of kNameStage characters.
    __shared__ __attribute__((aligned(16))) char stage[64 * (kNameStage + 4)];
    const int lane = threadIdx.x;
    const int rec0 = blockIdx.x * 64;
    const int rec1 = min(rec0 + 64, n);
    const uint32_t base = name_off[rec0], total = name_off[rec1] - base;
    const bool flat = total <= (uint32_t)sizeof(stage) - 16;
    if (flat) {
        for (uint32_t o = 16u * lane; o < total; o += 1024u) {
            uint32_t w[4];
            __builtin_memcpy(w, names + base + o, 16);  // (the name buffer carries 16 spare bytes behind its end)
            __builtin_memcpy(stage + o, w, 16);
        }
    } else {
        for (int r = 0; r < rec1 - rec0; r++) {
            const uint32_t a = name_off[rec0 + r];
            const int len = min((int)(name_off[rec0 + r + 1] - a), kNameStage);
            for (int k = lane; k < len; k += 64) stage[r * (kNameStage + 4) + k] = names[a + k];
        }
    }
    __syncthreads();
    const int i = rec0 + lane;
    if (i >= n) return;
    UmiParsed P;
    P.win = 0;
    P.bc = 0;
    P.cpos = 0;
    P.q = 0.0f;
    P.flags = 0;
    const uint32_t a = name_off[i];
    const int nlen = (int)(name_off[i + 1] - a);
    NameView v{flat ? stage + (a - base) : stage + lane * (kNameStage + 4), names + a, nlen, flat ? nlen : min(nlen, kNameStage)};
    const uint16_t fl = flags[i];
    if (fl & 16) P.flags |= UP_REV;
    // The characters go through a rolling 64-bit window (one LDS read per position): the first `_REV_`, else the first `_FWD_`, in pass
    // one; the first occurrence of every tag behind it in pass two.  (Comparing every position against every pattern character by
    // character cost thirteen reads per position.)
    auto pat = [](const char *t, int n) {
        uint64_t w = 0;
        for (int k = 0; k < n; k++) w = (w << 8) | (uint8_t)t[k];
        return w;
    };
    const uint64_t kRev = pat("_REV_", 5), kFwd = pat("_FWD_", 5), kM5 = 0xFFFFFFFFFFull, kM3 = 0xFFFFFFull, kM2 = 0xFFFFull, kM6 = 0xFFFFFFFFFFFFull;
    // (the staged characters are read a dword at a time: a quarter of the LDS instructions of byte reads)
    struct ByteStream {
        const NameView &v;
        const uint32_t *w32;
        uint32_t cur;
        int abs, k;
        bool fast;
        __device__ ByteStream(const NameView &nv, const char *stage_base, bool flat, int from) : v(nv), w32(reinterpret_cast<const uint32_t *>(stage_base)), cur(0), abs(0), k(from), fast(flat) {
            if (fast) {
                abs = (int)(nv.lds - stage_base) + from;
                cur = w32[abs >> 2];
            }
        }
        __device__ __forceinline__ uint8_t next() {
            if (!fast) return (uint8_t)v.at(k++);
            const uint8_t b = (uint8_t)(cur >> (8 * (abs & 3)));
            abs++;
            k++;
            if ((abs & 3) == 0) cur = w32[abs >> 2];
            return b;
        }
    };
    int m_rev = -1, m_fwd = -1;
    {
        uint64_t w = 0;
        ByteStream bs(v, stage, flat, 0);
        for (int k = 0; k < v.len && m_rev < 0; k++) {
            w = (w << 8) | bs.next();
            if (k >= 4) {
                if ((w & kM5) == kRev) m_rev = k - 4;
                if ((w & kM5) == kFwd && m_fwd < 0) m_fwd = k - 4;
            }
        }
    }
    const int mark = m_rev >= 0 ? m_rev : m_fwd;
    if (mark >= 0) {
        const int s0 = mark + 4;  // `sub` of the reference: from the marker's closing '_' on
        int p_ae = -1, p_ps = -1, p_ed = -1, p_bc = -1, p_bce = -1, p_x = -1, p_q = -1;
        const uint64_t kAE = pat("AE=", 3), kPS = pat("PS=", 3), kED = pat("ed=", 3), kBC = pat("bc=", 3), kX = pat("X=", 2), kQ = pat("Q=", 2),
                       kBCE = pat("bcEnd=", 6);
        uint64_t w = 0;
        ByteStream bs(v, stage, flat, s0);
        for (int k = s0; k < v.len; k++) {
            w = (w << 8) | bs.next();
            const int have = k - s0 + 1;  // characters of `sub` in the window
            if (have >= 2) {
                if ((w & kM2) == kX && p_x < 0) p_x = k + 1;
                if ((w & kM2) == kQ && p_q < 0) p_q = k + 1;
            }
            if (have >= 3) {
                const uint64_t t = w & kM3;
                if (t == kAE && p_ae < 0) p_ae = k + 1;
                if (t == kPS && p_ps < 0) p_ps = k + 1;
                if (t == kED && p_ed < 0) p_ed = k + 1;
                if (t == kBC && p_bc < 0) p_bc = k + 1;
            }
            if (have >= 6 && (w & kM6) == kBCE && p_bce < 0) p_bce = k + 1;
        }
        auto value_end = [&](int from) {
            int b = from;
            while (b < v.len && v.at(b) != '_') b++;
            return b;
        };
        bool nonstd = false;
        long ae = 0, ps = 0, ed = 0, bc_end = 0;
        if (p_ae < 0 || !parse_int(v, p_ae, value_end(p_ae), &ae, &nonstd)) {
            P.flags |= nonstd ? UP_NONSTD : UP_ERROR;  // AdapterInfoNotFoundInReadException
        } else {
            P.flags |= UP_PRESENT;
            const bool has_ps = p_ps >= 0 && parse_int(v, p_ps, value_end(p_ps), &ps, &nonstd);
            bool has_bc = false, has_bc_end = false;
            if (p_ed >= 0 && parse_int(v, p_ed, value_end(p_ed), &ed, &nonstd) && (bc_edit_limit < 0 || ed <= bc_edit_limit)) {
                if (p_bc >= 0) {
                    has_bc = true;
                    const int e = value_end(p_bc);
                    if (e - p_bc != 16)
                        nonstd = true;  // grouping keys are 16-mers here; anything else goes through the host path
                    else {
                        uint32_t key = 0;
                        for (int k = 0; k < 16; k++) {
                            const char c = v.at(p_bc + k);
                            const uint32_t t = c == 'A' ? 0u : c == 'G' ? 1u : c == 'C' ? 2u : c == 'T' ? 3u : 4u;
                            if (t > 3u) nonstd = true;
                            key = (key << 2) | (t & 3u);
                        }
                        P.bc = key;
                    }
                }
                if (p_bce >= 0) has_bc_end = parse_int(v, p_bce, value_end(p_bce), &bc_end, &nonstd);
            }
            if (has_bc) P.flags |= UP_HAS_BC;
            // Q=: Float.parseFloat of the text up to a blank; the forms scanfastq writes ("12", "12.3", ".5") are evaluated exactly
            // (digits / 10^k in fp32 division is the correctly rounded value); anything else is left to the host
            bool has_q = false;
            if (p_q >= 0) {
                const int e = value_end(p_q);
                if (e - p_q < 30) {
                    int b = p_q;
                    uint32_t mant = 0;
                    int n_dig = 0, n_frac = -1;
                    for (; b < e && v.at(b) != ' '; b++) {
                        const char c = v.at(b);
                        if (c >= '0' && c <= '9') {
                            mant = mant * 10u + (uint32_t)(c - '0');
                            n_dig++;
                            if (n_frac >= 0) n_frac++;
                        } else if (c == '.' && n_frac < 0)
                            n_frac = 0;
                        else
                            nonstd = true;
                    }
                    if (b > p_q) {
                        has_q = true;
                        if (n_dig == 0 || n_dig > 7 || n_frac > 3) nonstd = true;
                        const float den = n_frac <= 0 ? 1.0f : n_frac == 1 ? 10.0f : n_frac == 2 ? 100.0f : 1000.0f;
                        P.q = __fdiv_rn((float)mant, den);
                    }
                }
            }
            // the read's own UMI window (umi_window of smi_worker.hip)
            if (has_bc && has_bc_end && p_x >= 0 && has_q) {
                const int xe = value_end(p_x), x_len = xe - p_x;
                const long pos = five ? bc_end - ae + 3 : ae + 3 - bc_end;
                if (pos >= 1 && pos + umi_len + 1 <= (long)x_len) {   // umi_len + 2 bases: the umi_length-mers at offsets -1 / 0 / +1
                    uint64_t w = 0;
                    for (int k = 0; k < umi_len + 2; k++) {
                        const uint32_t c = five ? ucode4(v.at(p_x + (int)(pos - 1 + k))) : ucomp4(ucode4(v.at(p_x + x_len - (int)(pos + k))));
                        w |= (uint64_t)c << (4 * k);
                    }
                    if (random_umi_seed) {  // assignumis -f: a random window instead (random_umi_window of smi_worker.hip)
                        uint64_t z = random_umi_seed + 0x9E3779B97F4A7C15ull * ((uint64_t)i + 1);
                        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                        z ^= z >> 31;
                        w = 0;
                        for (int k = 0; k < umi_len + 2; k++) w |= (uint64_t)(1u << ((z >> (2 * k)) & 3u)) << (4 * k);
                    }
                    P.win = w;
                    P.flags |= UP_HAS_W;
                }
            }
            // clustering position
            if ((five || has_ps) && !(fl & 4)) {
                const int read_pos = five ? (int)ae + 16 + umi_len + grouping_distance : (int)ps - grouping_distance;
                int p = 0;
                if (ref_position(cigars + cigar_off[i], (int)(cigar_off[i + 1] - cigar_off[i]), pos0[i] + 1, read_pos, &p)) {
                    P.flags |= UP_HAS_POS;
                    P.cpos = p;
                }
            }
            if (nonstd) P.flags |= UP_NONSTD;
        }
    }
    out[i] = P;
}

// ---------------------------------------------------------------------------------------------------------------------------
// grouping: (cell barcode, region) sets of two or more reads, members in input order
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void k_umi_keys(const UmiParsed *__restrict__ parsed, const int32_t *__restrict__ region, int n, int n_done, uint64_t *__restrict__ keys,
                           uint32_t *__restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool elig = i < n_done && (parsed[i].flags & UP_HAS_W) && region[i] >= 0;
    keys[i] = elig ? ((uint64_t)(uint32_t)region[i] << 32) | parsed[i].bc : ~0ull;
    idx[i] = (uint32_t)i;
}

// runs of equal keys -> the groups that are clustered (two or more members): their sizes, squared sizes, pair counts, run starts
__global__ void k_umi_group_sizes(const uint64_t *__restrict__ run_keys, const uint32_t *__restrict__ run_len, const uint32_t *__restrict__ n_runs, uint32_t *__restrict__ gsize,
                                  uint64_t *__restrict__ gpairs, uint64_t *__restrict__ gmat, int cap) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= cap) return;
    const bool ok = r < (int)*n_runs && run_keys[r] != ~0ull && run_len[r] >= 2;  // UmiClustering.lambda$cluster$6: groups of one are dropped
    const uint64_t k = ok ? run_len[r] : 0;
    gsize[r] = (uint32_t)k;
    gpairs[r] = k * (k + 1) / 2;
    gmat[r] = ok ? umi_mat_bytes(k, true) : 0;  // (padded rows for the groups the tiled kernel writes: smi_umi_stage.h)
}

__global__ void k_umi_kept_flags(const uint32_t *__restrict__ gsize, uint32_t *__restrict__ gkept, int cap) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < cap) gkept[r] = gsize[r] ? 1u : 0u;
}
__global__ void k_umi_close_offsets(uint32_t *group_off, uint64_t *pair_off, uint64_t *mat_off, uint32_t n_groups, uint32_t members, uint64_t pairs, uint64_t mat) {
    group_off[n_groups] = members;
    pair_off[n_groups] = pairs;
    mat_off[n_groups] = mat;
}

// compaction of the kept runs into consecutive groups: group_off / pair_off / mat_off and the members' record indices, windows, qualities
__global__ void k_umi_fill_groups(const uint32_t *__restrict__ run_len, const uint32_t *__restrict__ run_start, const uint32_t *__restrict__ gsize,
                                  const uint32_t *__restrict__ goff_run, const uint64_t *__restrict__ poff_run, const uint64_t *__restrict__ moff_run,
                                  const uint32_t *__restrict__ gslot, const uint32_t *__restrict__ n_runs, const uint32_t *__restrict__ sorted_idx,
                                  const UmiParsed *__restrict__ parsed, uint32_t *__restrict__ group_off, uint64_t *__restrict__ pair_off,
                                  uint64_t *__restrict__ mat_off, uint32_t *__restrict__ order, uint64_t *__restrict__ wpk, float *__restrict__ qv) {
    const int r = blockIdx.x;  // one block per run
    if (r >= (int)*n_runs || gsize[r] == 0) return;
    const uint32_t g = gslot[r], o = goff_run[r];
    if (threadIdx.x == 0) {
        group_off[g] = o;
        pair_off[g] = poff_run[r];
        mat_off[g] = moff_run[r];
    }
    for (uint32_t j = threadIdx.x; j < gsize[r]; j += blockDim.x) {
        const uint32_t rec = sorted_idx[run_start[r] + j];
        order[o + j] = rec;
        wpk[o + j] = parsed[rec].win;
        qv[o + j] = parsed[rec].q;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// K-UCLUST: ClusterOneHierarchical for one group per wave (groups of up to kClustMax reads)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kClustMax = kUmiClusterDeviceMax;   // NRECORDS_SWITCH_TO_OWNCLUSTERING: larger groups go to ClusterOne_MyClustering (host)

template <int KMAX>
struct ClustLds {
    static constexpr int TAB = KMAX <= 24 ? 33 : 260;  // fastutil's table never grows beyond 32 slots for up to 24 keys
    uint8_t mat[KMAX * KMAX];    // the group's distance matrix (ed | pos1 << 4 | pos2 << 6), staged once
    uint8_t score[KMAX * KMAX];
    uint16_t id[KMAX * KMAX];
    int16_t nb[KMAX];       // group-local read index of element e
    int16_t label[KMAX];    // slot of the cluster element e belongs to
    int32_t cnum[KMAX];     // cluster number of a slot (elements 0..k-1, merged clusters k, k+1, ...: creation order)
    uint8_t alive[KMAX];
    int16_t csize[KMAX];
    int16_t members[KMAX];  // of the cluster being tagged, ascending
    int16_t ord[KMAX];      // ... in fastutil iteration order
    int32_t tab[TAB], tab2[TAB]; // fastutil open-addressing tables
    uint8_t inside[KMAX];
    uint8_t skipped[KMAX];
};

// iteration order of a fastutil IntOpenHashSet that received `keys` (ascending) one by one: fastutil_order of smi_cluster.hip, serial
__device__ void fastutil_order_dev(const int16_t *keys, int m, int32_t *tab, int32_t *nt, int16_t *out) {
    auto mix = [](uint32_t x) {
        const uint32_t h = x * 0x9E3779B9u;
        return h ^ (h >> 16);
    };
    int n = 32;
    for (int i = 0; i <= n; i++) tab[i] = 0;
    bool zero = false;
    int size = 0;
    for (int q = 0; q < m; q++) {
        const int k = keys[q];
        if (k == 0)
            zero = true;
        else {
            int pos = (int)(mix((uint32_t)k) & (uint32_t)(n - 1));
            while (tab[pos] != 0) pos = (pos + 1) & (n - 1);
            tab[pos] = k;
        }
        const int max_fill = min((int)ceil((double)n * 0.75), n - 1);
        if (size++ >= max_fill) {
            const int need = (int)ceil((double)(size + 1) / 0.75);
            int nn = 2;
            while (nn < need) nn <<= 1;
            for (int i = 0; i <= nn; i++) nt[i] = 0;
            for (int i = n; i-- > 0;) {
                if (tab[i] == 0) continue;
                int pos = (int)(mix((uint32_t)tab[i]) & (uint32_t)(nn - 1));
                while (nt[pos] != 0) pos = (pos + 1) & (nn - 1);
                nt[pos] = tab[i];
            }
            for (int i = 0; i <= nn; i++) tab[i] = nt[i];
            n = nn;
        }
    }
    int o = 0;
    if (zero) out[o++] = 0;
    for (int pos = n; pos-- > 0;)
        if (tab[pos] != 0) out[o++] = (int16_t)tab[pos];
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t w = ((uint64_t)(uint32_t)__shfl_xor((int)(v >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, o);
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// KMAX: largest group this instantiation takes (its LDS is sized for it); WAVES groups per block, one wave each.  Two instantiations run
// over the same group list: <16, 4> takes the groups of up to 16 reads (the bulk: ~1.6 KB of LDS per group, many waves per CU), <100, 1> the rest
template <int KMAX, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_umi_cluster(const uint8_t *__restrict__ dist, const uint64_t *__restrict__ mat_off, const uint32_t *__restrict__ group_off,
                                                            uint32_t n_groups, const float *__restrict__ qv_all, smi_umi_cluster_config cfg, int dev_max, int n_above,
                                                            smi_umi_assignment *__restrict__ out_all, uint8_t *__restrict__ skipped_all, int padded) {
    __shared__ ClustLds<KMAX> L_all[WAVES];
    ClustLds<KMAX> &L = L_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const uint32_t a0 = group_off[g];
    const int n = (int)(group_off[g + 1] - a0);
    if (n <= n_above || (KMAX < kClustMax && n > KMAX)) return;  // the other instantiation's group
    smi_umi_assignment *out = out_all + a0;
    uint8_t *skipped_out = skipped_all + a0;
    for (int i = lane; i < n; i += 64) {
        out[i] = smi_umi_assignment{-1, 0, -1, -1, 0};
        skipped_out[i] = 0;
    }
    if (n < 2 || n > dev_max || n > KMAX) return;  // larger groups stay "not clustered": ClusterOne_MyClustering on the host
    // the matrix into LDS: every distance is read many times (neighbour test, queue, centre, second-best), each of them a dependent
    // global load before this copy existed
    {
        const uint8_t *Mg = dist + mat_off[g];
        const int ld = (int)umi_ld((uint64_t)n, padded != 0);  // (rows of a group above 64 reads may be padded to whole lines)
        if (ld == n) {
            for (int p = lane; p < n * n; p += 64) L.mat[p] = Mg[p];
        } else {
            for (int p = lane; p < n * n; p += 64) L.mat[p] = Mg[(p / n) * ld + p % n];
        }
    }
    wave_sync();
    const uint8_t *M = L.mat;
    auto ed = [&](int i, int j) { return (int)(M[i * n + j] & 15u); };
    const int ced = cfg.complete_link_ed;
    // ---- reads with a neighbour (DistanceMatrix.java:L87-88), ascending -----------------------------------------------------------
    int k = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        bool any = false;
        if (i < n)
            for (int j = 0; j < n && !any; j++) any = i != j && ed(i, j) <= ced;
        const uint64_t m = __ballot(any);
        if (any) L.nb[k + __popcll(m & ((1ull << lane) - 1ull))] = (int16_t)i;
        k += __popcll(m);
    }
    for (int i = lane; i < n; i += 64) L.skipped[i] = 0;
    wave_sync();
    if (k <= 1) return;
    // ---- LingPipe complete link on elements 0..k-1 ------------------------------------------------------------------------------------
    // every pair of live clusters has a score and the id its queue entry was created with; initial entries in row-major order
    for (int p = lane; p < k * k; p += 64) {
        const int x = p / k, y = p - x * k;
        if (x < y) {
            const int s = ed(L.nb[x], L.nb[y]);
            const int id = x * (2 * k - x - 1) / 2 + (y - x - 1);
            L.score[x * k + y] = L.score[y * k + x] = (uint8_t)s;
            L.id[x * k + y] = L.id[y * k + x] = (uint16_t)id;
        }
    }
    for (int e = lane; e < k; e += 64) {
        L.label[e] = (int16_t)e;
        L.cnum[e] = e;
        L.alive[e] = 1;
    }
    wave_sync();
    int next_num = k, next_id = k * (k - 1) / 2, n_alive = k;
    while (n_alive > 1) {
        // the head of the queue: least score, among equals the entry offered LAST (BoundedPriorityQueue$EntryComparator L458-464)
        uint32_t best = 0xFFFFFFFFu;
        for (int p = lane; p < k * k; p += 64) {
            const int x = p / k, y = p - x * k;
            if (x < y && L.alive[x] && L.alive[y]) best = min(best, ((uint32_t)L.score[p] << 16) | (0xFFFFu - (uint32_t)L.id[p]));
        }
        const uint32_t head = wave_min_u32(best);
        if (head == 0xFFFFFFFFu) break;
        const int head_score = (int)(head >> 16), head_id = (int)(0xFFFFu - (head & 0xFFFFu));
        if (head_score > ced) break;  // merge heights never decrease: the clusters of Dendrogram.partitionDistance(ced) are the live ones
        // whose entry that is (ids are unique): found by the lanes again
        int bx = -1, by = -1;
        for (int p = lane; p < k * k; p += 64) {
            const int x = p / k, y = p - x * k;
            if (x < y && L.alive[x] && L.alive[y] && L.id[p] == head_id && L.score[p] == head_score) {
                bx = x;
                by = y;
            }
        }
        const uint64_t who = __ballot(bx >= 0);
        const int src = __builtin_ctzll(who);
        bx = __shfl(bx, src);
        by = __shfl(by, src);
        // PairScore(a, b): initial pairs have a < b (element order); a pair offered after a merge is (merged cluster, other cluster)
        const int numx = L.cnum[bx], numy = L.cnum[by];
        const int d1 = (numx >= k || numy >= k) ? (numx > numy ? bx : by) : (numx < numy ? bx : by);
        const int d2 = d1 == bx ? by : bx;
        // the new entries (d12, d3) are offered in the creation order of d2's live entries (index[d2], L155-166)
        int ns[2] = {0, 0}, nid[2] = {0, 0};
        for (int t = 0; t < 2; t++) {
            const int z = lane + 64 * t;
            if (z < k && L.alive[z] && z != d1 && z != d2) {
                const int s1 = L.score[d1 * k + z], s2 = L.score[d2 * k + z];
                ns[t] = max(s1, s2);
                const int mine = L.id[d2 * k + z];
                int rank = 0;
                for (int w = 0; w < k; w++)
                    if (L.alive[w] && w != d1 && w != d2 && (int)L.id[d2 * k + w] < mine) rank++;
                nid[t] = next_id + rank;
            }
        }
        wave_sync();
        for (int t = 0; t < 2; t++) {
            const int z = lane + 64 * t;
            if (z < k && L.alive[z] && z != d1 && z != d2) {
                L.score[d1 * k + z] = L.score[z * k + d1] = (uint8_t)ns[t];
                L.id[d1 * k + z] = L.id[z * k + d1] = (uint16_t)nid[t];
            }
        }
        for (int e = lane; e < k; e += 64)
            if (L.label[e] == d2) L.label[e] = (int16_t)d1;
        wave_sync();
        if (lane == 0) {
            L.alive[d2] = 0;
            L.cnum[d1] = next_num;
        }
        next_num++;
        next_id += n_alive - 2;
        n_alive--;
        wave_sync();
    }
    // ---- clusters of two or more, the fold-depth filter (ClusterOneHierarchical.java:L101-131) ---------------------------------------
    for (int s = lane; s < k; s += 64) L.csize[s] = 0;
    wave_sync();
    if (lane == 0)
        for (int e = 0; e < k; e++) L.csize[L.label[e]]++;
    wave_sync();
    int mx = 0, n_kept = 0;
    for (int s = 0; s < k; s++)
        if (L.alive[s] && L.csize[s] > 1) mx = max(mx, (int)L.csize[s]);
    for (int s = 0; s < k; s++)
        if (L.alive[s] && L.csize[s] > 1 && (long)L.csize[s] * cfg.fold_depth_below_max > (long)mx) n_kept++;
    for (int e = lane; e < k; e += 64) {
        const int s = L.label[e];
        if (L.csize[s] > 1 && !((long)L.csize[s] * cfg.fold_depth_below_max > (long)mx)) L.skipped[L.nb[e]] = 1;
    }
    wave_sync();
    for (int i = lane; i < n; i += 64) skipped_out[i] = L.skipped[i];
    // ---- centre and tags of every kept cluster --------------------------------------------------------------------------------------
    for (int s = 0; s < k; s++) {
        if (!L.alive[s] || L.csize[s] <= 1 || !((long)L.csize[s] * cfg.fold_depth_below_max > (long)mx)) continue;
        const int m = L.csize[s];
        if (lane == 0) {
            int o = 0;
            for (int e = 0; e < k; e++)
                if (L.label[e] == s) L.members[o++] = L.nb[e];  // ascending (nb is)
            fastutil_order_dev(L.members, m, L.tab, L.tab2, L.ord);
        }
        wave_sync();
        for (int i = lane; i < n; i += 64) L.inside[i] = 0;
        wave_sync();
        for (int j = lane; j < m; j += 64) L.inside[L.members[j]] = 1;
        wave_sync();
        // OneUmiCluster.setClusterCenter* (L49-65): two members -> reads 0 and 1 OF THE GROUP decide; else the least sum of squared distances,
        // the first such member in the set's iteration order
        int center;
        if (m == 2)
            center = qv_all[a0] > qv_all[a0 + 1] ? L.ord[0] : L.ord[1];
        else {
            uint32_t bestc = 0xFFFFFFFFu;
            for (int r = lane; r < m; r += 64) {
                const int sidx = L.ord[r];
                uint32_t tot = 0;
                for (int w = 0; w < m; w++) {
                    const int widx = L.ord[w];
                    if (widx != sidx) tot += (uint32_t)(ed(sidx, widx) * ed(sidx, widx));
                }
                bestc = min(bestc, (tot << 8) | (uint32_t)r);
            }
            center = L.ord[wave_min_u32(bestc) & 0xFFu];
        }
        // tag_members: the centre's 12-mer offset = round(mean of (pos1 - 1) over the other members); U1 = distance to the centre; U2 = least
        // distance to a read outside the cluster when there are several clusters
        int sum = 0;
        for (int j = lane; j < m; j += 64) {
            const int vv = L.members[j];
            if (vv != center) sum += (int)((M[center * n + vv] >> 4) & 3u) - 1;
        }
        sum = wave_sum_i(sum);
        const int offset = (int)floor((double)sum / (double)(m - 1) + 0.5);
        for (int j = lane; j < m; j += 64) {
            const int idx = L.members[j];
            int sec = -1;
            if (n_kept > 1)
                for (int w = 0; w < n; w++)
                    if (!L.inside[w] && (sec < 0 || ed(idx, w) < sec)) sec = ed(idx, w);
            const uint8_t cm = M[center * n + idx];
            out[idx] = smi_umi_assignment{center, (int8_t)offset, (int8_t)(cm & 15u), (int8_t)sec, (int8_t)((cm >> 6) & 3u)};
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// K-UTAG: the per-record values behind U8 / U7 / U1 / U2
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ char udec4(uint32_t c) { return c == 1 ? 'A' : c == 2 ? 'G' : c == 4 ? 'C' : c == 8 ? 'T' : 'N'; }

__global__ void k_umi_tag_base(const UmiParsed *__restrict__ parsed, const int32_t *__restrict__ region, int n, int n_done, int umi_len, smi_umi_tag *__restrict__ tags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    smi_umi_tag t;
    memset(&t, 0, sizeof t);
    t.region = i < n_done ? region[i] : -1;
    t.center = -1;
    t.u1 = t.u2 = -1;
    if (i < n_done) {
        const UmiParsed P = parsed[i];
        if ((P.flags & UP_PRESENT) && (P.flags & UP_HAS_BC)) t.flags |= SMI_UMI_HAS_BC;
        if (P.flags & UP_HAS_W) {
            t.flags |= SMI_UMI_HAS_U7;
            for (int k = 0; k < umi_len; k++) t.u7[k] = udec4((uint32_t)(P.win >> (4 * (k + 1))) & 15u);
        }
    }
    tags[i] = t;
}

__global__ void k_umi_tag_groups(const uint32_t *__restrict__ group_off, uint32_t n_groups, const uint32_t *__restrict__ order, const uint64_t *__restrict__ wpk,
                                 const smi_umi_assignment *__restrict__ asg, const uint8_t *__restrict__ skipped, uint32_t m, int umi_len, smi_umi_tag *__restrict__ tags) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    uint32_t lo = 0, hi = n_groups;  // group of member j
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (group_off[mid] <= j)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t g0 = group_off[lo];
    smi_umi_tag &t = tags[order[j]];
    const smi_umi_assignment a = asg[j];
    if (a.center < 0) {
        if (skipped[j]) t.flags |= SMI_UMI_SKIPPED;
        return;
    }
    t.flags |= SMI_UMI_CLUSTERED;
    t.center = (int32_t)order[g0 + (uint32_t)a.center];
    t.u1 = a.ed;
    t.u2 = a.ed_second;
    const uint64_t cw = wpk[g0 + (uint32_t)a.center];
    for (int k = 0; k < umi_len; k++) t.u8[k] = udec4((uint32_t)(cw >> (4 * (k + 1 + a.offset))) & 15u);
}

// ---------------------------------------------------------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------------------------------------------------------
int launch_umi_parse(smi_ctx *, const char *d_names, const uint32_t *d_name_off, const uint16_t *d_flags, const int32_t *d_pos0, const uint32_t *d_cigars,
                     const uint32_t *d_cigar_off, int n, int five, int grouping_distance, int bc_edit_limit, int umi_len, uint64_t random_umi_seed, UmiParsed *d_out, hipStream_t s) {
    if (!n) return SMI_OK;
    hipLaunchKernelGGL(k_umi_parse, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, d_names, d_name_off, d_flags, d_pos0, d_cigars, d_cigar_off, n, five,
                       grouping_distance, bc_edit_limit, umi_len, random_umi_seed, d_out);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

// sort keys of the region grouping (ReadGrouper sorts the chunk's reads by clustering position, stable in BAM order): biased position in the
// upper half, record number and strand below it; records without a position sort to the end.  Also: how many have a position, whether any
// name needs the host parser, and one bit per record "has a position" (the host asks for ranks among those in one rare case).
__global__ __launch_bounds__(256) void k_umi_region_keys(const UmiParsed *__restrict__ P, int n, uint64_t *__restrict__ keys, uint32_t *__restrict__ counters,
                                                         uint64_t *__restrict__ has_bits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t f = i < n ? P[i].flags : 0u;
    const bool has = (f & UP_HAS_POS) != 0;
    if (i < n)
        keys[i] = has ? ((uint64_t)((uint32_t)P[i].cpos ^ 0x80000000u) << 32) | ((uint64_t)(uint32_t)i << 1) | ((f & UP_REV) ? 1u : 0u) : ~0ull;
    const uint64_t m = __ballot(has);
    const uint64_t bad = __ballot((f & (UP_NONSTD | UP_ERROR)) != 0);
    if ((threadIdx.x & 63) == 0) {
        if (i < n) has_bits[i >> 6] = m;
        if (m) atomicAdd(&counters[0], (uint32_t)__popcll(m));
        if (bad) atomicOr(&counters[1], 1u);
    }
}

// -> B.keys_sorted[0 .. n): the keys in ascending order; d_counters[0] = records with a position, [1] = some name needs the host parser
int launch_umi_region_keys(smi_ctx *, const UmiParsed *d_parsed, int n, UmiGroupBuffers &B, uint32_t *d_counters, uint64_t *d_has_bits, hipStream_t s) {
    if (!n) return SMI_OK;
    SMI_HIP(hipMemsetAsync(d_counters, 0, 8, s));
    hipLaunchKernelGGL(k_umi_region_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_parsed, n, B.keys, d_counters, d_has_bits);
    size_t tmp = B.tmp_bytes;
    SMI_HIP(hipcub::DeviceRadixSort::SortKeys(B.tmp, tmp, B.keys, B.keys_sorted, n, 0, 64, s));
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

size_t umi_group_scratch_bytes(int n) {
    size_t sort_tmp = 0, rle_tmp = 0, scan32 = 0, scan64 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, n);
    (void)hipcub::DeviceRunLengthEncode::Encode(nullptr, rle_tmp, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan32, (uint32_t *)nullptr, (uint32_t *)nullptr, n + 1);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan64, (uint64_t *)nullptr, (uint64_t *)nullptr, n + 1);
    return std::max(std::max(sort_tmp, rle_tmp), std::max(scan32, scan64)) + 256;
}

// keys -> sorted -> runs -> kept groups.  Device buffers (capacities in comments) come from the caller's arena; totals[0..3] = groups, members,
// pairs, matrix bytes (host; the stream is synchronised once to read them)
int launch_umi_groups(smi_ctx *, const UmiParsed *d_parsed, const int32_t *d_region, int n, int n_done, UmiGroupBuffers &B, uint64_t *totals, hipStream_t s) {
    totals[0] = totals[1] = totals[2] = totals[3] = 0;
    if (!n) return SMI_OK;
    const unsigned gb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_umi_keys, dim3(gb), dim3(256), 0, s, d_parsed, d_region, n, n_done, B.keys, B.idx);
    size_t tmp = B.tmp_bytes;
    SMI_HIP(hipcub::DeviceRadixSort::SortPairs(B.tmp, tmp, B.keys, B.keys_sorted, B.idx, B.idx_sorted, n, 0, 64, s));
    tmp = B.tmp_bytes;
    SMI_HIP(hipcub::DeviceRunLengthEncode::Encode(B.tmp, tmp, B.keys_sorted, B.run_keys, B.run_len, B.n_runs, n, s));
    // run starts (exclusive scan of the run lengths; entries behind n_runs are garbage and unused), group sizes of the kept runs
    SMI_HIP(hipMemsetAsync(B.run_len + n, 0, 4, s));
    tmp = B.tmp_bytes;
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(B.tmp, tmp, B.run_len, B.run_start, n + 1, s));
    hipLaunchKernelGGL(k_umi_group_sizes, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, s, B.run_keys, B.run_len, B.n_runs, B.gsize, B.gpairs, B.gmat, n + 1);
    // kept runs -> consecutive group slots; offsets of members / pairs / matrices per run
    hipLaunchKernelGGL(k_umi_kept_flags, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, s, B.gsize, B.gkept, n + 1);
    tmp = B.tmp_bytes;
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(B.tmp, tmp, B.gkept, B.gslot, n + 1, s));
    tmp = B.tmp_bytes;
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(B.tmp, tmp, B.gsize, B.goff_run, n + 1, s));
    tmp = B.tmp_bytes;
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(B.tmp, tmp, B.gpairs, B.poff_run, n + 1, s));
    tmp = B.tmp_bytes;
    SMI_HIP(hipcub::DeviceScan::ExclusiveSum(B.tmp, tmp, B.gmat, B.moff_run, n + 1, s));
    uint32_t h_groups = 0, h_members = 0;
    uint64_t h_pairs = 0, h_mat = 0;
    SMI_HIP(hipMemcpyAsync(&h_groups, B.gslot + n, 4, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(&h_members, B.goff_run + n, 4, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(&h_pairs, B.poff_run + n, 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipMemcpyAsync(&h_mat, B.moff_run + n, 8, hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    totals[0] = h_groups;
    totals[1] = h_members;
    totals[2] = h_pairs;
    totals[3] = h_mat;
    if (!h_groups) return SMI_OK;
    hipLaunchKernelGGL(k_umi_fill_groups, dim3((unsigned)n), dim3(64), 0, s, B.run_len, B.run_start, B.gsize, B.goff_run, B.poff_run, B.moff_run, B.gslot, B.n_runs,
                       B.idx_sorted, d_parsed, B.group_off, B.pair_off, B.mat_off, B.order, B.wpk, B.qv);
    hipLaunchKernelGGL(k_umi_close_offsets, dim3(1), dim3(1), 0, s, B.group_off, B.pair_off, B.mat_off, h_groups, h_members, h_pairs, h_mat);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

int launch_umi_cluster(smi_ctx *, const uint8_t *d_dist, const uint64_t *d_mat_off, const uint32_t *d_group_off, uint32_t n_groups, const float *d_qv,
                       const smi_umi_cluster_config &cfg, int dev_max, smi_umi_assignment *d_asg, uint8_t *d_skipped, hipStream_t s, bool padded) {
    if (!n_groups) return SMI_OK;
    hipLaunchKernelGGL((k_umi_cluster<16, 4>), dim3((n_groups + 3) / 4), dim3(256), 0, s, d_dist, d_mat_off, d_group_off, n_groups, d_qv, cfg, dev_max, 0, d_asg, d_skipped, padded ? 1 : 0);
    hipLaunchKernelGGL((k_umi_cluster<kClustMax, 1>), dim3(n_groups), dim3(64), 0, s, d_dist, d_mat_off, d_group_off, n_groups, d_qv, cfg, dev_max, 16, d_asg,
                       d_skipped, padded ? 1 : 0);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

int launch_umi_tags(smi_ctx *, const UmiParsed *d_parsed, const int32_t *d_region, int n, int n_done, const UmiGroupBuffers &B, uint32_t n_groups, uint32_t m,
                    const smi_umi_assignment *d_asg, const uint8_t *d_skipped, int umi_len, smi_umi_tag *d_tags, hipStream_t s) {
    if (!n) return SMI_OK;
    hipLaunchKernelGGL(k_umi_tag_base, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_parsed, d_region, n, n_done, umi_len, d_tags);
    if (m) hipLaunchKernelGGL(k_umi_tag_groups, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, B.group_off, n_groups, B.order, B.wpk, d_asg, d_skipped, m, umi_len, d_tags);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

}  // namespace smi
