// smi_internal.h -- shared declarations of libsicelore_mi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "sicelore_mi.h"

namespace smi {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what);

// smi_inflate_host.hip: the host's DEFLATE decoder and CRC-32
uint32_t host_crc32(uint32_t crc, const uint8_t *p, size_t n);                                   // zlib's crc32()
int host_inflate_exact(const uint8_t *in, size_t n_in, uint8_t *out, size_t n_out);               // 0 = the stream filled out[0 .. n_out)
// members of a gzip stream from *in_pos on, appended at *out_pos while they fit in cap: SMI_OK = input used up, 1 = the member at *in_pos
// needs more room than cap - *out_pos (both positions stay at that member), SMI_ERR_INVALID = malformed (error text set)
int host_gunzip(const uint8_t *in, size_t n_in, size_t *in_pos, uint8_t *out, size_t cap, size_t *out_pos);

#define SMI_HIP(call)                                        \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) return smi::hip_fail(e__, #call); \
    } while (0)

// Membership pyramid over the 2^32 universe of 16-mers (A=0 G=1 C=2 T=3, first base most significant).
//   l0  : 1 bit per 2^G0 consecutive keys  (2^(32-G0) bits = 4 MiB at G0 = 7; the size was chosen by measurement:
//         a finer top level lets fewer of the 620 mutants through to the levels below)
//   l1  : 2^(32-G1) bits, Infinity-Cache resident: word = key >> (G1 + 5) (1024 consecutive keys), bit = a 5-bit hash of the
//         key's low 10 bits.  A probe that passed the top level has a barcode within 128 keys of it; with one bit per 32
//         CONSECUTIVE keys that neighbour shared the probe's bit one time in four (27 % of the survivors went on to the
//         exact level), hashed it does so one time in 32 (5.8 %)
//   fine: 1 bit per key                    (2^32 bits = 512 MiB, HBM)
// A probe walks l0 -> l1 -> fine and stops at the first clear bit, so it is exact; consecutive keys share
// a bit, so the mutants of one window that keep its leading bases hit the same 128-B line.
//   l0s : a second top level with the key's LAST 7 bases (and its first bit) selecting the 128-B line and its first
//         5.5 bases the bit (the bases between are dropped), for the mutants that change the leading bases: all substitutions at positions
//         0..6 share the window's line, all early insertions share the "shifted right" line and all early
//         deletions the "shifted left" line -- 4 lines per offset instead of ~54 with one prefix-ordered table.
// rank[k] = number of set keys below (k << 8): ordinal(key) = rank[key >> 8] + popcount of the fine bits
// of that 256-key block below key  (pass-1 histogram index).
constexpr int kG0 = 7;  // 4 MiB per top level: measured best (8: 9.6 ms, 7: 8.3 ms, 6: 9.6 ms per 10 M reads)
constexpr int kG1 = 5;
constexpr size_t kL0Words = (size_t(1) << (32 - kG0)) / 32;
constexpr size_t kL1Words = (size_t(1) << (32 - kG1)) / 32;
constexpr size_t kFineWords = (size_t(1) << 32) / 32;
constexpr size_t kRankEntries = size_t(1) << 24;

// Hand-off of LDS data between lanes of ONE wavefront: LDS operations of a wave execute in order, so all that is
// needed is that the compiler neither reorders memory operations across this point nor leaves them pending.
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// The reference's 4-bit base codes (NucleicAcidByteCodeBase: A 1, G 2, C 4, T 8; anything else counts as N = 15 here) of FOUR ASCII
// characters at once, one code per byte.  (c >> 1) & 7 separates the upper-cased letters A C T G (0 1 2 3), so one v_perm_b32 against a
// 4-byte table gives the codes (COMPLEMENT: those of the complementary bases) and a second one against the letters themselves proves that
// each character was one of the four; every other byte becomes 15.  14 VALU operations per four bases instead of 9 per base.
template <bool COMPLEMENT>
__device__ __forceinline__ uint32_t enc4x4(uint32_t w) {
    const uint32_t u = w & 0xDFDFDFDFu;
    const uint32_t idx = (u >> 1) & 0x07070707u;
    const uint32_t letters = 'A' | ('C' << 8) | ('T' << 16) | ((uint32_t)'G' << 24);
    const uint32_t codes = COMPLEMENT ? (8u | (2u << 8) | (1u << 16) | (4u << 24)) : (1u | (4u << 8) | (8u << 16) | (2u << 24));
    const uint32_t t = __builtin_amdgcn_perm(0u, letters, idx) ^ u;                   // zero byte = one of the four letters
    const uint32_t nz = (((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;        // 0x80 in every other byte
    return __builtin_amdgcn_perm(0x0F0F0F0Fu, codes, idx) | (nz - (nz >> 7));         // ... which becomes 0x7F: all four code bits set
}
// bit c of each of the four code bytes -> four consecutive bits (byte 0 first)
__device__ __forceinline__ uint32_t plane_nibble(uint32_t code4x4, int c) {
    const uint32_t x = code4x4 & (0x01010101u << c);
    // y = x | x << 7 | x << 14 | x << 21: bits 21+c .. 24+c = bytes 0 .. 3.  As inline asm: written in C the compiler recognises a
    // multiplication by 0x204081 and emits v_mul_lo_u32, which issues at a quarter of the rate of these three
    uint32_t y;
    asm("v_lshl_or_b32 %0, %1, 7, %1" : "=v"(y) : "v"(x));
    asm("v_lshl_or_b32 %0, %1, 14, %2" : "=v"(y) : "v"(x), "v"(y));
    asm("v_lshl_or_b32 %0, %1, 21, %2" : "=v"(y) : "v"(x), "v"(y));
    return (y >> (21 + c)) & 0xFu;
}

// Two-stage top level of K-BC1 (`t2`, 16 MiB): the words of l0 | l0s interleaved with a second word each.  Entry j (8 B) =
// { word j of l0 | l0s , stage-2 word j }: the stage-2 bit of a key sits at the same word position as its stage-1 bit and is
// chosen by five OTHER bits of the key (prefix table: the low five, which the 128-key cell ignores; twin: bits 14..18, which
// the twin drops).  A word position covers 4096 keys = 3.4 barcodes, so a probe that passes stage 1 by chance passes stage 2
// one time in ten -- the selectivity of a separate level, from the same 8-byte load.
__host__ __device__ inline uint32_t t2_prefix_bit(uint32_t key) { return key & 31u; }
__host__ __device__ inline uint32_t t2_twin_bit(uint32_t key) { return (key >> 14) & 31u; }

// position of a key in l1
__host__ __device__ inline uint32_t l1_word(uint32_t key) { return key >> (kG1 + 5); }
__host__ __device__ inline uint32_t l1_bit(uint32_t key) { return (key ^ (key >> 5)) & 31u; }

struct Pyramid {
    const uint32_t *l0;
    const uint32_t *l0s;  // suffix-major twin of l0 (same 2^24 bits): bit ((key & 0x3FFF) << 10 | key >> 22)
    const uint32_t *l1;
    const uint32_t *fine;
    const uint32_t *rank;
    const uint32_t *t2;  // two-stage top level of K-BC1: entry j = {word j of l0 | l0s, stage-2 word j}
    const uint32_t *n1;  // K-BC2, short used lists only (else null): the sequences one mutation step away FROM which a barcode can be reached (l1 layout)
    const uint32_t *n2;  // ... from which TWO OR MORE different barcodes can be reached (same layout)
    const uint32_t *n1s, *n2s;  // the same two filters indexed by the last nine bases first (smi_bc.hip n1_cell_s), or null
    const uint32_t *nb2; // K-BC2's offset filter for short used lists (else null): 1 bit per key, set for every sequence TWO steps away from a barcode
    const uint32_t *nb;  // K-BC1's offset filter (else null): 1 bit per key, set for every sequence one mutation step away from a barcode (512 MiB)
    const uint32_t *nb5; // the same filter laid out by the 12 bases the five offsets' windows share (else null; smi_bc.hip "nb5"): the five bits of a read in 160 consecutive bytes
    const uint64_t *nt;  // K-BC1's neighbourhood table (else null): open addressing, entry = 1 << 40 | sequence << 8 | the step that leads from it to a barcode
    uint32_t nt_cap;     // its slots
};

}  // namespace smi

struct smi_ctx {
    int device = -1;
    uint32_t *l0 = nullptr;
    uint32_t *l0s = nullptr;
    uint32_t *l1 = nullptr;
    uint32_t *t2 = nullptr;
    uint32_t *n1 = nullptr;   // allocated with the first short barcode list (2 x 16 MiB: n1, then n2)
    bool n1_valid = false;    // describes the set that is loaded now
    bool n1s_valid = false;   // ... and the suffix-major copies behind n1 / n2 are built
    int polya_len = 0, polya_window = 0;  // smi_ctx_set_polya: the chunk workers' polyA finder parameters (0: the shipped config.xml values)
    float polya_frac = 0.0f;
    smi_run_knobs knobs = {};  // smi_ctx_set_knobs: config.xml's knobs for the chunk workers of this context (knobs_set false: the shipped file)
    bool knobs_set = false;
    uint64_t random_bc_seed = 0;  // smi_ctx_set_random_barcodes (scanfastq -e): pass 2 matches random windows
    bool nb2_valid = false;   // the build scratch n1_owner holds the two-step neighbourhood bitmap of the set that is loaded now
    uint32_t *nb = nullptr;   // allocated with the first barcode set (512 MiB)
    uint32_t *nb5 = nullptr;  // allocated with the first neighbourhood table (2.5 GiB); nb5_valid: describes the set that is loaded now
    bool nb5_valid = false;
    uint32_t *n1_owner = nullptr;  // scratch of the n2 build (one u32 per n1 cell, 512 MiB), kept: hipMalloc / hipFree of that size cost ~100 ms per set load
    bool nb_valid = false;
    uint64_t *nt = nullptr;   // grown on demand: 1.6 slots of 8 bytes per (barcode, step) pair, 7.8 GB for the 3.6 M list
    size_t nt_alloc = 0;      // slots allocated
    bool nt_small_only = false;  // the 3-slot table did not fit this device: lists of that size keep the 1.6-slot one (no retry per load)
    uint32_t nt_cap = 0;      // slots in use by the set that is loaded now (0: no table)
    uint8_t *bc_codes = nullptr;  // K-BC1, table path: one byte per (read, offset) between its two kernels (grow-only, private to a context or lane)
    size_t bc_codes_bytes = 0;
    uint32_t *fine = nullptr;
    uint32_t *rank = nullptr;
    uint32_t *block_counts = nullptr;  // scratch for the rank scan
    size_t n_keys = 0;                 // distinct keys loaded
    uint64_t set_build_us = 0;         // wall time of the last launch_build_pyramid (smi_set_stats)
    int set_mode = -1;
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // around the last timed kernel
    bool ev_valid = false;
    hipEvent_t kev[SMI_K_COUNT][2] = {};       // per kernel id (smi_kernel_ms)
    bool kev_valid[SMI_K_COUNT] = {};
    // staging buffers of the *_batch entry points (grown on demand)
    void *stage_in = nullptr;
    void *stage_out = nullptr;
    size_t stage_in_bytes = 0, stage_out_bytes = 0;
    hipStream_t stream = nullptr;  // private stream of the *_batch entry points
    hipStream_t side_stream = nullptr;  // created on first use: independent kernels of one call run beside the caller's stream (K-WNAME || K-WRITE)
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    void *scan_tmp = nullptr;      // scratch of the FASTQ indexer (block counts + hipcub temp storage)
    size_t scan_tmp_bytes = 0;
    const smi_ctx *set_owner = nullptr;  // worker lane (smi_ctx_create_lane): the pyramid pointers above belong to this context
    uint32_t *chim_list = nullptr; // K-CHIM: queue of the reads the filter pass could not clear (+ its counter), grow-only
    size_t chim_list_bytes = 0;
    void *chim_slots = nullptr;    // K-CHIM: matches handed from the exact TSO scan to the rules kernel
    size_t chim_slot_bytes = 0;
    void *chim_work = nullptr;     // K-CHIM second generation: per-read heads, the global queue of positions to align, their error counts (grow-only)
    size_t chim_work_bytes = 0;
    void *umi_own = nullptr;       // ClusterOne_MyClustering on the device: index / count / sum scratch of one large group (grow-only)
    size_t umi_own_bytes = 0;
    const uint8_t *fq_swept_text = nullptr;  // launch_fastq_sweep's text while its flags are valid in scan_tmp (consumed by the next launch_fastq_index)
    size_t fq_swept_bytes = 0;
    void *pin_words = nullptr;     // 4 KiB of page-locked host memory for read-backs of a few words (pin_words() in smi_ctx.hip)
    void *umi_plan = nullptr;      // K-UMI: per group its pairs in the flat kernel / tiles in the tiled one, and their prefix sums (grow-only)
    size_t umi_plan_bytes = 0;
    void *chim_flat = nullptr;     // K-CHIM-A second generation: owner of every plane word, gate / bound / trigger words, verdicts (grow-only)
    size_t chim_flat_bytes = 0;
    void *arena = nullptr;         // device memory of the chunk workers (smi_worker.hip), grow-only
    size_t arena_bytes = 0;
    uint8_t *host_out[2] = {nullptr, nullptr};  // pinned: passed / failed text of the last smi_scanfastq_pass2_chunk
    size_t host_out_bytes[2] = {0, 0};
    std::vector<smi_scan_result> host_scan;  // its per-record results when asked for
    std::vector<smi_bc_result> host_bc;
    // packed chunk workers: page-locked, grow-only host buffers (index, offsets, planes, quality tails / sums, decisions)
    enum { HB_RECS = 0, HB_OFFS, HB_PSTART, HB_PLANES, HB_QTAIL, HB_QSUM, HB_CHIM, HB_FOFFS, HB_FSRC, HB_SCAN, HB_BC, HB_RANK, HB_RKEYS, HB_RBITS, HB_REGION, HB_OWN, HB_COUNT };
    smi_scan_stats host_stats = {};  // scan statistics of the last packed pass-2 chunk (want_results)
    void *umi_dist = nullptr;      // K-UMI matrices of the device UMI stage (smi_assignumis_chunk), grow-only
    size_t umi_dist_bytes = 0;
    void *region_work = nullptr;      // host scratch of the region grouping (smi_cluster.hip), kept between calls
    void *deflate_scratch = nullptr;  // K-DEFLATE: block slots, sizes, offsets, scan storage (grow-only)
    size_t deflate_scratch_bytes = 0;
    void *host_buf[HB_COUNT] = {};
    size_t host_buf_bytes[HB_COUNT] = {};
};

namespace smi {
// work forked onto a context's side stream: unless disarmed (the join is in place), leaving the scope waits for the side stream -- an error return between a
// fork and its join does not leave kernels running on buffers the caller is about to reuse
struct SideStreamGuard {
    hipStream_t side;
    bool armed;
    ~SideStreamGuard() {
        if (armed) (void)hipStreamSynchronize(side);
    }
};
// the configurations a chunk worker of `ctx` runs with: its knobs (smi_ctx_set_knobs), then -p / -f / -w (smi_ctx_set_polya); the splitter's
// strings point into the context (smi_ctx.hip)
int worker_scan_config(const smi_ctx *ctx, int pass, int five_prime, int dont_search_polya, smi_scan_config *sc);
int worker_chimera_config(const smi_ctx *ctx, int five_prime, smi_chimera_config *cc);
// ClusterOne_MyClustering of one group above 100 reads on the matrix in HBM (smi_cluster.hip)
int umi_cluster_own_device(smi_ctx *ctx, const uint8_t *d_mat, int n, const float *d_qv, const smi_umi_cluster_config &cfg, smi_umi_assignment *d_out,
                           uint8_t *d_skipped, hipStream_t s, int ld = 0);  // ld: row stride of d_mat (0: n)
int time_begin(smi_ctx *ctx, int kid, hipStream_t s);
int time_end(smi_ctx *ctx, int kid, hipStream_t s);
Pyramid pyramid_of(const smi_ctx *ctx);
int launch_bc_match(smi_ctx *ctx, const smi_bc_window *d_win, size_t n, int max_ed, int five_prime,
                    smi_bc_result *d_out, hipStream_t s);
int launch_extract_windows(smi_ctx *ctx, const uint8_t *d_reads, const uint64_t *d_offsets, const int32_t *d_ae,
                           size_t n, int five_prime, smi_bc_window *d_win, hipStream_t s);
int launch_build_pyramid(smi_ctx *ctx, const uint32_t *d_keys, size_t n, hipStream_t s, bool membership_only = false);
int launch_hist(smi_ctx *ctx, const uint32_t *d_keys, const uint8_t *d_pass, size_t n, uint32_t *d_hist,
                hipStream_t s);
int launch_set_digests(smi_ctx *ctx, uint64_t *out5, hipStream_t s);  // bits of nb / nb5, digest of nb5, slots / entries of nt
int launch_bc_counts(smi_ctx *ctx, const smi_bc_result *d_res, size_t n, uint32_t *d_counts, hipStream_t s);
int launch_hist_windows(smi_ctx *ctx, const smi_bc_window *d_win, const smi_scan_result *d_scan, size_t n,
                        uint32_t *d_hist, hipStream_t s);
int launch_keys_windows(smi_ctx *ctx, const smi_bc_window *d_win, const smi_scan_result *d_scan, size_t n, uint64_t *d_keys, size_t cap,
                        unsigned long long *d_count, hipStream_t s);
int launch_count_keys(smi_ctx *ctx, const uint64_t *d_keys, size_t n, uint64_t *d_unique, uint32_t *d_counts, uint64_t *d_n_unique, hipStream_t s);
int launch_scan(smi_ctx *ctx, const uint32_t *d_ends, const int32_t *d_len, const uint8_t *d_qtail,
                const uint32_t *d_qsum, size_t n, const smi_scan_config *cfg, smi_scan_result *d_out,
                smi_bc_window *d_win, hipStream_t s);
int launch_umi_dist(smi_ctx *ctx, const uint64_t *d_windows, const uint32_t *d_group_off, const uint64_t *d_pair_off,
                    const uint64_t *d_mat_off, uint32_t n_groups, uint64_t total_pairs, uint8_t *d_out, hipStream_t s, int umi_len = 12, bool padded = false);
// umis/umi_length of a context's knobs (12 without knobs)
inline int ctx_umi_length(const smi_ctx *ctx) { return ctx->knobs_set ? ctx->knobs.umi_length : 12; }
int launch_pack_ends(smi_ctx *ctx, const uint8_t *d_reads, const uint8_t *d_quals, const uint64_t *d_offsets, const uint64_t *d_starts,
                     size_t n, int head_quals, uint32_t *d_ends, int32_t *d_len, uint8_t *d_qtail, uint32_t *d_qsum, hipStream_t s);
int launch_count_lines(smi_ctx *ctx, const uint8_t *d_text, size_t n_bytes, size_t *n_lines, hipStream_t s);
int launch_frag_text_starts(smi_ctx *ctx, const uint64_t *d_seq_start, const uint64_t *d_qual_start, const uint64_t *d_offsets,
                            const uint64_t *d_frag_offsets, const uint32_t *d_frag_src, size_t m, uint64_t *d_bstart, uint64_t *d_qstart,
                            hipStream_t s);
size_t read_planes_stride(uint64_t total_bases, size_t n);
int launch_pack_reads(smi_ctx *ctx, const uint8_t *d_reads, const uint64_t *d_offsets, const uint64_t *d_starts, size_t n,
                      uint64_t total_bases, uint32_t *d_planes, hipStream_t s);
// d_pstart != nullptr: read r starts at word d_pstart[r] of each plane and the planes are stride_override words apart (segmented host packer)
int launch_chimera(smi_ctx *ctx, const uint32_t *d_planes, const uint64_t *d_offsets, size_t n, uint64_t total_bases,
                   const smi_chimera_config *cfg, smi_chimera_result *d_out, hipStream_t s, const uint32_t *d_pstart = nullptr,
                   size_t stride_override = 0);
int launch_split_offsets(smi_ctx *ctx, const smi_chimera_result *d_chim, const uint64_t *d_offsets, size_t n,
                         uint32_t *d_scratch, uint64_t *d_total, uint64_t *d_frag_offsets, uint32_t *d_frag_src,
                         hipStream_t s);
int launch_fastq_sweep(smi_ctx *ctx, const uint8_t *d_text, size_t n_bytes, size_t *n_lines, hipStream_t s);  // first half of the index + line count
int launch_fastq_index(smi_ctx *ctx, const uint8_t *d_text, size_t n_bytes, uint64_t *d_line_start, size_t cap_lines,
                       uint64_t *d_name_start, uint32_t *d_name_len, uint64_t *d_seq_start, uint32_t *d_seq_len,
                       uint64_t *d_qual_start, uint64_t *d_offsets, size_t cap_records, size_t *n_records, uint32_t *errors,
                       hipStream_t s, uint64_t *total_bases = nullptr);  // total_bases: d_offsets[n_records], read back on the same wait
int launch_fastq_gather(smi_ctx *ctx, const uint8_t *d_text, const uint64_t *d_start, const uint64_t *d_offsets, size_t n,
                        uint8_t *d_out, hipStream_t s);
// read planes (K-PACKR / smi_pack_reads_host): [4][stride] u32; read r starts at word plane_start(offsets[r], r) of each plane and owns
// ceil(len / 32) data words + 4 zero words (gates and windows run past the end)
constexpr int kReadPadWords = 5;
__host__ __device__ inline size_t plane_start(uint64_t base_offset, size_t r) { return (size_t)(base_offset >> 5) + kReadPadWords * r; }
int launch_ends_from_planes(smi_ctx *ctx, const uint32_t *d_planes, size_t stride, const uint64_t *d_read_offsets, const uint64_t *d_rec_offsets,
                            const uint32_t *d_frag_src, size_t m, uint32_t *d_ends, int32_t *d_len, hipStream_t s, const uint32_t *d_pstart = nullptr);
int ensure_host_buf(smi_ctx *ctx, int which, size_t bytes);
// 4 KiB of page-locked host memory of the context for reading a few words back (a copy into pageable memory is staged by the runtime and
// blocks for about twice as long); valid until the next call that uses it, i.e. read it before calling on -- nullptr if the allocation failed
void *pin_words(smi_ctx *ctx);
// smi_region_group from keys sorted on the device (smi_cluster.hip)
int region_group_from_sorted(void **work, const uint64_t *keys, size_t n_pos, int32_t n, const uint64_t *has_bits, int32_t max_dist, int keep_data_end,
                             int32_t *region, int32_t *n_done);  // *work: scratch kept between calls (ctx->region_work)
void region_work_free(void *work);
int region_group_from_sorted(void **work, const uint64_t *keys, size_t n_pos, int32_t n, const uint64_t *has_bits, int32_t max_dist, int keep_data_end,
                             int32_t *region, int32_t *n_done);  // *work: scratch kept between calls (ctx->region_work)
void region_work_free(void *work);
int region_group_strided(const void *recs, size_t stride, size_t pos_off, size_t flags_off, uint32_t has_pos_bit, uint32_t rev_bit, int32_t n,
                         int32_t max_dist, int keep_data_end, int32_t *region, int32_t *n_done);
// K-DEFLATE (smi_deflate.hip): d_in -> one gzip member / raw deflate stream in d_out; d_total[0] = its size, d_total[1] = error flags (device)
size_t deflate_bound(size_t n_bytes);
size_t deflate_scratch_bytes(size_t n_bytes);
int launch_deflate(smi_ctx *ctx, const uint8_t *d_in, size_t n_bytes, uint8_t *d_out, size_t out_cap, uint8_t *d_scratch, uint64_t *d_total, int gzip,
                   hipStream_t s);
int ensure_deflate_scratch(smi_ctx *ctx, size_t n_bytes, size_t extra_bytes = 0);
// two buffers -> two gzip members in the context's deflate scratch (behind the block slots); d_totals: [size a, flags a, size b, flags b]
int deflate_pair(smi_ctx *ctx, const uint8_t *d_a, size_t na, const uint8_t *d_b, size_t nb, uint8_t **d_za, uint8_t **d_zb, uint64_t **d_totals,
                 hipStream_t s);
}  // namespace smi
