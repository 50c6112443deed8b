// smi_umi_stage.h -- shared between smi_umi_stage.hip (kernels) and smi_worker.hip (smi_assignumis_chunk)
#pragma once
#include "smi_internal.h"

namespace smi {

enum : uint32_t { UP_PRESENT = 1u, UP_HAS_BC = 2u, UP_HAS_W = 4u, UP_HAS_POS = 8u, UP_REV = 16u, UP_NONSTD = 32u, UP_ERROR = 64u };
struct UmiParsed {   // K-UPARSE's record: what FastqRecordExt.getScanDatFromReadName + generateReadScanData yield for the UMI step
    uint64_t win;    // the read's 14-base UMI window, 4-bit codes (UP_HAS_W)
    uint32_t bc;     // cell barcode, 2 bits per base (grouping key; UP_HAS_BC)
    int32_t cpos;    // clustering position (UP_HAS_POS)
    float q;         // Q= of the name
    uint32_t flags;  // UP_*; UP_NONSTD: something the device parser does not evaluate itself -> the chunk takes the host path
};
static_assert(sizeof(UmiParsed) == 24, "UmiParsed layout");

struct UmiGroupBuffers {  // device scratch of the grouping step; capacities for a chunk of n records
    uint64_t *keys, *keys_sorted, *run_keys;        // n, n, n + 1
    uint32_t *idx, *idx_sorted, *run_len, *run_start, *n_runs;  // n, n, n + 2, n + 2, 1
    uint32_t *gsize, *gkept, *gslot, *goff_run;     // n + 2 each
    uint64_t *gpairs, *gmat, *poff_run, *moff_run;  // n + 2 each
    uint32_t *group_off, *order;                    // n / 2 + 2, n
    uint64_t *pair_off, *mat_off, *wpk;             // n / 2 + 2, n / 2 + 2, n
    float *qv;                                      // n
    void *tmp;
    size_t tmp_bytes;
};

int launch_umi_parse(smi_ctx *ctx, const char *d_names, const uint32_t *d_name_off, const uint16_t *d_flags, const int32_t *d_pos0, const uint32_t *d_cigars,
                     const uint32_t *d_cigar_off, int n, int five, int grouping_distance, int bc_edit_limit, int umi_len, uint64_t random_umi_seed, UmiParsed *d_out, hipStream_t s);
size_t umi_group_scratch_bytes(int n);
int launch_umi_region_keys(smi_ctx *ctx, const UmiParsed *d_parsed, int n, UmiGroupBuffers &B, uint32_t *d_counters, uint64_t *d_has_bits, hipStream_t s);
int launch_umi_groups(smi_ctx *ctx, const UmiParsed *d_parsed, const int32_t *d_region, int n, int n_done, UmiGroupBuffers &B, uint64_t *totals, hipStream_t s);
// groups of up to dev_max reads are clustered (ClusterOneHierarchical); larger ones are left untouched (all "not clustered") for the host
int launch_umi_cluster(smi_ctx *ctx, const uint8_t *d_dist, const uint64_t *d_mat_off, const uint32_t *d_group_off, uint32_t n_groups, const float *d_qv,
                       const smi_umi_cluster_config &cfg, int dev_max, smi_umi_assignment *d_asg, uint8_t *d_skipped, hipStream_t s, bool padded = false);
int launch_umi_tags(smi_ctx *ctx, const UmiParsed *d_parsed, const int32_t *d_region, int n, int n_done, const UmiGroupBuffers &B, uint32_t n_groups, uint32_t m,
                    const smi_umi_assignment *d_asg, const uint8_t *d_skipped, int umi_len, smi_umi_tag *d_tags, hipStream_t s);
constexpr int kUmiClusterDeviceMax = 100;
// Row stride and size of a group's distance matrix.  Dense (the public contract of smi_umi_dist_device): n x n bytes.  Padded (round 6, the chunk worker's
// own matrices): groups above 64 reads -- the ones the tiled kernel writes -- have rows of ld = n rounded up to 64 bytes and start on a 64-byte boundary, so
// every 64-byte row piece of a tile is one whole line: a dense row starts at any byte, the first and last line of a piece were shared with the tile beside
// it, written twice, and K-UMI's stores came to 1.55 x the matrix bytes (profiles/r05/umi_pmc.json).
__host__ __device__ inline uint64_t umi_ld(uint64_t n, bool padded) { return padded && n > 64 ? (n + 63) & ~(uint64_t)63 : n; }
__host__ __device__ inline uint64_t umi_mat_bytes(uint64_t n, bool padded) { return padded ? (umi_ld(n, true) * n + 63) & ~(uint64_t)63 : n * n; }

}  // namespace smi
