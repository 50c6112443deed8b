/*
 * sor.h -- ORACLE interface (test infrastructure; see the header of sor_bc.c).
 * "sor" = sicelore oracle.  Plain C so that tests can bind it with ctypes.
 */
#ifndef SOR_H
#define SOR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* BarcodeMatchTester$Matches$OneMatch */
typedef struct {
    int64_t read_seq;    /* un-mutated window (2-bit long, may be N-poisoned) */
    int64_t matching_bc; /* barcode that was hit */
    int32_t ed, subs, ins, dels, offset, length;
} sor_match_t;

/* what Parser.assignBarcode leaves in ReadScanResult$BarcodeResult */
typedef struct {
    int64_t bc;
    int32_t found, ed, ed_sec, offset, ins_minus_del, bc_start, bc_end, n_matches;
    uint64_t n_probes;
} sor_assign_t;

typedef struct sor_set sor_set;

int64_t sor_twobit_encode(const char *s, int n);
void sor_twobit_decode(int64_t seq, int len, char *out);
int64_t sor_twobit_revcomp(int64_t seq, int len);
void sor_replace_deg(int64_t seq, int pos, int len, int64_t out[4]);
void sor_insert_deg(int64_t seq, int pos, int len, int64_t out[4]);
int64_t sor_delete_byte(int64_t seq, int base_add_at_end4, int pos, int len);
int sor_fourbit_encode_char(unsigned char c);
int sor_fourbit_complement(int b);

sor_set *sor_set_new(const int64_t *keys, size_t n);
void sor_set_free(sor_set *s);
size_t sor_set_size(const sor_set *s);
int sor_set_contains(const sor_set *s, int64_t key);

int sor_bc_match(const sor_set *search, int64_t seq, int len, int ed, int skip_full_matches, int allow_indels,
                 const uint8_t *post4, int post_len, int offset, int do_next_level_if_match_found, sor_match_t *out,
                 int max_out, uint64_t *n_probes);

int sor_assign_barcode(const sor_set *search, const char *stranded, int read_len, int adapterpos, int max_ed,
                       int test_plus_minus, int five_prime, int bc_len, sor_assign_t *res);


/* ---- read scan (sor_scan.c) ------------------------------------------------------------------------------ */
/* bit values = ReadFlags$Flags.getValue() (FJ!nanoporereadscanner/stats/ReadFlags.java:L72-109: FAILED = 0x20, PASSED_FWD = 0x100,
 * ...), pinned by tests/golden/ref_exec_pass2_*.json `flag_values` (read from the reference's own enum by tools/jvm_exec.py);
 * not the ordinals: the all-ones ALL_READS_AFTER_SPLIT sits between bit 2 and bit 3 */
#define SOR_F_PASSED_TOTAL (1ull << 4)
#define SOR_F_FAILED (1ull << 5)
#define SOR_F_PASSED_TOT_TSO (1ull << 10)
#define SOR_F_TSO_5P (1ull << 17)
#define SOR_F_TSO_3P (1ull << 18)
#define SOR_F_TSO_5P_AND_3P (1ull << 22)
#define SOR_F_TSO_5P_AND_3P_FAILED (1ull << 23)
#define SOR_F_PASSED_FWD (1ull << 8)
#define SOR_F_PASSED_REV (1ull << 9)
#define SOR_F_POLY_T_5P (1ull << 11)
#define SOR_F_POLY_A_3P (1ull << 12)
#define SOR_F_POLY_A_NOT_FOUND (1ull << 13)
#define SOR_F_POLY_T_5P_POLY_A_3P (1ull << 14)
#define SOR_F_ADAPTER_5P (1ull << 15)
#define SOR_F_ADAPTER_3P (1ull << 16)
#define SOR_F_ADAPTER_SELECTED_DESP_BOTH (1ull << 19)
#define SOR_F_READ_TOO_SHORT (1ull << 20)
#define SOR_F_ADAPTER_5P_AND_3P (1ull << 21)

typedef struct { /* shipped values: Jar/config.xml:21,55-59,95-105 */
    int32_t min_read_length;        /* 200 */
    int32_t polya_len;              /* 15 */
    float polya_frac;               /* 0.75 */
    int32_t window_polya;           /* 150 */
    int32_t min_adapter_3p_matches; /* 8 */
    int32_t min_mean_bc_qv;         /* 8 */
    int32_t min_mean_read_qv;       /* 8 */
    /* the TSO of the read scan (tso_for3pBarcoding: Jar/config.xml:155-166); tso[0] == 0: the shipped values */
    char tso[20];                   /* sequence: 16 bases, AACGCAGAGTACATGG */
    int32_t tso_window;             /* windowForTSOsearch 90 */
    int32_t tso_max_mm;             /* maxNeedlemanMismatches 5 */
    int32_t tso_min_consec;         /* minTSO_NeedlemanConsecutiveMatches 8 */
    int32_t tso_min_two;            /* minTSO_TwoBestConsecutiveMatches 12 */
} sor_scan_params;

typedef struct {
    uint64_t flags;
    int32_t adapter_found, reverse;       /* reverse = 1: stranded read = reverse complement of the raw read */
    int32_t polya_start, polya_end;       /* PS / PE, stranded 1-based */
    int32_t adapter_start, adapter_end;   /* AS / AE, stranded 1-based */
    int32_t scan_end;                     /* adapter end in scan orientation (1-based) */
    int32_t adapter_nmis;                 /* getNerrorsNeedleman of the accepted alignment */
    int32_t n_cand_fwd, n_cand_rev;       /* NW candidates per side (-1: no polyT on that side) */
    int32_t pass1_ok;                     /* UsedCellBCListGenerator quality filter */
    float mean_qv_bc, mean_qv_read;
    int32_t tso_start, tso_end;           /* TSOresult.start / .end (0 = null); T= prints tso_end */
} sor_scan_result;

uint64_t sor_finalize_flag(uint64_t flags);
int sor_find_polyt(const uint8_t *seq4, int n, int minlen, float minfrac, int window, int *begin1, int *end1);
int sor_scan_read_3p(const char *read, const char *qual, int len, const char *adapter, int max_mm,
                     const sor_scan_params *par, sor_scan_result *out);
/* 5' barcoding (PolyATadapterAnalyzer_5pBCUMI): max_mm = maxNeedlemanMismatches + 1, window = AdapterSearchWindow */
int sor_scan_read_5p(const char *read, const char *qual, int len, const char *adapter, int max_mm,
                     const sor_scan_params *par, int window, int dont_search_polya, sor_scan_result *out);
int sor_scan_batch_3p(const char *reads, const char *quals, const uint64_t *offsets, size_t n, const char *adapter,
                      int max_mm, const sor_scan_params *par, sor_scan_result *out, int32_t *status, int n_threads);
int sor_nw_strings(const char *adapter, const char *read_slice, char *a1, char *dots, char *a2, float *n_errors,
                   int *ins, int *del, int *sub, float *end5);

/* probes for tests/test_ref_exec.py: out9 = {alignment length, hasN3pConsecutiveMatches(6), nMismatchesInAlignment, substitutions,
 * deletions, insertions, getNconsecutiveMatchesNeedleman, getSumOfBestTwoMatchStretchesNeedleman, getOffsetForReadEnd};
 * out_f2 = {countErrorsInNeedleman, countIndelsMismatchesEndOfRead(5)}; last_row = bottom row of the score table */
int sor_nw_stats(const char *adapter, const char *read_slice, int32_t *out9, float *out_f2, int32_t *last_row);
int sor_kmers4_matching(const char *adapter, const char *read, int pos1);

/* ---- read-name writer (sor_name.c) ---- */
int sor_format_read_name(const char *read_name, const char *raw_seq, const char *raw_qual, int len,
                         const sor_scan_result *scan, const sor_assign_t *bc, int rank, uint32_t read_id, int five_prime,
                         char *out, size_t cap);
int sor_fmt_dec1(float f, char *out);
int sor_fastq_record(const char *read_name, const char *qual_header, const char *raw_seq, const char *raw_qual, int len,
                     const sor_scan_result *scan, const sor_assign_t *bc, int rank, uint32_t read_id, int five_prime,
                     int trim_fastq, int force_failed, char *out, size_t cap, int *passed);

/* ---- UMI pair distances (sor_umi.c) ---- */
int sor_umi_pair(const uint8_t *w1, const uint8_t *w2);
void sor_umi_matrix(const uint8_t *windows, int n, uint8_t *out);
int sor_umi_window_3p(const char *x, int xlen, int adapter_end, int bc_end, uint8_t *out14);
int sor_umi_window_5p(const char *x, int xlen, int adapter_end, int bc_end, uint8_t *out14);
int sor_limited_compare(const uint8_t *a, int n, const uint8_t *b, int m, int threshold);
/* umis/umi_length other than 12 (config.xml:264): windows of umi_len + 2 codes */
int sor_umi_pair_len(const uint8_t *w1, const uint8_t *w2, int umi_len);
void sor_umi_matrix_len(const uint8_t *windows, int n, int umi_len, uint8_t *out);
int sor_umi_window_3p_len(const char *x, int xlen, int adapter_end, int bc_end, int umi_len, uint8_t *out);
int sor_umi_window_5p_len(const char *x, int xlen, int adapter_end, int bc_end, int umi_len, uint8_t *out);

/* ---- pass-1 finalize (sor_final.c) ---- */
int sor_finalize_used_list(const int64_t *keys, const uint32_t *counts, size_t n, uint32_t record_count, int merge_ed,
                           int min_count_fold, int cells_fold_below_max, int64_t *out_keys, uint32_t *out_counts,
                           uint32_t *out_rank, size_t *n_out);

#ifdef __cplusplus
}
#endif

/* ---- chimera splitter (sor_chimera.c) ---------------------------------------------------------------------- */
#define SOR_F_CHIMERIC_READS_SPLIT (1ull << 1)
#define SOR_F_MULTI_CHIMERIC_READS_DISCARDED (1ull << 2)
#define SOR_F_READS_AFTER_SPLIT (1ull << 3)
/* ChimeraFindernew$SplitPosition$SplitReason ordinals (ChimeraFindernew.java:L364-370) */
enum { SOR_SPLIT_REV_ADAPTER = 0, SOR_SPLIT_FWD_ADAPTER, SOR_SPLIT_RA_FA, SOR_SPLIT_RA_FT, SOR_SPLIT_RT_FA, SOR_SPLIT_RT_FT,
       SOR_SPLIT_READSTART };
typedef struct {
    const char *tso_complete;     /* AAGCAGTGGTATCAACGCAGAGTACAT (config.xml:170) */
    const char *adapter_complete; /* CTACACGACGCTCTTCCGATCT (config.xml:113) */
    int32_t tso_max_errors;       /* 6 */
    int32_t adapter_max_errors;   /* 5 */
    int32_t internal_pat_len;     /* 15 */
    float internal_pat_frac;      /* 0.70 */
    int32_t window_polya;         /* 150 */
    int32_t bc_umi_len;           /* 16 + 12 */
} sor_chimera_params;
typedef struct {
    int32_t n_split;        /* 0..2 cut positions (0-based offsets into the read, String.substring semantics) */
    int32_t pos[2];
    int32_t reason[2];
    int32_t multi_chimeric; /* > 2 split positions: MULTI_CHIMERIC_READS_DISCARDED | FAILED, read kept whole */
    int32_t n_matches;      /* adapter / TSO matches that entered the split rules (diagnostic) */
} sor_chimera_result;
/* 0 ok, -1 where the reference would throw (substring range) */
int sor_chimera_split(const char *read, int len, const sor_chimera_params *par, sor_chimera_result *out);
/* name of fragment k (0-based, 0..n_split) of a split read; returns the length or -1 */
int sor_chimera_fragment_name(const char *read_name, const sor_chimera_result *res, int fragment, char *out, size_t cap);


/* ---- UMI clustering of one (cell, region) group (sor_cluster.c) ---------------------------------------------- */
typedef struct {                 /* shipped values: Jar/config.xml:270-278, UmiClustering.java:L52, UMIparameters.java:L118 */
    int32_t complete_link_ed;    /* umi_completelinkclusteringED 2 */
    int32_t single_link_ed;      /* umi_singlelinkclusteringED 1 */
    int32_t single_link_switch;  /* complexity_threshold_for_switch_to_single_link_clustering 3000 */
    int32_t fold_depth_below_max; /* foldDepthBelowMaxDiscardForClustering 50 */
    int32_t own_clusterer_above; /* NRECORDS_SWITCH_TO_OWNCLUSTERING 100 */
} sor_umi_cluster_params;
typedef struct {
    int32_t center;    /* group-local index of the cluster centre whose UMI the read takes, -1 = not clustered */
    int8_t offset;     /* mean offset (-1, 0, +1) at which the centre's 12-mer is cut (getPostBCUMIseqOffset) */
    int8_t ed;         /* UMI_ED: distance to the centre */
    int8_t ed_second;  /* UMI_ED_SECOND_BEST_MATCH, -1 = tag absent */
    int8_t pos2;       /* PlusMinusOneEnum value of the read's own best position vs the centre (predicted-pos flag) */
} sor_umi_assignment;
/* mat: n x n bytes ed | pos1 << 4 | pos2 << 6 (sor_umi_matrix layout); skipped_out (may be NULL): reads flagged
 * UMI_CLUSTERING_SKIPPED_HIGHCOMPLEXITY by the fold-depth filter */
int sor_umi_cluster_group(const uint8_t *mat, int32_t n, const float *mean_qv, const sor_umi_cluster_params *par,
                          sor_umi_assignment *out, uint8_t *skipped_out);


/* ---- genomic-region grouping (sor_group.c) ------------------------------------------------------------------- */
/* pos / has_pos / reverse per read in BAM order (reverse = SAM flag 16); region[i] = ordinal of the read's region or
 * -1; n_done = number of leading reads of the chunk that are final (the rest is re-grouped with the next chunk when
 * keep_data_end is set).  max_dist = max_GenomeDistance_forGrouping (500). */
int sor_region_group(const int32_t *pos, const uint8_t *has_pos, const uint8_t *reverse, int32_t n, int32_t max_dist,
                     int keep_data_end, int32_t *region, int32_t *n_done);
/* returns 1 and *out = reference position, or 0 (Optional.absent) */
int sor_ref_position_at_read_position(const uint32_t *cigar, int n_cigar, int32_t alignment_start, int32_t position,
                                      int32_t *out);

#endif
